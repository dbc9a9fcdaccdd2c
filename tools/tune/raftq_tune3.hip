// raftq_tune3.hip -- round-2 A/B of the sweep's LAUNCH SHAPE on MI355X (measurement tool, not part of
// libraftq.so).  Round 1 found the 1M x 5 sweep bound by its launch boundary, not by bytes (12.0 us from
// HBM vs 10.9 us from the Infinity Cache).  This tuner times one "rotation" -- K independent 1M-group
// members, 1.6 GB, HBM regime -- through:
//   single   K launches of the shipped sweep_kernel (round 1's figure)
//   set      ONE launch, grid = tiles x K, members' pointers from a device table (sweep_set_kernel)
//   persist  ONE launch of resident workgroups walking all K x tiles, next tile's loads in flight
//            (sweep_persist_kernel, compiler-scheduled register ping-pong)
//   ring     ONE launch, resident workgroups, each WAVE streams its tiles HBM -> LDS with
//            global_load_lds (nt) through a ring of R slots and runs the network out of LDS
//            (the loader shape MI355X_MICROARCH.md measures at 6.4-6.8 TB/s)
// plus the K-curve of `set`, store-less ceilings, and the other BASELINE shapes.  Every variant's
// commit indices, outcomes and tallies are compared with `single` on two members before it is timed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raftq_kernels.hpp"

using namespace raftqk;

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// ring: per-wave LDS ring fed by LDS-DMA.  GPL = 4 (tile = 1024 groups, a wave owns 256 of them: two rounds
// of 128 for the commit part, 4 vote bytes per lane per peer row).  Slot layout per wave (bytes):
//   [j][p] match rows (2 x N KiB) | [j] committed (2 KiB, stored as row N of round j) | [p] vote rows (N x 256 B)
struct ByteVotes {       // round 1's vote layout, kept for the ring variant only (see below)
  const uint8_t* votes8;  // [N][ld]
  uint8_t* outcome8;      // [ld]
};
typedef __attribute__((address_space(3))) void lds_void3_t;
typedef const __attribute__((address_space(1))) void global_cvoid3_t;

__device__ __forceinline__ uint32_t bytes_equal32(uint32_t v, uint32_t pattern) {
  const uint32_t k7f = 0x7f7f7f7fu;
  const uint32_t x = v ^ pattern;
  uint32_t y = (x & k7f) + k7f;
  y = ~(y | x | k7f);
  return y >> 7;
}

template <int N, int R, int AUX, bool GATED>
__global__ __launch_bounds__(256, 1) void sweep_ring_kernel(const SweepArgs* __restrict__ tab, const ByteVotes* __restrict__ btab,
                                                            uint32_t tiles_per_member, uint32_t total_tiles) {
  constexpr int GPL = 4, kRounds = 2;
  constexpr int kRows = N + 1 + (GATED ? 1 : 0);
  constexpr int kSlot = kRounds * kRows * 1024 + N * 256;
  constexpr int L = kRounds * kRows + N;  // LDS-DMA instructions per tile per wave
  static_assert((R - 1) * L <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* wbase = smem + (size_t)wave * R * kSlot;
  const uint32_t stride = gridDim.x;
  const uint32_t first = blockIdx.x;
  if (first >= total_tiles) return;
  const uint32_t n_mine = (total_tiles - first + stride - 1) / stride;

  auto issue = [&](uint32_t i) {
    const uint32_t lin = first + i * stride;
    const uint32_t m = lin / tiles_per_member, tile = lin - m * tiles_per_member;
    const SweepArgs a = tab[m];
    const ByteVotes bv = btab[m];
    unsigned char* sb = wbase + (size_t)(i % R) * kSlot;
    const uint64_t tile0 = (uint64_t)tile * 1024 + (uint64_t)wave * 256;
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
      const uint64_t g = tile0 + (uint64_t)j * 128 + 2 * lane;
#pragma unroll
      for (int p = 0; p < N; ++p)
        __builtin_amdgcn_global_load_lds((global_cvoid3_t*)(a.match + (uint64_t)p * a.ld + g),
                                         (lds_void3_t*)(sb + (j * kRows + p) * 1024), 16, 0, AUX);
      __builtin_amdgcn_global_load_lds((global_cvoid3_t*)(a.committed + g), (lds_void3_t*)(sb + (j * kRows + N) * 1024), 16,
                                       0, AUX);
      if constexpr (GATED)
        __builtin_amdgcn_global_load_lds((global_cvoid3_t*)(a.first_idx + g),
                                         (lds_void3_t*)(sb + (j * kRows + N + 1) * 1024), 16, 0, AUX);
    }
#pragma unroll
    for (int p = 0; p < N; ++p)
      __builtin_amdgcn_global_load_lds((global_cvoid3_t*)(bv.votes8 + (uint64_t)p * a.ld + tile0 + 4 * lane),
                                       (lds_void3_t*)(sb + kRounds * kRows * 1024 + p * 256), 4, 0, AUX);
  };

  for (uint32_t i = 0; i + 1 < (uint32_t)R && i < n_mine; ++i) issue(i);

  for (uint32_t i = 0; i < n_mine; ++i) {
    if (i + R - 1 < n_mine) {
      issue(i + R - 1);
      // R-1 newer tiles (and the stores of older ones) may stay in flight; everything older has landed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * L) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint32_t lin = first + i * stride;
    const uint32_t m = lin / tiles_per_member, tile = lin - m * tiles_per_member;
    const SweepArgs a = tab[m];
    const ByteVotes bv = btab[m];
    const unsigned char* sb = wbase + (size_t)(i % R) * kSlot;
    const uint64_t tile0 = (uint64_t)tile * 1024 + (uint64_t)wave * 256;
    uint32_t n_changed = 0;
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
      const uint64_t g = tile0 + (uint64_t)j * 128 + 2 * lane;
      uint64_t v0[N], v1[N];
#pragma unroll
      for (int p = 0; p < N; ++p) {
        const u64x2 t = *reinterpret_cast<const u64x2*>(sb + (j * kRows + p) * 1024 + lane * 16);
        v0[p] = t.x;
        v1[p] = t.y;
      }
      const u64x2 c = *reinterpret_cast<const u64x2*>(sb + (j * kRows + N) * 1024 + lane * 16);
      u64x2 f;
      f.x = f.y = 0;
      if constexpr (GATED) f = *reinterpret_cast<const u64x2*>(sb + (j * kRows + N + 1) * 1024 + lane * 16);
      const uint64_t mci0 = select_quorum_network<N>(v0), mci1 = select_quorum_network<N>(v1);
      u64x2 o;
      o.x = maybe_commit<GATED>(mci0, c.x, f.x);
      o.y = maybe_commit<GATED>(mci1, c.y, f.y);
      const uint64_t b0 = __ballot(o.x != c.x), b1 = __ballot(o.y != c.y);
      n_changed += __popcll(b0) + __popcll(b1);
      if (a.changed_bits != nullptr && lane == 0) {
        u64x2 w;
        w.x = b0;
        w.y = b1;
        stg<false>(reinterpret_cast<u64x2*>(a.changed_bits + (g >> 6)), w);
      }
      stg<true>(reinterpret_cast<u64x2*>(a.committed_out + g), o);
    }
    uint32_t granted = 0, rejected = 0;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(sb + kRounds * kRows * 1024 + p * 256 + lane * 4);
      granted += bytes_equal32(v, 0x01010101u);
      rejected += bytes_equal32(v, 0x02020202u);
    }
    constexpr uint32_t q = N / 2 + 1;
    constexpr uint32_t bias = (0x80u - q) * 0x01010101u;
    const uint32_t won = ((granted + bias) & 0x80808080u) >> 7;
    const uint32_t lost = (((rejected + bias) & 0x80808080u) >> 7) & ~won;
    stg<true>(reinterpret_cast<uint32_t*>(bv.outcome8 + tile0 + 4 * lane), won | (lost << 1));
    const uint32_t wl = wave_sum_u32((uint32_t)__popc(won) | ((uint32_t)__popc(lost) << 16));
    if (lane == 0) {
      uint4 r;
      r.x = n_changed;
      r.y = wl & 0xffffu;
      r.z = wl >> 16;
      r.w = 0;
      stg_u4(a.partials + ((uint64_t)tile * 4 + wave), r);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// xcd: the set grid with an XCD-aware tile order.  Consecutive workgroup ids go round-robin to the 8 XCDs, so in the
// plain grid every XCD touches every 2 MB page of every row; here workgroup id w works on tile (w % 8) * (T / 8) + w / 8
// of the K x tiles sequence, i.e. every XCD streams one contiguous eighth (its own pages, its own L2 / TLB working set).
template <int N, int GPL, bool GATED, bool VOTES, int POLICY>
__global__ __launch_bounds__(256) void sweep_set_xcd_kernel(const SweepArgs* __restrict__ tab, uint32_t tiles_per_member,
                                                            uint32_t total_tiles) {
  const uint32_t w = blockIdx.x;
  const uint32_t per = total_tiles >> 3;  // total_tiles % 8 == 0 (tiles per member is a multiple of 8)
  const uint32_t lin = (w & 7u) * per + (w >> 3);
  const uint32_t m = lin / tiles_per_member;
  SweepArgs a = tab[m];
  a.changed_bits = nullptr;
  sweep_tile<N, GPL, true, GATED, VOTES, POLICY, true>(a, lin - m * tiles_per_member);
}

// ---------------------------------------------------------------------------------------------------
// tiled: round 1 measured a workgroup-tiled match layout ([tile][N][T]: a workgroup's N rows are ONE contiguous
// N*T*8-byte chunk instead of N rows megabytes apart) within +-1.5 % of the peer-major rows -- but in the
// launch-bound regime of one 1M-group launch at a time.  Here it is again under the set dispatch, where the launch
// boundary no longer hides what the memory system does.  Only the match rows move; everything else as shipped.
template <int N, int GPL, bool VOTES, int POLICY>
__global__ __launch_bounds__(256) void sweep_set_tiled_kernel(const SweepArgs* __restrict__ tab, const uint64_t* const* __restrict__ match_t) {
  constexpr int T = 256 * GPL, kRounds = GPL / 2;
  constexpr bool NT = (POLICY & kLdNT) != 0;
  SweepArgs a = tab[blockIdx.y];
  a.changed_bits = nullptr;
  const uint64_t* mt = match_t[blockIdx.y];
  const uint32_t tid = threadIdx.x, tile = blockIdx.x;
  const uint64_t tile0 = (uint64_t)tile * T;
  TileRegs<N, GPL, true, false, VOTES> r;
  if constexpr (VOTES) {
    if (tid < T / 8) r.vw[0] = ldg<NT>(reinterpret_cast<const u32x4p*>(a.votes + (tile0 + 8ull * tid) * vote_word_bytes(N)));
  }
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint32_t in_tile = (tid >> 6) * (64 * GPL) + j * 128 + 2 * (tid & 63);
#pragma unroll
    for (int p = 0; p < N; ++p) r.m[j][p] = ldg<NT>(reinterpret_cast<const u64x2*>(mt + ((uint64_t)tile * N + p) * T + in_tile));
    r.c[j] = ldg<NT>(reinterpret_cast<const u64x2*>(a.committed + tile0 + in_tile));
  }
  tile_finish<N, GPL, true, false, VOTES, POLICY, true>(r, a, tile);
}

// chunk: EVERYTHING a tile reads in one contiguous chunk -- [N match rows][committed][vote words], T*(8N + 8 + 2) bytes --
// and everything it writes in another -- [committed'][outcome bits], T*8.25 bytes: one input stream and one output
// stream per workgroup, the closest a sweep can get to the plain copy of `mix`.  T = 2048.
struct ChunkArgs {
  const uint8_t* in;   // [tiles][T*(8N+8+2)]
  uint8_t* out;        // [tiles][T*8 + T/4]
  uint4* partials;
};
template <int N, int POLICY>
__global__ __launch_bounds__(256) void sweep_set_chunk_kernel(const ChunkArgs* __restrict__ tab) {
  constexpr int GPL = 8, T = 2048, kRounds = 4;
  constexpr size_t kIn = (size_t)T * (8 * N + 8 + 2), kOut = (size_t)T * 8 + T / 4;
  constexpr bool NT = (POLICY & kLdNT) != 0;
  const ChunkArgs ca = tab[blockIdx.y];
  const uint32_t tid = threadIdx.x, tile = blockIdx.x;
  const uint8_t* in = ca.in + (size_t)tile * kIn;
  uint8_t* out = ca.out + (size_t)tile * kOut;
  TileRegs<N, GPL, true, false, true> r;
  r.vw[0] = ldg<NT>(reinterpret_cast<const u32x4p*>(in + (size_t)T * (8 * N + 8) + 16 * tid));
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint32_t in_tile = (tid >> 6) * (64 * GPL) + j * 128 + 2 * (tid & 63);
#pragma unroll
    for (int p = 0; p < N; ++p) r.m[j][p] = ldg<NT>(reinterpret_cast<const u64x2*>(in + ((size_t)p * T + in_tile) * 8));
    r.c[j] = ldg<NT>(reinterpret_cast<const u64x2*>(in + ((size_t)N * T + in_tile) * 8));
  }
  // finish through the shipped code: SweepArgs whose output pointers are biased so that "tile * T + x" lands in the chunk
  SweepArgs a;
  a.match = nullptr; a.committed = nullptr; a.first_idx = nullptr; a.votes = nullptr; a.changed_bits = nullptr;
  a.committed_out = reinterpret_cast<uint64_t*>(out) - (size_t)tile * T;
  a.outcome = out + (size_t)T * 8 - ((size_t)tile * T) / 4;
  a.partials = ca.partials;
  a.ld = 0;
  tile_finish<N, GPL, true, false, true, POLICY, true>(r, a, tile);
}

// gather a member's shipped arrays into the chunk layout / scatter the chunk outputs back (verification only)
template <int N>
__global__ void to_chunks_kernel(SweepArgs a, uint64_t G, uint8_t* in) {
  constexpr int T = 2048;
  constexpr size_t kIn = (size_t)T * (8 * N + 8 + 2);
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; g < G; g += stride) {
    uint8_t* c = in + (g / T) * kIn;
    const uint64_t x = g % T;
    for (int p = 0; p < N; ++p) reinterpret_cast<uint64_t*>(c)[(size_t)p * T + x] = a.match[(uint64_t)p * a.ld + g];
    reinterpret_cast<uint64_t*>(c)[(size_t)N * T + x] = a.committed[g];
    reinterpret_cast<uint16_t*>(c + (size_t)T * (8 * N + 8))[x] = reinterpret_cast<const uint16_t*>(a.votes)[g];
  }
}
__global__ void from_chunks_kernel(const uint8_t* out, uint64_t G, uint64_t* committed_out, uint8_t* outcome) {
  constexpr int T = 2048;
  constexpr size_t kOut = (size_t)T * 8 + T / 4;
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; g < G; g += stride) {
    const uint8_t* c = out + (g / T) * kOut;
    committed_out[g] = reinterpret_cast<const uint64_t*>(c)[g % T];
    if ((g & 3) == 0) outcome[g >> 2] = c[(size_t)T * 8 + (g % T) / 4];
  }
}

template <typename E>
__global__ void retile_kernel(const E* src, E* dst, uint64_t G, uint64_t ld, int N, int T) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, n = (uint64_t)N * G;
  for (; i < n; i += stride) {
    const uint64_t p = i / G, g = i % G;
    dst[((g / T) * N + p) * T + g % T] = src[p * ld + g];
  }
}

// ---------------------------------------------------------------------------------------------------
// The shipped layout keeps the RequestVote state as ONE packed word per group (2 bits per peer) and the outcome in
// 2 bits (raftq_kernels.hpp).  It started here as an A/B against round 1's N byte rows + byte outcome (VERDICT r01
// item 2(c)): 10.07 vs 10.79 us per 1M x 5 batch, 26.2 vs 29.1 us per 2M x 7 (profiles/r02/tune3_packed_votes.jsonl)
// -- more than the byte count alone predicts (58.25 vs 62 B), because N row streams per wave become one.  The byte
// rows live on in this tool only for the LDS-ring variant, which was measured in that layout.

template <typename W>
__global__ void pack_votes_kernel(const uint8_t* votes, uint64_t ld, int n, uint64_t groups, W* out) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; g < groups; g += stride) {
    uint32_t w = 0;
    for (int p = 0; p < n; ++p) {
      const uint8_t v = votes[(uint64_t)p * ld + g];
      w |= (uint32_t)(v == 1 ? 1u : v == 2 ? 2u : 0u) << (2 * p);
    }
    out[g] = (W)w;
  }
}

// ---------------------------------------------------------------------------------------------------
__global__ void fill_kernel(uint64_t* p, uint64_t n, uint64_t seed, uint64_t mask, uint64_t add) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    p[i] = (z & mask) + add;
  }
}
__global__ void fill_votes_kernel(uint8_t* p, uint64_t n, uint64_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 29;
    const unsigned u = (unsigned)(z % 10);
    p[i] = u < 3 ? 0 : (u < 8 ? 1 : 2);
  }
}

// streaming copy of a KNOWN byte count with the sweep's access width (16 B per lane, non-temporal): the
// calibration target for the FETCH_SIZE / WRITE_SIZE counters (tools/pmc_traffic.py)
__global__ __launch_bounds__(256) void copy_ref_kernel(const u64x2* __restrict__ in, uint64_t n_in, u64x2* __restrict__ out,
                                                        uint64_t n_out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  u64x2 acc = {0, 0};
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_in; k += stride) {
    const u64x2 v = __builtin_nontemporal_load(in + k);
    acc ^= v;
    if (k < n_out) __builtin_nontemporal_store(v, out + k);
  }
  if (acc.x == 0x123456789abcdefull && acc.y == 1) out[0] = acc;  // keep the loads alive
}

// the same read : write mix as the sweep (50 B in, 8.25 B out per group at N = 5), as a plain streaming copy of every
// member's arena in ONE launch (blockIdx.y = member): what the memory system gives this mix when nothing is computed
// and nothing is strided -- the practical ceiling the sweep is measured against
__global__ __launch_bounds__(256) void copy_mix_kernel(const SweepArgs* __restrict__ tab, uint64_t n_in, uint64_t n_out) {
  const SweepArgs a = tab[blockIdx.y];
  const u64x2* in = reinterpret_cast<const u64x2*>(a.match);
  u64x2* out = reinterpret_cast<u64x2*>(a.committed_out);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  u64x2 acc = {0, 0};
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_in; k += stride) {
    const u64x2 v = ldg<true>(in + k);
    acc ^= v;
    if (k < n_out) stg<true>(out + k, v);
  }
  if (acc.x == 0x123456789abcdefull && acc.y == 1) stg<false>(out, acc);  // keep the loads alive
}

struct Member {
  uint8_t* arena;
  SweepArgs a;   // votes / outcome in the shipped packed layout
  ByteVotes bv;  // the same votes as byte rows (ring variant)
  uint64_t* match_t8;  // match rows re-laid workgroup-tiled for 2048-group tiles (tiled variant)
  ChunkArgs ck;        // everything per tile in one input / one output chunk (chunk variant; N <= 8)
};

// the product's shape: rows padded to a multiple of 2048 groups plus the 288-group stagger
static Member make_member(int N, uint64_t G, uint64_t seed) {
  Member s;
  const uint64_t ld = G + 288;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 4095) / 4096 * 4096; return o; };
  const size_t o_m = carve((size_t)N * ld * 8), o_c = carve(ld * 8), o_co = carve(ld * 8), o_f = carve(ld * 8),
               o_v = carve((size_t)N * ld), o_o = carve(ld), o_p = carve(G / 128 * sizeof(uint4)), o_v16 = carve(ld * 4), o_mt = carve((size_t)N * G * 8), o_ci = carve((size_t)G * (8 * N + 8 + 2)), o_co2 = carve((size_t)G * 8 + G / 4),
               o_o2 = carve(ld / 4 + 64);
  CK(hipMalloc(&s.arena, off));
  s.a.match = (uint64_t*)(s.arena + o_m);
  s.a.committed = (uint64_t*)(s.arena + o_c);
  s.a.committed_out = (uint64_t*)(s.arena + o_co);
  s.a.first_idx = (uint64_t*)(s.arena + o_f);
  s.bv.votes8 = s.arena + o_v;
  s.bv.outcome8 = s.arena + o_o;
  s.a.votes = s.arena + o_v16;
  s.a.outcome = s.arena + o_o2;
  s.a.changed_bits = nullptr;
  s.a.partials = (uint4*)(s.arena + o_p);
  s.a.ld = ld;
  const uint64_t base = 1ull << 30;
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint64_t*)s.a.match, (uint64_t)N * ld, seed, 2047ull, base);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint64_t*)s.a.committed, ld, seed + 1, 1023ull, base + 512);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint64_t*)s.a.first_idx, ld, seed + 2, 2047ull, base);
  hipLaunchKernelGGL(fill_votes_kernel, dim3(2048), dim3(256), 0, 0, (uint8_t*)s.bv.votes8, (uint64_t)N * ld, seed + 3);
  CK(hipMemsetAsync(s.arena + o_v16, 0, ld * 4, 0));
  s.ck.in = s.arena + o_ci;
  s.ck.out = s.arena + o_co2;
  s.ck.partials = s.a.partials;
  s.match_t8 = (uint64_t*)(s.arena + o_mt);
  hipLaunchKernelGGL(retile_kernel<uint64_t>, dim3(2048), dim3(256), 0, 0, s.a.match, s.match_t8, G, ld, N, 2048);
  if (N <= 8)
    hipLaunchKernelGGL(pack_votes_kernel<uint16_t>, dim3(2048), dim3(256), 0, 0, s.bv.votes8, ld, N, ld, (uint16_t*)(s.arena + o_v16));
  else
    hipLaunchKernelGGL(pack_votes_kernel<uint32_t>, dim3(2048), dim3(256), 0, 0, s.bv.votes8, ld, N, ld, (uint32_t*)(s.arena + o_v16));
  if (N == 3) hipLaunchKernelGGL(to_chunks_kernel<3>, dim3(2048), dim3(256), 0, 0, s.a, G, (uint8_t*)s.ck.in);
  if (N == 5) hipLaunchKernelGGL(to_chunks_kernel<5>, dim3(2048), dim3(256), 0, 0, s.a, G, (uint8_t*)s.ck.in);
  if (N == 7) hipLaunchKernelGGL(to_chunks_kernel<7>, dim3(2048), dim3(256), 0, 0, s.a, G, (uint8_t*)s.ck.in);
  return s;
}

struct Ctx {
  int N;
  uint64_t G;
  std::vector<Member> mem;
  SweepArgs* tab;  // device table of all members
  ByteVotes* btab;
  uint64_t** mtab;  // per-member tiled match arrays
  ChunkArgs* ctab;
  hipStream_t st;
  int cus;
};

typedef void (*rot_fn)(const Ctx&, uint32_t K, int param);

template <int N, int GPL, bool GATED, bool VOTES, int POLICY>
static void rot_single(const Ctx& c, uint32_t K, int) {
  for (uint32_t k = 0; k < K; ++k)
    hipLaunchKernelGGL((sweep_kernel<N, GPL, true, GATED, VOTES, POLICY, true>), dim3((unsigned)(c.G / (256 * GPL))), dim3(256), 0,
                       c.st, c.mem[k].a);
}
template <int N, int GPL, bool GATED, bool VOTES, int POLICY>
static void rot_set(const Ctx& c, uint32_t K, int) {
  hipLaunchKernelGGL((sweep_set_kernel<N, GPL, true, GATED, VOTES, POLICY, true>), dim3((unsigned)(c.G / (256 * GPL)), K),
                     dim3(256), 0, c.st, (const SweepArgs*)c.tab, 0u);
}
template <int N, int GPL, bool GATED, bool VOTES, int POLICY, int MINW>
static void rot_persist(const Ctx& c, uint32_t K, int wg_per_cu) {
  const uint32_t tiles = (uint32_t)(c.G / (256 * GPL));
  hipLaunchKernelGGL((sweep_persist_kernel<N, GPL, true, GATED, VOTES, POLICY, true, MINW>), dim3(c.cus * wg_per_cu), dim3(256), 0,
                     c.st, (const SweepArgs*)c.tab, tiles, tiles * K, 0u);
}
template <int N, bool VOTES, int POLICY>
static void rot_set_tiled(const Ctx& c, uint32_t K, int) {
  hipLaunchKernelGGL((sweep_set_tiled_kernel<N, 8, VOTES, POLICY>), dim3((unsigned)(c.G / 2048), K), dim3(256), 0, c.st,
                     (const SweepArgs*)c.tab, (const uint64_t* const*)c.mtab);
}
template <int N, int POLICY>
static void rot_set_chunk(const Ctx& c, uint32_t K, int) {
  hipLaunchKernelGGL((sweep_set_chunk_kernel<N, POLICY>), dim3((unsigned)(c.G / 2048), K), dim3(256), 0, c.st, (const ChunkArgs*)c.ctab);
}
template <int N, int GPL, bool GATED, bool VOTES, int POLICY>
static void rot_set_xcd(const Ctx& c, uint32_t K, int) {
  const uint32_t tiles = (uint32_t)(c.G / (256 * GPL));
  hipLaunchKernelGGL((sweep_set_xcd_kernel<N, GPL, GATED, VOTES, POLICY>), dim3(tiles * K), dim3(256), 0, c.st, (const SweepArgs*)c.tab,
                     tiles, tiles * K);
}
template <int N, int R, int AUX, bool GATED>
static void rot_ring(const Ctx& c, uint32_t K, int wg_per_cu) {
  constexpr int kRows = N + 1 + (GATED ? 1 : 0);
  constexpr size_t lds = (size_t)4 * R * (2 * kRows * 1024 + N * 256);
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute((const void*)sweep_ring_kernel<N, R, AUX, GATED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  const uint32_t tiles = (uint32_t)(c.G / 1024);
  hipLaunchKernelGGL((sweep_ring_kernel<N, R, AUX, GATED>), dim3(c.cus * wg_per_cu), dim3(256), lds, c.st, (const SweepArgs*)c.tab,
                     (const ByteVotes*)c.btab, tiles, tiles * K);
}

struct Variant {
  const char* name;
  int N, gated, votes;
  uint64_t G;
  int param;       // workgroups per CU for the persistent shapes
  const char* note;
  rot_fn fn;
  bool is_ref;
};

static double bytes_per_group(int N, int gated, int votes) {
  return 8.0 * N + 8 + 8 + (gated ? 8 : 0) + (votes ? (N <= 8 ? 2 : 4) + 0.25 : 0);  // shipped packed layout
}

int main(int argc, char** argv) {
  int reps = 40;
  const char* only = nullptr;
  if (argc > 1) reps = atoi(argv[1]);
  if (argc > 2) only = argv[2];
  const uint32_t k_override = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;  // members per rotation (default: 1.6 GB worth)
  const int n_only = argc > 4 ? atoi(argv[4]) : 0;                       // restrict to one peer count
  CK(hipSetDevice(0));
  Ctx c;
  CK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
  CK(hipDeviceGetAttribute(&c.cus, hipDeviceAttributeMultiprocessorCount, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint64_t M1 = 1ull << 20, M2 = 2ull << 20;
#define SINGLE(N, GPL, GA, VO, P) rot_single<N, GPL, GA, VO, P>
#define SETV(N, GPL, GA, VO, P) rot_set<N, GPL, GA, VO, P>
  const std::vector<Variant> variants = {
      // ---- config 3: 1M x 5, commit + votes
      {"single", 5, 0, 1, M1, 0, "K launches, nt loads (shipped r01)", SINGLE(5, 4, false, true, 1), true},
      {"single", 5, 0, 1, M1, 0, "K launches, nt loads+stores", SINGLE(5, 4, false, true, 3), false},
      {"set", 5, 0, 1, M1, 0, "GPL4 nt loads", SETV(5, 4, false, true, 1), false},
      {"set", 5, 0, 1, M1, 0, "GPL4 nt loads+stores", SETV(5, 4, false, true, 3), false},
      {"set", 5, 0, 1, M1, 0, "GPL2 nt loads+stores", SETV(5, 2, false, true, 3), false},
      {"set", 5, 0, 1, M1, 0, "GPL8 nt loads+stores", SETV(5, 8, false, true, 3), false},
      {"set", 5, 0, 1, M1, 0, "GPL4 cached", SETV(5, 4, false, true, 0), false},
      {"set", 5, 0, 1, M1, 0, "GPL4 nt, NO STORES (ceiling)", SETV(5, 4, false, true, 3 | kNoStore), false},
      {"set", 5, 0, 1, M1, 0, "GPL8 nt, NO STORES (ceiling)", SETV(5, 8, false, true, 3 | kNoStore), false},
      {"set", 5, 0, 1, M1, 0, "GPL8 nt, NO STORES NO COMPUTE", SETV(5, 8, false, true, 3 | kNoStore | kNoCompute), false},
      {"set", 5, 0, 1, M1, 0, "GPL8 nt, NO COMPUTE (loads + stores only)", SETV(5, 8, false, true, 3 | kNoCompute), false},
      {"set", 5, 0, 1, M1, 0, "GPL2 nt, NO COMPUTE (loads + stores only)", SETV(5, 2, false, true, 3 | kNoCompute), false},
      {"set", 5, 0, 1, M1, 0, "GPL8 nt loads+stores (again)", SETV(5, 8, false, true, 3), false},
      {"persist", 5, 0, 1, M1, 1, "GPL4 nt l+s, 1 WG/CU", (rot_persist<5, 4, false, true, 3, 1>), false},
      {"persist", 5, 0, 1, M1, 2, "GPL4 nt l+s, 2 WG/CU", (rot_persist<5, 4, false, true, 3, 1>), false},
      {"persist", 5, 0, 1, M1, 3, "GPL4 nt l+s, 3 WG/CU", (rot_persist<5, 4, false, true, 3, 1>), false},
      {"persist", 5, 0, 1, M1, 3, "GPL4 nt loads, 3 WG/CU", (rot_persist<5, 4, false, true, 1, 1>), false},
      {"persist", 5, 0, 1, M1, 4, "GPL2 nt l+s, 4 WG/CU", (rot_persist<5, 2, false, true, 3, 1>), false},
      {"persist", 5, 0, 1, M1, 6, "GPL2 nt l+s, 6 WG/CU", (rot_persist<5, 2, false, true, 3, 1>), false},
      {"ring", 5, 0, 1, M1, 1, "R2 nt (aux 2), 1 WG/CU", (rot_ring<5, 2, 2, false>), false},
      {"ring", 5, 0, 1, M1, 1, "R3 nt (aux 2), 1 WG/CU", (rot_ring<5, 3, 2, false>), false},
      {"ring", 5, 0, 1, M1, 1, "R3 default policy, 1 WG/CU", (rot_ring<5, 3, 0, false>), false},
      {"ring", 5, 0, 1, M1, 2, "R2 nt, 2 WG/CU (queued)", (rot_ring<5, 2, 2, false>), false},
      // ---- config 5: 1M x 5 gated
      {"single", 5, 1, 0, M1, 0, "K launches", SINGLE(5, 4, true, false, 1), true},
      {"set", 5, 1, 0, M1, 0, "GPL4 nt loads+stores", SETV(5, 4, true, false, 3), false},
      {"persist", 5, 1, 0, M1, 3, "GPL4 nt l+s, 3 WG/CU", (rot_persist<5, 4, true, false, 3, 1>), false},
      // ---- config 2: 1M x 3 commit
      {"single", 3, 0, 0, M1, 0, "K launches", SINGLE(3, 4, false, false, 1), true},
      {"set", 3, 0, 0, M1, 0, "GPL4 nt loads+stores", SETV(3, 4, false, false, 3), false},
      {"set", 3, 0, 0, M1, 0, "GPL8 nt loads+stores", SETV(3, 8, false, false, 3), false},
      {"persist", 3, 0, 0, M1, 4, "GPL4 nt l+s, 4 WG/CU", (rot_persist<3, 4, false, false, 3, 1>), false},
      // ---- config 4 shard: 2M x 7 commit + votes
      {"single", 7, 0, 1, M2, 0, "K launches, nt loads+stores", SINGLE(7, 4, false, true, 3), true},
      {"set", 7, 0, 1, M2, 0, "GPL4 nt loads+stores", SETV(7, 4, false, true, 3), false},
      {"set", 7, 0, 1, M2, 0, "GPL2 nt loads+stores", SETV(7, 2, false, true, 3), false},
      {"persist", 7, 0, 1, M2, 2, "GPL4 nt l+s, 2 WG/CU", (rot_persist<7, 4, false, true, 3, 1>), false},
      {"ring", 7, 0, 1, M2, 1, "R2 nt, 1 WG/CU", (rot_ring<7, 2, 2, false>), false},
      // ---- "focus": tile size x store policy of the set shape, timed interleaved (median of 9) per family
      {"focus", 5, 0, 1, M1, 0, "single GPL4 P1", SINGLE(5, 4, false, true, 1), true},
      {"focus", 5, 0, 1, M1, 0, "set GPL2 P1", SETV(5, 2, false, true, 1), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL2 P3", SETV(5, 2, false, true, 3), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL4 P1", SETV(5, 4, false, true, 1), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL4 P3", SETV(5, 4, false, true, 3), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL8 P1", SETV(5, 8, false, true, 1), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL8 P3", SETV(5, 8, false, true, 3), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL8 ld:plain st:nt", SETV(5, 8, false, true, 2), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL8 P3 TILED match", (rot_set_tiled<5, true, 3>), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL8 P3 CHUNK (one stream in, one out)", (rot_set_chunk<5, 3>), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL8 P3 XCD-contiguous", (rot_set_xcd<5, 8, false, true, 3>), false},
      {"focus", 5, 0, 1, M1, 0, "set GPL4 P3 XCD-contiguous", (rot_set_xcd<5, 4, false, true, 3>), false},
      {"focus", 5, 1, 0, M1, 0, "single GPL4 P1", SINGLE(5, 4, true, false, 1), true},
      {"focus", 5, 1, 0, M1, 0, "set GPL2 P3", SETV(5, 2, true, false, 3), false},
      {"focus", 5, 1, 0, M1, 0, "set GPL4 P3", SETV(5, 4, true, false, 3), false},
      {"focus", 5, 1, 0, M1, 0, "set GPL8 P3", SETV(5, 8, true, false, 3), false},
      {"focus", 5, 1, 0, M1, 0, "set GPL4 P1", SETV(5, 4, true, false, 1), false},
      {"focus", 3, 0, 0, M1, 0, "single GPL4 P1", SINGLE(3, 4, false, false, 1), true},
      {"focus", 3, 0, 0, M1, 0, "set GPL2 P3", SETV(3, 2, false, false, 3), false},
      {"focus", 3, 0, 0, M1, 0, "set GPL4 P3", SETV(3, 4, false, false, 3), false},
      {"focus", 3, 0, 0, M1, 0, "set GPL8 P3", SETV(3, 8, false, false, 3), false},
      {"focus", 3, 0, 0, M1, 0, "set GPL8 P1", SETV(3, 8, false, false, 1), false},
      {"focus", 3, 0, 0, M1, 0, "set GPL8 P3 TILED match", (rot_set_tiled<3, false, 3>), false},
      {"focus", 7, 0, 1, M2, 0, "single GPL4 P3", SINGLE(7, 4, false, true, 3), true},
      {"focus", 7, 0, 1, M2, 0, "set GPL2 P3", SETV(7, 2, false, true, 3), false},
      {"focus", 7, 0, 1, M2, 0, "set GPL4 P3", SETV(7, 4, false, true, 3), false},
      {"focus", 7, 0, 1, M2, 0, "set GPL2 P1", SETV(7, 2, false, true, 1), false},
      {"focus", 7, 0, 1, M2, 0, "set GPL8 P3 TILED match", (rot_set_tiled<7, true, 3>), false},
      {"focus", 7, 0, 1, M2, 0, "set GPL8 P3 CHUNK (one stream in, one out)", (rot_set_chunk<7, 3>), false},
      {"focus", 7, 0, 1, M2, 0, "set GPL8 P3", SETV(7, 8, false, true, 3), false},
      {"focus", 9, 0, 1, M1, 0, "single GPL4 P3", SINGLE(9, 4, false, true, 3), true},
      {"focus", 9, 0, 1, M1, 0, "set GPL2 P3", SETV(9, 2, false, true, 3), false},
      {"focus", 9, 0, 1, M1, 0, "set GPL4 P3", SETV(9, 4, false, true, 3), false},
  };
  if (only && !strcmp(only, "calib")) {
    // 20 launches, each reading 53 MiB-units and writing 9 of a different member (rotating: cache-cold)
    const uint64_t G = M1;
    std::vector<Member> mem;
    for (int k = 0; k < 27; ++k) mem.push_back(make_member(5, G, 7000 + k));
    CK(hipDeviceSynchronize());
    const uint64_t n_in = G * 53 / 16, n_out = G * 9 / 16;
    for (int r = 0; r < 25; ++r) {
      const Member& m = mem[r % mem.size()];
      hipLaunchKernelGGL(copy_ref_kernel, dim3(2048), dim3(256), 0, c.st, (const u64x2*)m.a.match, n_in, (u64x2*)m.a.committed_out, n_out);
    }
    CK(hipStreamSynchronize(c.st));
    printf("{\"kernel\":\"copy_ref\",\"launches\":25,\"read_bytes\":%llu,\"write_bytes\":%llu}\n", (unsigned long long)(n_in * 16),
           (unsigned long long)(n_out * 16));
    return 0;
  }
  if (only && !strcmp(only, "mix")) {
    const uint64_t G = M1;
    const uint32_t K = k_override ? k_override : 27;
    std::vector<Member> mem;
    std::vector<SweepArgs> host;
    for (uint32_t k = 0; k < K; ++k) {
      mem.push_back(make_member(5, G, 9000 + k));
      host.push_back(mem.back().a);
    }
    SweepArgs* tab = nullptr;
    CK(hipMalloc((void**)&tab, K * sizeof(SweepArgs)));
    CK(hipMemcpy(tab, host.data(), K * sizeof(SweepArgs), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    // per member: the packed layout's bytes (50 in / 8.25 out per group) and round 1's (53 / 9), as 16-byte elements;
    // the match rows + committed are contiguous in the arena (43 MB + 8 MB), so n_in stays inside it
    const struct { const char* what; double rd, wr; } mixes[] = {{"50 B in / 8.25 B out per group (packed votes)", 50, 8.25},
                                                                 {"53 B in / 9 B out per group (round-1 layout)", 53, 9},
                                                                 {"reads only, 50 B per group", 50, 0},
                                                                 {"1 : 1 copy, 29 B in / 29 B out per group", 29.125, 29.125}};
    for (const auto& mx : mixes) {
      const uint64_t n_in = (uint64_t)(G * mx.rd / 16), n_out = (uint64_t)(G * mx.wr / 16);
      for (int blocks : {512, 1024, 2048}) {
        std::vector<double> us;
        for (int rep = 0; rep < 7; ++rep) {
          for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(copy_mix_kernel, dim3(blocks, K), dim3(256), 0, c.st, (const SweepArgs*)tab, n_in, n_out);
          CK(hipEventRecord(e0, c.st));
          for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_mix_kernel, dim3(blocks, K), dim3(256), 0, c.st, (const SweepArgs*)tab, n_in, n_out);
          CK(hipEventRecord(e1, c.st));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          us.push_back(1e3 * ms / reps / K);
        }
        std::sort(us.begin(), us.end());
        const double med = us[us.size() / 2], bytes = (n_in + n_out) * 16.0;
        printf("{\"kernel\":\"copy_mix\",\"what\":\"%s\",\"K\":%u,\"blocks_per_member\":%d,\"us_per_member_median\":%.3f,\"GBps\":%.1f,"
               "\"frac_of_8TBps\":%.4f}\n", mx.what, K, blocks, med, bytes / med / 1e3, bytes / med / 1e3 / 8000.0);
        fflush(stdout);
      }
    }
    return 0;
  }
  const bool focus = only && !strcmp(only, "focus");
  struct Timed { const Variant* v; std::vector<double> us; };
  std::vector<Timed> family;  // verified focus variants of the current (N, G, gated, votes) family
  auto flush_family = [&](uint32_t K) {
    if (family.empty()) return;
    for (int rep = 0; rep < 9; ++rep)
      for (auto& t : family) {
        for (int w = 0; w < 2; ++w) t.v->fn(c, K, t.v->param);
        CK(hipEventRecord(e0, c.st));
        for (int i = 0; i < reps; ++i) t.v->fn(c, K, t.v->param);
        CK(hipEventRecord(e1, c.st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.us.push_back(1e3 * ms / reps / K);
      }
    for (auto& t : family) {
      std::sort(t.us.begin(), t.us.end());
      const double med = t.us[t.us.size() / 2], bpg = bytes_per_group(t.v->N, t.v->gated, t.v->votes);
      printf("{\"kernel\":\"focus\",\"N\":%d,\"G\":%llu,\"gated\":%d,\"votes\":%d,\"K\":%u,\"note\":\"%s\",\"us_per_batch_median\":%.3f,"
             "\"us_min\":%.3f,\"us_max\":%.3f,\"GBps_median\":%.1f,\"frac_of_8TBps\":%.4f}\n",
             t.v->N, (unsigned long long)t.v->G, t.v->gated, t.v->votes, K, t.v->note, med, t.us.front(), t.us.back(),
             t.v->G * bpg / med / 1e3, t.v->G * bpg / med / 1e3 / 8000.0);
    }
    fflush(stdout);
    family.clear();
  };
  int curN = -1;
  uint64_t curG = 0;
  uint32_t K = 0;
  c.tab = nullptr;
  c.btab = nullptr;
  c.mtab = nullptr;
  c.ctab = nullptr;
  std::vector<uint64_t> ref_c[2];
  std::vector<uint8_t> ref_o[2];
  uint64_t ref_tally[3] = {0, 0, 0};
  int ref_gated = -1, ref_votes = -1;
  auto tallies = [&](uint32_t k, uint64_t G, uint64_t out[3]) {
    std::vector<uint4> p(G / 128);  // >= any variant's wave count; zeroed before each checked run
    CK(hipMemcpy(p.data(), c.mem[k].a.partials, p.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = 0;
    for (auto& v : p) { out[0] += v.x; out[1] += v.y; out[2] += v.z; }
  };
  for (const Variant& v : variants) {
    if (focus != !strcmp(v.name, "focus")) continue;
    if (!focus && only && !strstr(v.name, only) && !v.is_ref) continue;
    if (n_only && v.N != n_only) continue;
    if (focus && v.is_ref) flush_family(K);
    if (v.N != curN || v.G != curG) {
      for (auto& m : c.mem) (void)hipFree(m.arena);
      c.mem.clear();
      if (c.tab) (void)hipFree(c.tab);
      K = (uint32_t)(1.6 * 1024 * 1024 * 1024 / (v.G * (8.0 * v.N + 16 + v.N + 1))) + 1;
      if (k_override) K = k_override;
      std::vector<SweepArgs> host;
      std::vector<ByteVotes> phost;
      std::vector<uint64_t*> mhost;
      std::vector<ChunkArgs> chost;
      for (uint32_t k = 0; k < K; ++k) {
        c.mem.push_back(make_member(v.N, v.G, 5000 * v.N + k));
        host.push_back(c.mem.back().a);
        phost.push_back(c.mem.back().bv);
        mhost.push_back(c.mem.back().match_t8);
        chost.push_back(c.mem.back().ck);
      }
      CK(hipMalloc((void**)&c.tab, K * sizeof(SweepArgs)));
      CK(hipMemcpy(c.tab, host.data(), K * sizeof(SweepArgs), hipMemcpyHostToDevice));
      if (c.ctab) (void)hipFree(c.ctab);
      CK(hipMalloc((void**)&c.ctab, K * sizeof(ChunkArgs)));
      CK(hipMemcpy(c.ctab, chost.data(), K * sizeof(ChunkArgs), hipMemcpyHostToDevice));
      if (c.mtab) (void)hipFree(c.mtab);
      CK(hipMalloc((void**)&c.mtab, K * sizeof(uint64_t*)));
      CK(hipMemcpy(c.mtab, mhost.data(), K * sizeof(uint64_t*), hipMemcpyHostToDevice));
      if (c.btab) (void)hipFree(c.btab);
      CK(hipMalloc((void**)&c.btab, K * sizeof(ByteVotes)));
      CK(hipMemcpy(c.btab, phost.data(), K * sizeof(ByteVotes), hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      c.N = v.N; c.G = v.G; curN = v.N; curG = v.G;
      ref_gated = ref_votes = -1;
    }
    // ---- correctness on members 0 and K-1 against the family's `single` reference
    const uint32_t probe[2] = {0, K - 1};
    for (int i = 0; i < 2; ++i) {
      CK(hipMemsetAsync(c.mem[probe[i]].a.committed_out, 0xEE, v.G * 8, c.st));
      CK(hipMemsetAsync(c.mem[probe[i]].a.outcome, 0xEE, v.G / 4, c.st));
      CK(hipMemsetAsync(c.mem[probe[i]].bv.outcome8, 0xEE, v.G, c.st));
      CK(hipMemsetAsync(c.mem[probe[i]].a.partials, 0, v.G / 128 * sizeof(uint4), c.st));
    }
    v.fn(c, K, v.param);
    if (strstr(v.note, "CHUNK"))  // the chunk variant's outputs live in its own layout: bring them back for the comparison
      for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(from_chunks_kernel, dim3(2048), dim3(256), 0, c.st, (const uint8_t*)c.mem[probe[i]].ck.out, v.G,
                           c.mem[probe[i]].a.committed_out, c.mem[probe[i]].a.outcome);
    CK(hipStreamSynchronize(c.st));
    CK(hipGetLastError());
    bool ok = true;
    uint64_t tl[3];
    tallies(K - 1, v.G, tl);
    for (int i = 0; i < 2; ++i) {
      std::vector<uint64_t> cc(v.G);
      std::vector<uint8_t> oo(v.G);
      CK(hipMemcpy(cc.data(), c.mem[probe[i]].a.committed_out, v.G * 8, hipMemcpyDeviceToHost));
      if (!strcmp(v.name, "ring")) {  // the ring variant keeps round 1's byte layout
        CK(hipMemcpy(oo.data(), c.mem[probe[i]].bv.outcome8, v.G, hipMemcpyDeviceToHost));
      } else {                        // shipped layout: 2 bits per group, expanded to a byte each for the comparison
        std::vector<uint8_t> o2(v.G / 4);
        CK(hipMemcpy(o2.data(), c.mem[probe[i]].a.outcome, v.G / 4, hipMemcpyDeviceToHost));
        for (uint64_t g = 0; g < v.G; ++g) oo[g] = (uint8_t)((o2[g / 4] >> (2 * (g % 4))) & 3u);
      }
      if (v.is_ref) {
        ref_c[i] = cc; ref_o[i] = oo;
      } else {
        const bool nostore = strstr(v.note, "NO STORES") != nullptr || strstr(v.note, "NO COMPUTE") != nullptr;
        if (!nostore) {
          ok = ok && ref_gated == v.gated && ref_votes == v.votes && memcmp(ref_c[i].data(), cc.data(), v.G * 8) == 0;
          if (v.votes) ok = ok && memcmp(ref_o[i].data(), oo.data(), v.G) == 0;
        }
      }
    }
    if (v.is_ref) {
      ref_gated = v.gated; ref_votes = v.votes;
      memcpy(ref_tally, tl, sizeof tl);
    } else {
      if (!strstr(v.note, "NO COMPUTE"))  // (a skeleton variant's tallies mean nothing)
        ok = ok && tl[0] == ref_tally[0] && (!v.votes || (tl[1] == ref_tally[1] && tl[2] == ref_tally[2]));
    }
    if (!ok) {
      printf("{\"kernel\":\"%s\",\"N\":%d,\"note\":\"%s\",\"MISMATCH\":true}\n", v.name, v.N, v.note);
      fflush(stdout);
      continue;
    }
    if (focus) {
      family.push_back(Timed{&v, {}});
      continue;
    }
    // ---- timing: `reps` rotations (K members each), two repeats; plus the K-curve for the plain set shape
    std::vector<uint32_t> ks = {K};
    if (!strcmp(v.name, "set") && strstr(v.note, "GPL4 nt loads+stores") && v.N == 5 && !v.gated) ks = {1, 2, 4, 8, 16, K};
    for (uint32_t kk : ks) {
      for (int rep = 0; rep < 2; ++rep) {
        for (int w = 0; w < 3; ++w) v.fn(c, kk, v.param);
        CK(hipEventRecord(e0, c.st));
        const int r = kk == K ? reps : reps * (int)(K / kk);
        for (int i = 0; i < r; ++i) v.fn(c, kk, v.param);
        CK(hipEventRecord(e1, c.st));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_rot = 1e3 * ms / r, us_batch = us_rot / kk;
        printf("{\"kernel\":\"%s\",\"N\":%d,\"G\":%llu,\"gated\":%d,\"votes\":%d,\"K\":%u,\"wg_per_cu\":%d,\"note\":\"%s\",\"rep\":%d,"
               "\"us_per_rotation\":%.2f,\"us_per_batch\":%.3f,\"GBps\":%.1f,\"frac_of_8TBps\":%.4f,\"Gdec_per_s\":%.2f}\n",
               v.name, v.N, (unsigned long long)v.G, v.gated, v.votes, kk, v.param, v.note, rep, us_rot, us_batch,
               v.G * bytes_per_group(v.N, v.gated, v.votes) / us_batch / 1e3,
               v.G * bytes_per_group(v.N, v.gated, v.votes) / us_batch / 1e3 / 8000.0, v.G / us_batch / 1e3);
        fflush(stdout);
      }
    }
  }
  flush_family(K);
  return 0;
}
