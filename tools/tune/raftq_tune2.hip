// raftq_tune2.hip -- second-generation A/B of the sweep on MI355X: data-layout and
// instruction-count experiments for the headline config (N = 5, commit + votes) and
// the other BASELINE configs.  A measurement tool, not part of libraftq.so.
//
// Axes (all combined in one templated kernel, `sweep2`):
//   LAYOUT  0 = peer-major rows [N][ld] (rows ~8 MB apart: 13 concurrent HBM streams per wave)
//           1 = workgroup-tiled   [tile][N][T] (a workgroup's N match rows are one contiguous
//               N*T*8 B chunk, its vote rows one contiguous N*T B chunk)
//   LEAN    0 = C++ compare-exchange (hipcc emits v_cmp_gt + v_cmp_lt per CE, 64-bit VALU
//               address arithmetic per load)
//           1 = one v_cmp_gt_u64 per CE through inline asm, SGPR-base + 32-bit lane offset
//               addressing (global_load ... v_off, s[base]) -- fewer VALU instructions
//   XCD     1 = blockIdx -> tile remap so every XCD (blockIdx % 8) streams one contiguous
//               eighth of the groups
//   VOTE4   1 = every lane owns 4 vote bytes (all 4 waves share the vote work) instead of
//               8 bytes on the first kTile/8 lanes
// Every variant is checked against variant 0's outputs on the same inputs before it is timed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// ---- lean compare-exchange: one compare, selects as separate (dead-code-eliminable) asm ----
__device__ __forceinline__ uint32_t sel32(uint32_t if_false, uint32_t if_true, uint64_t mask) {
  uint32_t r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_false), "v"(if_true), "s"(mask));
  return r;
}
__device__ __forceinline__ void ce_desc_lean(uint64_t& a, uint64_t& b) {
  uint64_t m;
  asm("v_cmp_gt_u64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
  const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
  const uint64_t hi = ((uint64_t)sel32(bh, ah, m) << 32) | sel32(bl, al, m);
  const uint64_t lo = ((uint64_t)sel32(ah, bh, m) << 32) | sel32(al, bl, m);
  a = hi;
  b = lo;
}

template <int N, bool LEAN>
__device__ __forceinline__ uint64_t select_q(uint64_t (&v)[N]) {
  if constexpr (!LEAN) {
    return select_quorum_network<N>(v);
  } else {
#define CE(i, j) ce_desc_lean(v[i], v[j])
    if constexpr (N == 3) { CE(0, 2); CE(0, 1); CE(1, 2); }
    if constexpr (N == 5) {
      CE(0, 3); CE(1, 4); CE(0, 2); CE(1, 3); CE(0, 1); CE(2, 4); CE(1, 2); CE(3, 4); CE(2, 3);
    }
    if constexpr (N == 7) {
      CE(0, 6); CE(2, 3); CE(4, 5); CE(0, 2); CE(1, 4); CE(3, 6); CE(0, 1); CE(2, 5);
      CE(3, 4); CE(1, 2); CE(4, 6); CE(2, 3); CE(4, 5); CE(1, 2); CE(3, 4); CE(5, 6);
    }
#undef CE
    return v[N / 2];
  }
}

struct Args2 {
  const uint64_t* match;      // LAYOUT 0: [N][ld]; LAYOUT 1: [tile][N][T]
  const uint64_t* committed;  // [ld]
  uint64_t* committed_out;    // [ld]
  const uint64_t* first_idx;  // [ld]
  const uint8_t* votes;       // LAYOUT 0: [N][ld]; LAYOUT 1: [tile][N][T]
  uint8_t* outcome;           // [ld]
  uint4* partials;
  uint64_t ld;
  uint32_t n_tiles;
};

template <int N, int GPL, bool GATED, bool VOTES, int LAYOUT, bool LEAN, bool XCD, bool VOTE4, bool NT>
__global__ __launch_bounds__(256) void sweep2(Args2 a) {
  constexpr int T = 256 * GPL;  // groups per workgroup tile
  constexpr int kRounds = GPL / 2;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = LEAN ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  uint32_t tile = blockIdx.x;
  if constexpr (XCD) {
    const uint32_t per = a.n_tiles >> 3;  // n_tiles % 8 == 0
    tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  }
  const uint64_t tile0 = (uint64_t)tile * T;
  // row p of this tile: base element index + stride between peers
  const uint64_t mrow0 = LAYOUT == 0 ? tile0 : tile0 * N;
  const uint64_t mstride = LAYOUT == 0 ? a.ld : (uint64_t)T;
  uint32_t n_changed = 0, won_lost = 0;

  // ---- votes: loads first
  constexpr int VB = VOTE4 ? 4 : 8;            // vote bytes per lane
  constexpr int kVoteLanes = T / VB;           // 256 (VOTE4, GPL=4) or T/8
  static_assert(!VOTE4 || GPL == 4, "VOTE4 needs one 4-byte word per lane: T == 1024");
  const bool vote_lane = VOTES && tid < kVoteLanes;
  uint64_t vv[N];
  if constexpr (VOTES) {
    if (vote_lane) {
#pragma unroll
      for (int p = 0; p < N; ++p) {
        const uint8_t* row = a.votes + (mrow0 + (uint64_t)p * mstride);
        if constexpr (VOTE4) {
          const uint32_t* src = reinterpret_cast<const uint32_t*>(row + 4u * tid);
          vv[p] = NT ? ld_stream(src) : *src;
        } else {
          const uint64_t* src = reinterpret_cast<const uint64_t*>(row + 8u * tid);
          vv[p] = NT ? ld_stream(src) : *src;
        }
      }
    }
  }

  u64x2 m[kRounds][N], c[kRounds], f[kRounds];
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint32_t woff = wave * (64 * GPL) + j * 128;  // uniform when LEAN
    const uint32_t loff = 16u * lane;                   // bytes
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const char* row = reinterpret_cast<const char*>(a.match + (mrow0 + (uint64_t)p * mstride + woff));
      const u64x2* src = reinterpret_cast<const u64x2*>(row + loff);
      m[j][p] = NT ? ld_stream(src) : *src;
    }
    {
      const char* row = reinterpret_cast<const char*>(a.committed + (tile0 + woff));
      const u64x2* src = reinterpret_cast<const u64x2*>(row + loff);
      c[j] = NT ? ld_stream(src) : *src;
    }
    if constexpr (GATED) {
      const char* row = reinterpret_cast<const char*>(a.first_idx + (tile0 + woff));
      const u64x2* src = reinterpret_cast<const u64x2*>(row + loff);
      f[j] = NT ? ld_stream(src) : *src;
    }
  }
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint32_t woff = wave * (64 * GPL) + j * 128;
    uint64_t v0[N], v1[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
      v0[p] = m[j][p].x;
      v1[p] = m[j][p].y;
    }
    const uint64_t mci0 = select_q<N, LEAN>(v0), mci1 = select_q<N, LEAN>(v1);
    u64x2 o;
    o.x = maybe_commit<GATED>(mci0, c[j].x, GATED ? f[j].x : 0);
    o.y = maybe_commit<GATED>(mci1, c[j].y, GATED ? f[j].y : 0);
    const uint64_t b0 = __ballot(o.x != c[j].x), b1 = __ballot(o.y != c[j].y);
    n_changed += __popcll(b0) + __popcll(b1);
    char* row = reinterpret_cast<char*>(a.committed_out + (tile0 + woff));
    u64x2* dst = reinterpret_cast<u64x2*>(row + 16u * lane);
    if (NT) st_stream(dst, o); else *dst = o;
  }

  if constexpr (VOTES) {
    if (vote_lane) {
      uint64_t granted = 0, rejected = 0;
#pragma unroll
      for (int p = 0; p < N; ++p) {
        granted += bytes_equal(vv[p], 0x0101010101010101ull);
        rejected += bytes_equal(vv[p], 0x0202020202020202ull);
      }
      constexpr uint64_t q = N / 2 + 1;
      constexpr uint64_t bias = (0x80ull - q) * 0x0101010101010101ull;
      const uint64_t k80 = 0x8080808080808080ull;
      uint64_t won = ((granted + bias) & k80) >> 7;
      uint64_t lost = (((rejected + bias) & k80) >> 7) & ~won;
      if constexpr (VOTE4) {  // upper 4 byte lanes compared zeros against 1/2: never equal, but mask anyway
        won &= 0xffffffffull;
        lost &= 0xffffffffull;
      }
      const uint64_t out = won | (lost << 1);
      if constexpr (VOTE4) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(a.outcome + tile0 + 4u * tid);
        if (NT) st_stream(dst, (uint32_t)out); else *dst = (uint32_t)out;
      } else {
        uint64_t* dst = reinterpret_cast<uint64_t*>(a.outcome + tile0 + 8u * tid);
        if (NT) st_stream(dst, out); else *dst = out;
      }
      won_lost = (uint32_t)__popcll(won) | ((uint32_t)__popcll(lost) << 16);
    }
  }
  const uint32_t wl = VOTES ? wave_sum_u32(won_lost) : 0u;
  if (lane == 0) {
    uint4 r;
    r.x = n_changed; r.y = wl & 0xffffu; r.z = wl >> 16; r.w = 0;
    a.partials[(uint64_t)tile * 4 + (tid >> 6)] = r;
  }
}

// ---- cache-policy sweep: the shipped kernel's access pattern through raw buffer loads / stores so
// that every gfx950 cache-policy bit combination can be set per instruction
// (aux: 1 = sc0, 2 = nt, 16 = sc1).  Peer-major layout, N = 5 or 7, commit + votes.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int N, int GPL, int LDAUX, int STAUX>
__global__ __launch_bounds__(256) void sweep3(Args2 a) {
  constexpr int T = 256 * GPL;
  constexpr int kRounds = GPL / 2;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t tile0 = (uint64_t)blockIdx.x * T;
  const uint32_t row_bytes = (uint32_t)(T * 8);
  uint32_t n_changed = 0, won_lost = 0;
  constexpr int kVoteLanes = T / 8;
  const bool vote_lane = tid < kVoteLanes;
  uint64_t vv[N];
  if (vote_lane) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.votes + (uint64_t)p * a.ld + tile0), 0, T, 0x00020000);
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, 8u * tid, 0, LDAUX);
      vv[p] = (uint64_t)v.x | ((uint64_t)v.y << 32);
    }
  }
  u32x4 m[kRounds][N], c[kRounds];
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint32_t off = (wave * (64 * GPL) + j * 128) * 8 + 16u * lane;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.match + (uint64_t)p * a.ld + tile0), 0, row_bytes, 0x00020000);
      m[j][p] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LDAUX);
    }
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.committed + tile0), 0, row_bytes, 0x00020000);
    c[j] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LDAUX);
  }
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.committed_out + tile0), 0, row_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint32_t off = (wave * (64 * GPL) + j * 128) * 8 + 16u * lane;
    uint64_t v0[N], v1[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
      v0[p] = (uint64_t)m[j][p].x | ((uint64_t)m[j][p].y << 32);
      v1[p] = (uint64_t)m[j][p].z | ((uint64_t)m[j][p].w << 32);
    }
    const uint64_t c0 = (uint64_t)c[j].x | ((uint64_t)c[j].y << 32), c1 = (uint64_t)c[j].z | ((uint64_t)c[j].w << 32);
    const uint64_t o0 = maybe_commit<false>(select_quorum_network<N>(v0), c0, 0);
    const uint64_t o1 = maybe_commit<false>(select_quorum_network<N>(v1), c1, 0);
    n_changed += __popcll(__ballot(o0 != c0)) + __popcll(__ballot(o1 != c1));
    u32x4 o;
    o.x = (uint32_t)o0; o.y = (uint32_t)(o0 >> 32); o.z = (uint32_t)o1; o.w = (uint32_t)(o1 >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(o, wr, off, 0, STAUX);
  }
  if (vote_lane) {
    uint64_t granted = 0, rejected = 0;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      granted += bytes_equal(vv[p], 0x0101010101010101ull);
      rejected += bytes_equal(vv[p], 0x0202020202020202ull);
    }
    constexpr uint64_t q = N / 2 + 1;
    constexpr uint64_t bias = (0x80ull - q) * 0x0101010101010101ull;
    const uint64_t k80 = 0x8080808080808080ull;
    const uint64_t won = ((granted + bias) & k80) >> 7;
    const uint64_t lost = (((rejected + bias) & k80) >> 7) & ~won;
    const uint64_t out = won | (lost << 1);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.outcome + tile0), 0, T, 0x00020000);
    u32x2 ov;
    ov.x = (uint32_t)out; ov.y = (uint32_t)(out >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(ov, r, 8u * tid, 0, STAUX);
    won_lost = (uint32_t)__popcll(won) | ((uint32_t)__popcll(lost) << 16);
  }
  const uint32_t wl = wave_sum_u32(won_lost);
  if (lane == 0) {
    uint4 r;
    r.x = n_changed; r.y = wl & 0xffffu; r.z = wl >> 16; r.w = 0;
    a.partials[(uint64_t)blockIdx.x * 4 + wave] = r;
  }
}

// ---- data ------------------------------------------------------------------------------------
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
struct Set {
  uint8_t* arena;
  uint64_t *match, *match_t, *committed, *committed_out, *first_idx;
  uint8_t *votes, *votes_t, *outcome;
  uint4* partials;
};

static Set make_set(int N, uint64_t ld, uint64_t pad, int T, uint64_t seed) {
  // pad: row stagger (groups) of the peer-major layout; the tiled layout needs none
  Set s;
  const uint64_t lds = ld + pad;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 4095) / 4096 * 4096; return o; };
  const size_t o_m = carve((size_t)N * lds * 8), o_mt = carve((size_t)N * ld * 8), o_c = carve(lds * 8),
               o_co = carve(lds * 8), o_f = carve(lds * 8), o_v = carve((size_t)N * lds), o_vt = carve((size_t)N * ld),
               o_o = carve(lds), o_p = carve(ld / 64 * sizeof(uint4));
  CK(hipMalloc(&s.arena, off));
  s.match = (uint64_t*)(s.arena + o_m); s.match_t = (uint64_t*)(s.arena + o_mt);
  s.committed = (uint64_t*)(s.arena + o_c); s.committed_out = (uint64_t*)(s.arena + o_co);
  s.first_idx = (uint64_t*)(s.arena + o_f); s.votes = s.arena + o_v; s.votes_t = s.arena + o_vt;
  s.outcome = s.arena + o_o; s.partials = (uint4*)(s.arena + o_p);
  const uint64_t base = 1ull << 30;
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, s.match, (uint64_t)N * lds, seed, 2047ull, base);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, s.committed, lds, seed + 1, 1023ull, base + 512);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, s.first_idx, lds, seed + 2, 2047ull, base);
  hipLaunchKernelGGL(fill_votes_kernel, dim3(2048), dim3(256), 0, 0, s.votes, (uint64_t)N * lds, seed + 3);
  return s;
}

// row-wise retile that understands the stagger: dst[((g/T)*N + p)*T + g%T] = src[p*lds + g]
template <typename E>
__global__ void retile2_kernel(const E* src, E* dst, uint64_t ld, uint64_t lds, int N, int T) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, n = (uint64_t)N * ld;
  for (; i < n; i += stride) {
    const uint64_t p = i / ld, g = i % ld;
    dst[((g / T) * N + p) * T + g % T] = src[p * lds + g];
  }
}

typedef void (*launch2_fn)(const Set&, uint64_t ld, uint64_t pad, hipStream_t);

template <int N, int GPL, bool GATED, bool VOTES, int LAYOUT, bool LEAN, bool XCD, bool VOTE4, bool NT>
static void launch2(const Set& s, uint64_t ld, uint64_t pad, hipStream_t st) {
  Args2 a;
  a.match = LAYOUT ? s.match_t : s.match;
  a.votes = LAYOUT ? s.votes_t : s.votes;
  a.committed = s.committed; a.committed_out = s.committed_out; a.first_idx = s.first_idx;
  a.outcome = s.outcome; a.partials = s.partials;
  a.ld = ld + pad;
  a.n_tiles = (uint32_t)(ld / (256 * GPL));
  hipLaunchKernelGGL((sweep2<N, GPL, GATED, VOTES, LAYOUT, LEAN, XCD, VOTE4, NT>), dim3(a.n_tiles), dim3(256), 0, st, a);
}

template <int N, int GPL, int LDAUX, int STAUX>
static void launch3(const Set& s, uint64_t ld, uint64_t pad, hipStream_t st) {
  Args2 a;
  a.match = s.match; a.votes = s.votes;
  a.committed = s.committed; a.committed_out = s.committed_out; a.first_idx = s.first_idx;
  a.outcome = s.outcome; a.partials = s.partials;
  a.ld = ld + pad;
  a.n_tiles = (uint32_t)(ld / (256 * GPL));
  hipLaunchKernelGGL((sweep3<N, GPL, LDAUX, STAUX>), dim3(a.n_tiles), dim3(256), 0, st, a);
}

struct Variant {
  const char* name;
  int N, GPL, gated, votes, layout, lean, xcd, vote4, nt;
  launch2_fn fn;
};
#define V2(N, GPL, GA, VO, LA, LE, XC, V4, NT) \
  { "sweep2", N, GPL, GA, VO, LA, LE, XC, V4, NT, launch2<N, GPL, GA, VO, LA, LE, XC, V4, NT> }

#define V3(N, GPL, LD, ST) \
  { "sweep3", N, GPL, 0, 1, LD, ST, 0, 0, 1, launch3<N, GPL, LD, ST> }

static const Variant kVariants[] = {
    // config 3 (headline): N=5 commit+votes, streaming policy
    V2(5, 4, false, true, 0, false, false, false, true),  // == shipped kernel (reference for the others)
    V2(5, 4, false, true, 0, true, false, false, true),   // lean
    V2(5, 4, false, true, 0, false, true, false, true),   // xcd
    V2(5, 4, false, true, 0, false, false, true, true),   // vote4
    V2(5, 4, false, true, 0, true, true, true, true),     // lean+xcd+vote4
    V2(5, 4, false, true, 1, false, false, false, true),  // tiled
    V2(5, 4, false, true, 1, true, false, false, true),   // tiled lean
    V2(5, 4, false, true, 1, true, true, false, true),    // tiled lean xcd
    V2(5, 4, false, true, 1, true, false, true, true),    // tiled lean vote4
    V2(5, 4, false, true, 1, true, true, true, true),     // tiled lean xcd vote4
    V2(5, 2, false, true, 1, true, false, false, true),   // tiled lean GPL=2
    V2(5, 8, false, true, 1, true, false, false, true),   // tiled lean GPL=8
    V2(5, 4, false, true, 1, true, false, false, false),  // tiled lean cached policy
    // cache-policy bits through buffer loads/stores (layout column = load aux, lean column = store aux;
    // aux: 1 sc0, 2 nt, 16 sc1): reference nt/nt = (2, 2)
    V3(5, 4, 2, 2), V3(5, 4, 0, 0), V3(5, 4, 2, 0), V3(5, 4, 0, 2), V3(5, 4, 18, 2), V3(5, 4, 16, 2), V3(5, 4, 17, 2),
    V3(5, 4, 19, 2), V3(5, 4, 2, 18), V3(5, 4, 2, 3), V3(5, 4, 2, 19), V3(5, 4, 2, 17), V3(5, 4, 18, 18), V3(5, 4, 3, 3),
    V3(5, 4, 1, 2), V3(5, 4, 2, 16),
    // config 5: gated
    V2(5, 4, true, false, 0, false, false, false, true), V2(5, 4, true, false, 1, true, false, false, true),
    // config 2: N=3 commit only
    V2(3, 4, false, false, 0, false, false, false, true), V2(3, 4, false, false, 1, true, false, false, true),
    V2(3, 8, false, false, 1, true, false, false, true),
    // config 4 shard: 2M x 7
    V2(7, 4, false, true, 0, false, false, false, true), V2(7, 4, false, true, 0, true, false, false, true),
    V2(7, 4, false, true, 1, true, false, false, true), V2(7, 4, false, true, 1, true, true, true, true),
};

static double bytes_per_group(const Variant& v) {
  double b = 8.0 * v.N + 8 + 8;
  if (v.gated) b += 8;
  if (v.votes) b += v.N + 1;
  return b;
}

int main(int argc, char** argv) {
  int reps = 300;
  if (argc > 1) reps = atoi(argv[1]);
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint64_t pad = 288;
  int curN = -1, curT = -1;
  uint64_t curG = 0;
  std::vector<Set> sets;
  std::vector<uint64_t> ref_c;
  std::vector<uint8_t> ref_o;
  for (const Variant& v : kVariants) {
    const uint64_t G = v.N == 7 ? 2ull << 20 : 1ull << 20, ld = G;
    const int T = 256 * v.GPL;
    if (v.N != curN || G != curG || T != curT) {
      for (auto& s : sets) (void)hipFree(s.arena);
      sets.clear();
      const double set_bytes = ld * (16.0 * v.N + 24 + 2 * v.N + 1);
      const int K = (int)(1.6 * 1024 * 1024 * 1024 / (ld * (8.0 * v.N + 16 + v.N + 1))) + 1;
      (void)set_bytes;
      for (int k = 0; k < K; ++k) {
        Set s = make_set(v.N, ld, pad, T, 4000 * v.N + k);
        hipLaunchKernelGGL((retile2_kernel<uint64_t>), dim3(2048), dim3(256), 0, 0, s.match, s.match_t, ld, ld + pad, v.N, T);
        hipLaunchKernelGGL((retile2_kernel<uint8_t>), dim3(2048), dim3(256), 0, 0, s.votes, s.votes_t, ld, ld + pad, v.N, T);
        sets.push_back(s);
      }
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      curN = v.N; curG = G; curT = T;
    }
    // correctness of this variant against the first variant of its (N, gated, votes) family
    {
      CK(hipMemsetAsync(sets[0].committed_out, 0xEE, ld * 8, st));
      CK(hipMemsetAsync(sets[0].outcome, 0xEE, ld, st));
      v.fn(sets[0], ld, pad, st);
      CK(hipStreamSynchronize(st));
      std::vector<uint64_t> c(ld);
      std::vector<uint8_t> o(ld);
      CK(hipMemcpy(c.data(), sets[0].committed_out, ld * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(o.data(), sets[0].outcome, ld, hipMemcpyDeviceToHost));
      const bool is_ref = v.name[5] == '2' && v.layout == 0 && !v.lean && !v.xcd && !v.vote4;
      if (is_ref) {
        ref_c = c;
        ref_o = o;
      } else {
        const bool okc = ref_c.size() == c.size() && memcmp(ref_c.data(), c.data(), ld * 8) == 0;
        const bool oko = !v.votes || (ref_o.size() == o.size() && memcmp(ref_o.data(), o.data(), ld) == 0);
        if (!okc || !oko) {
          printf("{\"kernel\":\"sweep2\",\"N\":%d,\"layout\":%d,\"lean\":%d,\"xcd\":%d,\"vote4\":%d,\"MISMATCH\":\"commit %d votes %d\"}\n",
                 v.N, v.layout, v.lean, v.xcd, v.vote4, (int)okc, (int)oko);
          continue;
        }
      }
    }
    for (int rot = 0; rot < 2; ++rot) {
      for (int w = 0; w < 10; ++w) v.fn(sets[rot ? w % sets.size() : 0], ld, pad, st);
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) v.fn(sets[rot ? r % sets.size() : 0], ld, pad, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = 1e3 * ms / reps;
      printf("{\"kernel\":\"%s\",\"N\":%d,\"GPL\":%d,\"gated\":%d,\"votes\":%d,\"layout\":%d,\"lean\":%d,\"xcd\":%d,\"vote4\":%d,"
             "\"nt\":%d,\"G\":%llu,\"rotate\":%d,\"K\":%zu,\"us\":%.3f,\"GBps\":%.1f,\"Gdec_per_s\":%.2f}\n",
             v.name, v.N, v.GPL, v.gated, v.votes, v.layout, v.lean, v.xcd, v.vote4, v.nt, (unsigned long long)G, rot, sets.size(), us,
             G * bytes_per_group(v) / us / 1e3, G / us / 1e3);
      fflush(stdout);
    }
  }
  for (auto& s : sets) (void)hipFree(s.arena);
  return 0;
}
