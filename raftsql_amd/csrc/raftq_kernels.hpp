// raftq_kernels.hpp -- CDNA4 (gfx950) device code of the batched multi-raft
// quorum sweep.  Hand-written HIP, wave64, no MFMA: the path is integer
// selection over uint64 log indices and bit counts over packed votes, bound by
// HBM bandwidth (DESIGN.md "Kernels").
//
// What it computes, per raft group g (restating etcd/raft, the dependency the
// reference drives from raft.go:214,224,269 -- SURVEY.md 8a rows a5-a8):
//   mci          = q-th largest of match[0..N)[g],  q = N/2+1     (maybeCommit)
//   committed'   = mci > committed && gate ? mci : committed      (raftLog.maybeCommit)
//   outcome      = granted>=q ? won : rejected>=q ? lost : pending (poll)
//
// HBM layout (peer-major SoA, every row padded to `ld` groups, ld % 2048 == 0,
// padding zero-filled so a full tile is always safe to process):
//   match[p*ld + g] u64 | committed[g] u64 | first_idx[g] u64
//   votes[g]: ONE word per group, 2 bits per peer (peer p = bits 2p, 2p+1: 00 no response, 01 granted,
//             10 rejected); 16-bit words for N <= 8, 32-bit for N = 9 | outcome: 2 bits per group
//   (round 1 kept N byte rows and a byte of outcome: 5 + 1 B per decision at N = 5 instead of 2 + 0.25,
//    and N more row streams per wave -- measured 10.07 vs 10.79 us per 1M x 5 batch, profiles/r02)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace raftqk {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxPeers = 9;
constexpr int kTileMax = 2048;       // groups per block at GPL=8; ld granule

struct SweepArgs {
  const uint64_t* match;       // [N][ld]
  const uint64_t* committed;   // [ld]   current commit index
  uint64_t* committed_out;     // [ld]   shadow buffer (may alias committed)
  const uint64_t* first_idx;   // [ld]   first index of cur_term, 0 = none
  const uint8_t* votes;        // [ld] packed vote words (vote_word_bytes(N) each)
  uint8_t* outcome;            // [ld / 4] 2 bits per group
  uint64_t* changed_bits;      // [ld/64] lane-ordered bitmap, or nullptr
  uint4* partials;             // [ld/kTile * kWaves] {changed, won, lost, 0}
  uint64_t ld;
};

// ---------------------------------------------------------------------------
// uint64 compare-exchange, descending: a <- max, b <- min.  gfx950 has no
// v_max_u64; this is one v_cmp_gt_u64 + 4 v_cndmask_b32, and the dead half of
// a CE whose other output is never read is removed by the compiler.
__device__ __forceinline__ void ce_desc(uint64_t& a, uint64_t& b) {
  const bool gt = a > b;
  const uint64_t hi = gt ? a : b;
  const uint64_t lo = gt ? b : a;
  a = hi;
  b = lo;
}

// Optimal-size sorting networks for N <= 9 (0-1-principle checked in
// tests/test_networks.py against the same comparator lists).  Only element
// q-1 = N/2 of the descending order is consumed, so the compiler prunes the
// comparators (and halves of comparators) that cannot reach it.
template <int N>
__device__ __forceinline__ uint64_t select_quorum_network(uint64_t (&v)[N]) {
#define CE(i, j) ce_desc(v[i], v[j])
  if constexpr (N == 2) { CE(0, 1); }
  if constexpr (N == 3) { CE(0, 2); CE(0, 1); CE(1, 2); }
  if constexpr (N == 4) { CE(0, 2); CE(1, 3); CE(0, 1); CE(2, 3); CE(1, 2); }
  if constexpr (N == 5) {
    CE(0, 3); CE(1, 4); CE(0, 2); CE(1, 3); CE(0, 1); CE(2, 4); CE(1, 2); CE(3, 4); CE(2, 3);
  }
  if constexpr (N == 6) {
    CE(0, 5); CE(1, 3); CE(2, 4); CE(1, 2); CE(3, 4); CE(0, 3);
    CE(2, 5); CE(0, 1); CE(2, 3); CE(4, 5); CE(1, 2); CE(3, 4);
  }
  if constexpr (N == 7) {
    CE(0, 6); CE(2, 3); CE(4, 5); CE(0, 2); CE(1, 4); CE(3, 6); CE(0, 1); CE(2, 5);
    CE(3, 4); CE(1, 2); CE(4, 6); CE(2, 3); CE(4, 5); CE(1, 2); CE(3, 4); CE(5, 6);
  }
  if constexpr (N == 8) {
    CE(0, 2); CE(1, 3); CE(4, 6); CE(5, 7); CE(0, 4); CE(1, 5); CE(2, 6); CE(3, 7); CE(0, 1); CE(2, 3);
    CE(4, 5); CE(6, 7); CE(2, 4); CE(3, 5); CE(1, 4); CE(3, 6); CE(1, 2); CE(3, 4); CE(5, 6);
  }
  if constexpr (N == 9) {
    CE(0, 3); CE(1, 7); CE(2, 5); CE(4, 8); CE(0, 7); CE(2, 4); CE(3, 8); CE(5, 6); CE(0, 2);
    CE(1, 3); CE(4, 5); CE(7, 8); CE(1, 4); CE(3, 6); CE(5, 7); CE(0, 1); CE(2, 4); CE(3, 5);
    CE(6, 8); CE(2, 3); CE(4, 5); CE(6, 7); CE(1, 2); CE(3, 4); CE(5, 6);
  }
#undef CE
  return v[N / 2];
}

// Odd-even transposition sort (N rounds of neighbour exchanges): the network
// north_star names for the LDS-staged variant.
template <int N>
__device__ __forceinline__ uint64_t select_quorum_oddeven(uint64_t (&v)[N]) {
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int i = (r & 1); i + 1 < N; i += 2) ce_desc(v[i], v[i + 1]);
  }
  return v[N / 2];
}

// raftLog.maybeCommit's test; GATED uses the compact encoding:
// term(mci) == cur_term  <=>  first_idx != 0 && mci >= first_idx.
template <bool GATED>
__device__ __forceinline__ uint64_t maybe_commit(uint64_t mci, uint64_t committed, uint64_t first_idx) {
  bool adv = mci > committed;
  if constexpr (GATED) adv = adv && (first_idx != 0) && (mci >= first_idx);
  return adv ? mci : committed;
}

// ---------------------------------------------------------------------------
// SWAR byte lanes for the vote tally: a lane owns 8 consecutive groups of one
// peer row as one uint64.
__device__ __forceinline__ uint64_t bytes_equal(uint64_t v, uint64_t pattern) {
  // exact per-byte equality -> 0x01 in every equal byte (no cross-byte carry)
  const uint64_t k7f = 0x7f7f7f7f7f7f7f7full;
  const uint64_t x = v ^ pattern;
  uint64_t y = (x & k7f) + k7f;
  y = ~(y | x | k7f);  // 0x80 where the byte of x is zero
  return y >> 7;
}

// ---------------------------------------------------------------------------
// Packed RequestVote state: one word per group, 2 bits per peer.
__host__ __device__ constexpr int vote_word_bytes(int n_peers) { return n_peers <= 8 ? 2 : 4; }
typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));

// raft.poll over one group's word -> 2-bit outcome (0 pending, 1 won, 2 lost).  A field of 11 is never
// stored (loads canonicalise, ingest writes 01 / 10 only); it would count as "no response" here all the same.
template <int N>
__device__ __forceinline__ uint32_t poll_word(uint32_t w) {
  constexpr uint32_t kLow = 0x55555555u & ((N >= 16) ? 0xffffffffu : ((1u << (2 * N)) - 1u));
  constexpr uint32_t q = N / 2 + 1;
  const uint32_t granted = __popc(w & ~(w >> 1) & kLow);
  const uint32_t rejected = __popc((w >> 1) & ~w & kLow);
  const uint32_t won = granted >= q ? 1u : 0u;
  const uint32_t lost = (won == 0u && rejected >= q) ? 2u : 0u;
  return won | lost;
}

template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  return __builtin_nontemporal_load(p);
}
template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}

// The set kernels read their array pointers from a table in memory, where the compiler cannot see that they
// point to global memory and would fall back to flat_load / flat_store (which tie up lgkmcnt as well and make
// every wait a vmcnt(0)).  These state the address space.
#define RAFTQ_GLOBAL __attribute__((address_space(1)))
template <bool NT, typename T>
__device__ __forceinline__ T ldg(const T* p) {
  const RAFTQ_GLOBAL T* g = (const RAFTQ_GLOBAL T*)p;
  if constexpr (NT) return __builtin_nontemporal_load(g);
  else return *g;
}
template <bool NT, typename T>
__device__ __forceinline__ void stg(T* p, T v) {
  RAFTQ_GLOBAL T* g = (RAFTQ_GLOBAL T*)p;
  if constexpr (NT) __builtin_nontemporal_store(v, g);
  else *g = v;
}
// uint4 is a class type in HIP (no assignment through an address-space-qualified pointer): store it as a native vector
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stg_u4(uint4* p, uint4 v) {
  u32x4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(RAFTQ_GLOBAL u32x4*)p = t;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// ---------------------------------------------------------------------------
// The sweep.  One 256-thread workgroup owns a tile of 256*GPL consecutive
// groups, each of its 4 waves a contiguous 64*GPL of them.  Commit part:
// GPL/2 rounds; in round j a wave covers 128 consecutive groups, lane l
// holding the 16-byte pair {g, g+1}, so every global access is a fully
// coalesced 1 KiB per wave instruction.  All
// (N+1[+1]) * GPL/2 loads of a lane are issued before the first compare.
// Vote part: lane t owns the 8 groups [8t, 8t+8) of the tile as one 8-byte
// load per peer row (votes are 1 byte per group).
//
// NT: stream the inputs/outputs with the non-temporal policy (they are read
// once per sweep).  Chosen per launch by the host; see DESIGN.md for the A/B.
//
// POLICY bits: kLdNT (non-temporal loads), kStNT (non-temporal stores); kNoStore
// is a measurement-only ablation (tuner) that drops the output stores.
constexpr int kLdNT = 1, kStNT = 2, kNoStore = 4;
// kNoCompute (tuner only): the selection network and the vote tally replaced by an XOR of the operands -- the same
// loads and stores with no arithmetic between them: what the memory side alone costs in this kernel shape
constexpr int kNoCompute = 8;
// (The other gfx950 cache-policy bits of the bulk stores -- sc0 / sc1 with and without nt -- were swept in round 2
// through inline-asm stores: plain `nt` is the best, every other combination is equal or up to 7 % slower,
// profiles/r02/tune3_focus_store_policy_and_packed_votes_n5.jsonl.  The asm variants are not kept: stores the
// compiler does not see break its own vmcnt accounting for the loads around them.)

// A tile's inputs in registers: filled by tile_load (every load issued, nothing waited for), consumed by
// tile_finish.  Split so that a persistent kernel can have the next tile's loads in flight while it finishes
// the current one (sweep_persist_kernel); the one-tile kernels call them back to back.
template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES>
struct TileRegs {
  static constexpr int kRounds = GPL / 2;
  u64x2 m[COMMIT ? kRounds : 1][N];
  u64x2 c[COMMIT ? kRounds : 1];
  u64x2 f[COMMIT && GATED ? kRounds : 1];
  u32x4p vw[VOTES ? (N <= 8 ? 1 : 2) : 1];  // the 8 vote words of the lane's 8 groups
};

template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES, int POLICY, int BLOCK = kBlock>
__device__ __forceinline__ void tile_load(TileRegs<N, GPL, COMMIT, GATED, VOTES>& r, const SweepArgs& a,
                                          const uint32_t tile) {
  constexpr bool NT = (POLICY & kLdNT) != 0;
  constexpr int kTile = BLOCK * GPL;
  constexpr int kRounds = GPL / 2;
  const uint32_t tid = threadIdx.x;
  const uint64_t tile0 = (uint64_t)tile * kTile;
  // vote words first: their loads fly while the commit part computes
  constexpr int kVoteLanes = kTile / 8;  // lanes that own 8 groups each
  if constexpr (VOTES) {
    if (tid < kVoteLanes) {  // wave-uniform (kVoteLanes % 64 == 0)
      const uint8_t* src = a.votes + (tile0 + 8ull * tid) * vote_word_bytes(N);  // 16 B (N <= 8) or 32 B of words
      r.vw[0] = ldg<NT>(reinterpret_cast<const u32x4p*>(src));
      if constexpr (N > 8) r.vw[1] = ldg<NT>(reinterpret_cast<const u32x4p*>(src + 16));
    }
  }
  if constexpr (COMMIT) {
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
      const uint64_t g = tile0 + (uint64_t)(tid >> 6) * (64 * GPL) + (uint64_t)j * 128 + 2 * (tid & 63);
#pragma unroll
      for (int p = 0; p < N; ++p) {
        r.m[j][p] = ldg<NT>(reinterpret_cast<const u64x2*>(a.match + (uint64_t)p * a.ld + g));
      }
      r.c[j] = ldg<NT>(reinterpret_cast<const u64x2*>(a.committed + g));
      if constexpr (GATED) r.f[j] = ldg<NT>(reinterpret_cast<const u64x2*>(a.first_idx + g));
    }
  }
}

template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES, int POLICY, bool BITS, int BLOCK = kBlock>
__device__ __forceinline__ void tile_finish(const TileRegs<N, GPL, COMMIT, GATED, VOTES>& r, const SweepArgs& a,
                                            const uint32_t tile) {
  constexpr bool STNT = (POLICY & kStNT) != 0;
  constexpr bool NOSTORE = (POLICY & kNoStore) != 0;
  constexpr int kTile = BLOCK * GPL;
  constexpr int kWavesB = BLOCK / 64;
  constexpr int kRounds = GPL / 2;
  const uint32_t tid = threadIdx.x;
  const uint64_t tile0 = (uint64_t)tile * kTile;
  uint32_t n_changed = 0;  // wave-uniform
  uint32_t won_lost = 0;   // per lane: won | lost << 16
  constexpr int kVoteLanes = kTile / 8;
  const bool vote_lane = VOTES && tid < kVoteLanes;

  if constexpr (COMMIT) {
    // (A/B, profiles/r01/tune_sched_barrier.txt: forcing every load ahead of the first compare with
    // a sched_barrier is SLOWER -- 12.5 vs 12.3 us at N=5, 50 vs 30 us at N=7 from register
    // pressure; hipcc's own split of 8 loads up front + 4 interleaved is kept.)
#pragma unroll
    for (int j = 0; j < kRounds; ++j) {
      const uint64_t g = tile0 + (uint64_t)(tid >> 6) * (64 * GPL) + (uint64_t)j * 128 + 2 * (tid & 63);
      uint64_t v0[N], v1[N];
#pragma unroll
      for (int p = 0; p < N; ++p) {
        v0[p] = r.m[j][p].x;
        v1[p] = r.m[j][p].y;
      }
      uint64_t mci0, mci1;
      if constexpr ((POLICY & kNoCompute) != 0) {
        mci0 = v0[0];
        mci1 = v1[0];
#pragma unroll
        for (int p = 1; p < N; ++p) {
          mci0 ^= v0[p];
          mci1 ^= v1[p];
        }
      } else if constexpr (N == 1) {
        mci0 = v0[0];
        mci1 = v1[0];
      } else {
        mci0 = select_quorum_network<N>(v0);
        mci1 = select_quorum_network<N>(v1);
      }
      u64x2 o;
      if constexpr ((POLICY & kNoCompute) != 0) {
        o.x = mci0 ^ r.c[j].x;
        o.y = mci1 ^ r.c[j].y;
      } else {
        o.x = maybe_commit<GATED>(mci0, r.c[j].x, GATED ? r.f[j].x : 0);
        o.y = maybe_commit<GATED>(mci1, r.c[j].y, GATED ? r.f[j].y : 0);
      }
      const bool ch0 = o.x != r.c[j].x;
      const bool ch1 = o.y != r.c[j].y;
      const uint64_t b0 = __ballot(ch0);
      const uint64_t b1 = __ballot(ch1);
      n_changed += __popcll(b0) + __popcll(b1);
      if constexpr (BITS) {
        // lane-ordered bitmap: word 2k = even groups, 2k+1 = odd groups of the
        // k-th 128-group run (decoded by compact_changed_kernel)
        if (a.changed_bits != nullptr && (tid & 63) == 0) {
          u64x2 w;
          w.x = b0;
          w.y = b1;
          stg<false>(reinterpret_cast<u64x2*>(a.changed_bits + (g >> 6)), w);
        }
      }
      if constexpr (NOSTORE) {
        asm volatile("" ::"v"(o.x), "v"(o.y));
      } else {
        stg<STNT>(reinterpret_cast<u64x2*>(a.committed_out + g), o);
      }
    }
  }

  if constexpr (VOTES) {
    if (vote_lane) {
      const uint64_t g = tile0 + 8ull * tid;
      uint32_t out = 0, n_won = 0, n_lost = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t w;
        if constexpr (N <= 8) {
          const uint32_t pair = k < 2 ? r.vw[0].x : k < 4 ? r.vw[0].y : k < 6 ? r.vw[0].z : r.vw[0].w;
          w = (k & 1) ? pair >> 16 : pair & 0xffffu;
        } else {
          w = k == 0 ? r.vw[0].x : k == 1 ? r.vw[0].y : k == 2 ? r.vw[0].z : k == 3 ? r.vw[0].w
            : k == 4 ? r.vw[1].x : k == 5 ? r.vw[1].y : k == 6 ? r.vw[1].z : r.vw[1].w;
        }
        const uint32_t oc = poll_word<N>(w);
        out |= oc << (2 * k);
        n_won += oc & 1u;
        n_lost += oc >> 1;
      }
      uint16_t* dst = reinterpret_cast<uint16_t*>(a.outcome + (g >> 2));  // 8 groups = 16 bits
      if constexpr (NOSTORE) {
        asm volatile("" ::"v"(out));
      } else {
        stg<STNT>(dst, (uint16_t)out);
      }
      won_lost = n_won | (n_lost << 16);
    }
  }

  const uint32_t wl = VOTES ? wave_sum_u32(won_lost) : 0u;
  if ((tid & 63) == 0) {
    uint4 t;
    t.x = n_changed;
    t.y = wl & 0xffffu;
    t.z = wl >> 16;
    t.w = 0;
    stg_u4(a.partials + ((uint64_t)tile * kWavesB + (tid >> 6)), t);
  }
}

template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES, int POLICY, bool BITS, int BLOCK = kBlock>
__device__ __forceinline__ void sweep_tile(const SweepArgs& a, const uint32_t tile) {
  TileRegs<N, GPL, COMMIT, GATED, VOTES> r;
  tile_load<N, GPL, COMMIT, GATED, VOTES, POLICY, BLOCK>(r, a, tile);
  tile_finish<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS, BLOCK>(r, a, tile);
}

// One handle per launch: blockIdx.x = tile.
template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES, int POLICY, bool BITS, int BLOCK = kBlock>
static __global__ __launch_bounds__(BLOCK) void sweep_kernel(SweepArgs a) {
  sweep_tile<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS, BLOCK>(a, blockIdx.x);
}

// A set of handles per launch (raftq_set_sweep_async): blockIdx.y = member, blockIdx.x = tile.  Every member has
// the same N and padded size; its array pointers come from a device-resident table (wave-uniform: scalar loads).
// One dispatch over K members costs one launch boundary instead of K -- at 1M groups per member the boundary
// (ramp + drain of a 65 MB burst) is 10 % of a sweep (profiles/r02).
template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES, int POLICY, bool BITS, int BLOCK = kBlock>
static __global__ __launch_bounds__(BLOCK) void sweep_set_kernel(const SweepArgs* __restrict__ tab, uint32_t want_bits) {
  SweepArgs a = tab[blockIdx.y];
  if (!want_bits) a.changed_bits = nullptr;
  sweep_tile<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS, BLOCK>(a, blockIdx.x);
}

// Persistent form of the set sweep: gridDim.x workgroups stay resident and walk the K x tiles_per_member tiles
// of the set with stride gridDim.x, the loads of the NEXT tile issued before the current one is finished, so a
// lane always has a tile's worth of HBM requests in flight and no workgroup is ever being launched or retired
// while the sweep runs.  Same arithmetic, same outputs, same per-wave partials as sweep_set_kernel.
template <int N, int GPL, bool COMMIT, bool GATED, bool VOTES, int POLICY, bool BITS, int MINW = 1>
static __global__ __launch_bounds__(kBlock, MINW) void sweep_persist_kernel(const SweepArgs* __restrict__ tab,
                                                                      uint32_t tiles_per_member, uint32_t total_tiles,
                                                                      uint32_t want_bits) {
  using Regs = TileRegs<N, GPL, COMMIT, GATED, VOTES>;
  const uint32_t stride = gridDim.x;
  uint32_t t = blockIdx.x;
  if (t >= total_tiles) return;
  auto locate = [&](uint32_t lin, SweepArgs& a, uint32_t& tile) {
    const uint32_t m = lin / tiles_per_member;
    tile = lin - m * tiles_per_member;
    a = tab[m];
    if (!want_bits) a.changed_bits = nullptr;
  };
  // Two register sets, A and B, used alternately (no copies: a copy would have to wait for the data), and every
  // finish is preceded in straight-line code by the other set's loads, so the wait in front of a tile's
  // arithmetic is vmcnt(one tile of loads), never vmcnt(0) -- a lane always has a tile of requests in flight.
  Regs A, B;
  SweepArgs aA, aB;
  uint32_t tileA, tileB;
  locate(t, aA, tileA);
  tile_load<N, GPL, COMMIT, GATED, VOTES, POLICY>(A, aA, tileA);
  while (true) {
    t += stride;
    if (t >= total_tiles) {
      tile_finish<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS>(A, aA, tileA);
      break;
    }
    locate(t, aB, tileB);
    tile_load<N, GPL, COMMIT, GATED, VOTES, POLICY>(B, aB, tileB);
    tile_finish<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS>(A, aA, tileA);
    t += stride;
    if (t >= total_tiles) {
      tile_finish<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS>(B, aB, tileB);
      break;
    }
    locate(t, aA, tileA);
    tile_load<N, GPL, COMMIT, GATED, VOTES, POLICY>(A, aA, tileA);
    tile_finish<N, GPL, COMMIT, GATED, VOTES, POLICY, BITS>(B, aB, tileB);
  }
}

// Per-member tallies of a set sweep: one workgroup per member sums that member's per-wave partials
// into {changed, won, lost, 0} (u64) -- one small D2H copy for the whole set instead of one per member.
static __global__ __launch_bounds__(kBlock) void set_counts_kernel(const SweepArgs* __restrict__ tab,
                                                                    const uint64_t* __restrict__ n_partials_of,
                                                                    uint64_t* __restrict__ out) {
  __shared__ uint64_t red[3][kWaves];
  const uint4* p = tab[blockIdx.x].partials;
  const uint64_t n_partials = n_partials_of[blockIdx.x];  // a member swept on its own since may have another tile size
  uint64_t c = 0, w = 0, l = 0;
  for (uint64_t i = threadIdx.x; i < n_partials; i += kBlock) {
    const uint4 v = p[i];
    c += v.x;
    w += v.y;
    l += v.z;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    c += __shfl_xor(c, o, 64);
    w += __shfl_xor(w, o, 64);
    l += __shfl_xor(l, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = c;
    red[1][threadIdx.x >> 6] = w;
    red[2][threadIdx.x >> 6] = l;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint64_t t = 0;
    for (int k = 0; k < kWaves; ++k) t += red[threadIdx.x][k];
    out[(uint64_t)blockIdx.x * 4 + threadIdx.x] = t;
  }
  if (threadIdx.x == 3) out[(uint64_t)blockIdx.x * 4 + 3] = 0;
}

// ---------------------------------------------------------------------------
// A/B variant named by north_star: "one wavefront per tile of groups with an
// LDS-staged odd-even sort".  Each wave DMA-stages its tile rows straight from
// HBM into LDS (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR
// round trip), waits on its own vmcnt, reads its two groups' columns back with
// conflict-free ds_read_b128 and runs the odd-even transposition network.
// Every lane reads only bytes its own DMA lane wrote, so no barrier is needed.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_cvoid_t;

template <int N, int GPL, bool GATED, bool VOTES, bool BITS>
static __global__ __launch_bounds__(kBlock) void sweep_lds_kernel(SweepArgs a) {
  constexpr int kTile = kBlock * GPL;
  constexpr int kRounds = GPL / 2;
  constexpr int kRows = N + 1 + (GATED ? 1 : 0);   // match rows, committed, first_idx
  constexpr int kRowBytes = 64 * 16;               // one wave instruction
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  unsigned char* wbase = smem + (size_t)wave * (kRounds * kRows * kRowBytes);
  const uint64_t tile0 = (uint64_t)blockIdx.x * kTile;
  uint32_t n_changed = 0;
  uint32_t won_lost = 0;

#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint64_t g = tile0 + (uint64_t)(tid >> 6) * (64 * GPL) + (uint64_t)j * 128 + 2 * (tid & 63);
    unsigned char* rbase = wbase + (size_t)j * kRows * kRowBytes;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      __builtin_amdgcn_global_load_lds((global_cvoid_t*)(a.match + (uint64_t)p * a.ld + g),
                                       (lds_void_t*)(rbase + p * kRowBytes), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((global_cvoid_t*)(a.committed + g),
                                     (lds_void_t*)(rbase + N * kRowBytes), 16, 0, 0);
    if constexpr (GATED) {
      __builtin_amdgcn_global_load_lds((global_cvoid_t*)(a.first_idx + g),
                                       (lds_void_t*)(rbase + (N + 1) * kRowBytes), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint64_t g = tile0 + (uint64_t)(tid >> 6) * (64 * GPL) + (uint64_t)j * 128 + 2 * (tid & 63);
    const unsigned char* rbase = wbase + (size_t)j * kRows * kRowBytes;
    uint64_t v0[N], v1[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
      const u64x2 t = *reinterpret_cast<const u64x2*>(rbase + p * kRowBytes + lane * 16);
      v0[p] = t.x;
      v1[p] = t.y;
    }
    const u64x2 c = *reinterpret_cast<const u64x2*>(rbase + N * kRowBytes + lane * 16);
    u64x2 f;
    f.x = f.y = 0;
    if constexpr (GATED) f = *reinterpret_cast<const u64x2*>(rbase + (N + 1) * kRowBytes + lane * 16);
    const uint64_t mci0 = select_quorum_oddeven<N>(v0);
    const uint64_t mci1 = select_quorum_oddeven<N>(v1);
    u64x2 o;
    o.x = maybe_commit<GATED>(mci0, c.x, f.x);
    o.y = maybe_commit<GATED>(mci1, c.y, f.y);
    const uint64_t b0 = __ballot(o.x != c.x);
    const uint64_t b1 = __ballot(o.y != c.y);
    n_changed += __popcll(b0) + __popcll(b1);
    if constexpr (BITS) {
      if (a.changed_bits != nullptr && lane == 0) {
        u64x2 w;
        w.x = b0;
        w.y = b1;
        *reinterpret_cast<u64x2*>(a.changed_bits + (g >> 6)) = w;
      }
    }
    *reinterpret_cast<u64x2*>(a.committed_out + g) = o;
  }

  if constexpr (VOTES) {
    constexpr int kVoteLanes = kTile / 8;
    if (tid < kVoteLanes) {  // the vote words are not staged: 2 (4) bytes per group, one 16 (32) byte load per lane
      const uint64_t g = tile0 + 8ull * tid;
      const uint8_t* src = a.votes + g * vote_word_bytes(N);
      uint32_t words[8];
      if constexpr (N <= 8) {
        const u32x4p v = *reinterpret_cast<const u32x4p*>(src);
        const uint32_t pr[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) words[k] = (k & 1) ? pr[k >> 1] >> 16 : pr[k >> 1] & 0xffffu;
      } else {
        const u32x4p v0 = *reinterpret_cast<const u32x4p*>(src), v1 = *reinterpret_cast<const u32x4p*>(src + 16);
        words[0] = v0.x; words[1] = v0.y; words[2] = v0.z; words[3] = v0.w;
        words[4] = v1.x; words[5] = v1.y; words[6] = v1.z; words[7] = v1.w;
      }
      uint32_t out = 0, n_won = 0, n_lost = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t oc = poll_word<N>(words[k]);
        out |= oc << (2 * k);
        n_won += oc & 1u;
        n_lost += oc >> 1;
      }
      *reinterpret_cast<uint16_t*>(a.outcome + (g >> 2)) = (uint16_t)out;
      won_lost = n_won | (n_lost << 16);
    }
  }
  const uint32_t wl = VOTES ? wave_sum_u32(won_lost) : 0u;
  if (lane == 0) {
    uint4 r;
    r.x = n_changed;
    r.y = wl & 0xffffu;
    r.z = wl >> 16;
    r.w = 0;
    a.partials[(uint64_t)blockIdx.x * kWaves + wave] = r;
  }
}

// ---------------------------------------------------------------------------
// Sparse ingest.  Deltas arrive as the C-ABI's AoS structs in
// pinned, device-mapped host memory; the kernels read them straight over PCIe
// (coalesced: consecutive lanes, consecutive 24 / 16 byte structs).
struct DeltaRec {      // == raftq_delta_t
  uint64_t group, match;
  uint32_t peer, pad;
};
struct Delta16Rec {    // == raftq_delta16_t: the same update in 16 bytes for handles of fewer than 2^32 groups --
  uint64_t match;      // a third less PCIe traffic on the ingest side (the ingest is PCIe-bound)
  uint32_t group, peer;
};
struct VoteDeltaRec {  // == raftq_vote_delta_t
  uint64_t group;
  uint32_t peer;
  uint8_t vote, pad[3];
};

// A batch's verdict is kept per kind in two device words (epoch of the last bad batch).  The scatter kernels of
// ONE call check the words of every kind that call brought (kNoEpoch = that kind is not part of the call), so a
// bad vote record also withholds the match deltas that came with it: a cycle applies everything or nothing.
constexpr unsigned long long kNoEpoch = ~0ull;
__device__ __forceinline__ bool batch_is_bad(const unsigned long long* bad, unsigned long long epoch_match,
                                             unsigned long long epoch_votes) {
  return bad[0] == epoch_match || bad[1] == epoch_votes;
}

// Progress.maybeUpdate only ever raises Match, so a batch of MsgAppResp
// deltas is an order-independent atomic max.
// Ingest, pass 1: the records sit in pinned, device-mapped host memory (the caller's batch buffer); one coalesced
// read over PCIe brings them into HBM and range-checks them on the way.  A bad record stamps this batch's epoch
// into *bad_epoch (device) and *bad_host (mapped host word): pass 2 then applies nothing, and the host reports
// RAFTQ_EINVAL after its sync -- the all-or-nothing rule of the ABI without a 65K-iteration host loop (36 us).
template <typename Rec>
static __global__ __launch_bounds__(kBlock) void deltas_in_kernel(const Rec* __restrict__ src, Rec* __restrict__ dst,
                                                                  uint64_t n, uint64_t n_groups, uint32_t n_peers,
                                                                  unsigned long long* bad_epoch, uint64_t* bad_host,
                                                                  unsigned long long epoch) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const Rec r = src[i];
    dst[i] = r;
    bad = r.group >= n_groups || r.peer >= n_peers;
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) {
    atomicMax(bad_epoch, epoch);
    *bad_host = epoch;
  }
}

// RAFTQ_CYCLE_TRUSTED: the caller vouches for the ranges (a driver that built the records itself), so validation
// and scatter are ONE pass straight from the pinned batch -- no HBM copy, no second launch.  A record that is out
// of range after all is dropped on its own and reported; the others are applied.
template <typename Rec>
static __global__ __launch_bounds__(kBlock) void deltas_in_apply_kernel(const Rec* __restrict__ src, uint64_t n, uint64_t* match,
                                                                        uint64_t ld, uint64_t n_groups, uint32_t n_peers,
                                                                        unsigned long long* bad_epoch, uint64_t* bad_host,
                                                                        unsigned long long epoch) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const Rec r = src[i];
    bad = r.group >= n_groups || r.peer >= n_peers;
    if (!bad)
      atomicMax(reinterpret_cast<unsigned long long*>(match + (uint64_t)r.peer * ld + r.group), (unsigned long long)r.match);
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) {
    atomicMax(bad_epoch, epoch);
    *bad_host = epoch;
  }
}

template <typename Rec>
static __global__ __launch_bounds__(kBlock) void apply_deltas_kernel(uint64_t* match, uint64_t ld,
                                                              const Rec* __restrict__ d, uint64_t n,
                                                              const unsigned long long* bad,
                                                              unsigned long long epoch_match, unsigned long long epoch_votes) {
  if (batch_is_bad(bad, epoch_match, epoch_votes)) return;  // a record of this call is out of range: nothing is applied
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const Rec r = d[i];
  atomicMax(reinterpret_cast<unsigned long long*>(match + (uint64_t)r.peer * ld + r.group),
            (unsigned long long)r.match);
}

// Sparse term update: a group's leader changed term (became leader / appended
// the first entry of its term).  first_idx already has the cur_term == 0 rule
// folded in by the host.  Records are unique per group within a batch.
struct TermDeltaRec {  // == raftq_term_delta_t
  uint64_t group, cur_term, first_idx_cur_term;
};
static __global__ __launch_bounds__(kBlock) void apply_term_deltas_kernel(uint64_t* first_idx,
                                                                   const TermDeltaRec* __restrict__ d, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const TermDeltaRec r = d[i];
  first_idx[r.group] = r.cur_term == 0 ? 0ull : r.first_idx_cur_term;
}

// poll(): the first response of a peer wins -- also inside one batch, where
// "first" means lowest batch position.  Two launches make that deterministic
// without any host-side hashing: (1) every record claims its slot with
// atomicMin(batch position); (2) only the claim holder writes the vote (if the
// slot is still unanswered) and then releases the claim for the next batch.
// claim[] is u32 [N][ld], UINT32_MAX when free.
static __global__ __launch_bounds__(kBlock) void vote_deltas_in_kernel(const VoteDeltaRec* __restrict__ src,
                                                                       VoteDeltaRec* __restrict__ dst, uint64_t n,
                                                                       uint64_t n_groups, uint32_t n_peers,
                                                                       unsigned long long* bad_epoch, uint64_t* bad_host,
                                                                       unsigned long long epoch) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const VoteDeltaRec r = src[i];
    dst[i] = r;
    bad = r.group >= n_groups || r.peer >= n_peers || (unsigned)(r.vote - 1) > 1u;
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) {
    atomicMax(bad_epoch, epoch);
    *bad_host = epoch;
  }
}

// `each` (RAFTQ_CYCLE_TRUSTED): no batch verdict -- a record that is out of range is skipped on its own, the others apply
__device__ __forceinline__ bool vote_rec_ok(const VoteDeltaRec& r, uint64_t n_groups, uint32_t n_peers) {
  return r.group < n_groups && r.peer < n_peers && (unsigned)(r.vote - 1) <= 1u;
}

static __global__ __launch_bounds__(kBlock) void vote_claim_kernel(uint32_t* claim, uint64_t ld,
                                                            const VoteDeltaRec* __restrict__ d, uint64_t n,
                                                            const unsigned long long* bad, unsigned long long epoch_match,
                                                            unsigned long long epoch_votes, uint64_t n_groups,
                                                            uint32_t n_peers, int each) {
  if (!each && batch_is_bad(bad, epoch_match, epoch_votes)) return;
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const VoteDeltaRec r = d[i];
  if (each && !vote_rec_ok(r, n_groups, n_peers)) return;
  atomicMin(claim + (uint64_t)r.peer * ld + r.group, (uint32_t)i);
}

static __global__ __launch_bounds__(kBlock) void vote_apply_kernel(uint8_t* votes, int wide, uint32_t* claim, uint64_t ld,
                                                            const VoteDeltaRec* __restrict__ d, uint64_t n,
                                                            const unsigned long long* bad, unsigned long long epoch_match,
                                                            unsigned long long epoch_votes, uint64_t n_groups,
                                                            uint32_t n_peers, int each) {
  if (!each && batch_is_bad(bad, epoch_match, epoch_votes)) return;
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const VoteDeltaRec r = d[i];
  if (each && !vote_rec_ok(r, n_groups, n_peers)) return;
  const uint64_t slot = (uint64_t)r.peer * ld + r.group;
  if (claim[slot] != (uint32_t)i) return;  // an earlier record of this batch owns the slot
  // the group's vote word: 2 bits per peer.  Other peers' fields of the same word -- and, with 16-bit words, the
  // neighbouring group's word -- may be written by other lanes: CAS the field into the enclosing 32-bit word
  unsigned int* word = reinterpret_cast<unsigned int*>(votes) + (wide ? r.group : r.group >> 1);
  const unsigned sh = (wide ? 0u : (unsigned)(r.group & 1ull) * 16u) + 2u * r.peer;
  const unsigned nv = (unsigned)r.vote << sh;
  unsigned old = *word;
  while (true) {
    const unsigned cur = (old >> sh) & 3u;
    if (cur == 1u || cur == 2u) break;  // answered in an earlier batch
    const unsigned seen = atomicCAS(word, old, (old & ~(3u << sh)) | nv);
    if (seen == old) break;
    old = seen;
  }
  claim[slot] = 0xffffffffu;  // only the holder releases; later readers see "not mine" either way
}

// ---------------------------------------------------------------------------
// Changed-group compaction (the Ready side): turn the
// lane-ordered bitmap of a RAFTQ_SWEEP_CHANGED sweep into a dense, ascending
// list of {group, old_commit, new_commit}.  Deterministic two-pass: the sweep
// already left per-wave change counts in partials[].x; scan_partials_kernel
// turns them into exclusive offsets (one workgroup, the array is small), then
// every wave of compact_changed_kernel ranks its own bits with popcounts.
struct Advance {       // == raftq_advance_t
  uint64_t group, old_commit, new_commit;
};
struct Advance16 {     // == raftq_advance16_t: 16 bytes back over PCIe instead of 24 (handles of < 2^32 groups);
  uint64_t new_commit; // old_commit = new_commit - advanced_by, exact unless advanced_by saturated at 2^32 - 1
  uint32_t group, advanced_by;
};
__device__ __forceinline__ Advance make_advance(Advance*, uint64_t g, uint64_t o, uint64_t n) { return Advance{g, o, n}; }
__device__ __forceinline__ Advance16 make_advance(Advance16*, uint64_t g, uint64_t o, uint64_t n) {
  const uint64_t by = n - o;
  return Advance16{n, (uint32_t)g, by > 0xfffffffeull ? 0xffffffffu : (uint32_t)by};
}

// One workgroup, one pass over memory for the common sizes: thread t owns the contiguous chunk
// [t*per, (t+1)*per) of the per-wave counts (per = ceil(n/1024): 4 for 1M groups at 256-group waves), sums it
// with every load issued up front, the 1024 chunk sums are scanned by wave shuffles + one LDS hop, and the chunk's
// exclusive offsets are written from registers (chunks up to 8 items; longer ones are re-read).  `field` picks
// the counter: 0 = .x (changed groups / MsgHup), 1 = .y (MsgBeat).  7.2 -> ~3 us for 4096 counts: the old form
// walked the array in four dependent load -> scan -> barrier rounds.
static __global__ __launch_bounds__(1024) void scan_partials_kernel(const uint4* __restrict__ partials, uint64_t n_waves,
                                                             uint64_t* __restrict__ offsets, uint64_t* total, int field) {
  __shared__ uint64_t wave_tot[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint64_t per = (n_waves + 1023) / 1024;
  const uint64_t lo = (uint64_t)tid * per, hi = lo + per < n_waves ? lo + per : n_waves;
  const uint32_t* p32 = reinterpret_cast<const uint32_t*>(partials) + field;
  constexpr int kKeep = 8;
  uint32_t keep[kKeep];
  uint64_t sum = 0;
  if (per <= kKeep) {
#pragma unroll
    for (int k = 0; k < kKeep; ++k) keep[k] = (uint64_t)k < per && lo + k < hi ? p32[(lo + k) * 4] : 0u;
#pragma unroll
    for (int k = 0; k < kKeep; ++k) sum += keep[k];
  } else {
    for (uint64_t i = lo; i < hi; ++i) sum += p32[i * 4];
  }
  uint64_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = __shfl_up(incl, o, 64);
    if (lane >= (uint32_t)o) incl += y;
  }
  if (lane == 63) wave_tot[w] = incl;
  __syncthreads();
  uint64_t pre = 0;
  for (uint32_t k = 0; k < w; ++k) pre += wave_tot[k];
  uint64_t run = pre + incl - sum;  // exclusive offset of this thread's chunk
  if (per <= kKeep) {
#pragma unroll
    for (int k = 0; k < kKeep; ++k)
      if ((uint64_t)k < per && lo + k < hi) {
        offsets[lo + k] = run;
        run += keep[k];
      }
  } else {
    for (uint64_t i = lo; i < hi; ++i) {
      offsets[i] = run;
      run += p32[i * 4];
    }
  }
  if (tid == 1023) *total = pre + incl;
}

// Mirrors the sweep's geometry: wave `wv` of block `b` owns the 64*GPL
// consecutive groups from b*kTile + wv*64*GPL and produced, for each round j,
// one {even, odd} word pair for the 128 groups at + j*128 -- so (block, wave,
// round, lane, parity) order IS ascending group order.
// No separate scan launch: a workgroup computes the exclusive offset of its first wave itself, as the sum of the
// per-wave counts of every wave before it (<= 8192 L2-resident words, read by all 256 threads with the loads
// issued eight at a time), which is cheaper than the 5 us single-workgroup scan kernel plus the launch boundary it
// replaces.  The last workgroup also publishes the total.  (A first form had every workgroup
// __threadfence_system() and count itself in so that the last one could raise a completion flag for the host:
// the fence writes back and invalidates the L2 -- 127 us instead of 13.  The flag is now a stream write-value
// packet behind the kernel, raftq_capi.hip wait_turn.)
// Round 3 tried the other shape VERDICT r02 item 4 names -- 256 workgroups of 16 sweep-waves each, every bitmap word
// loaded at entry, the old / new pairs of all rounds gathered in one batch, the records staged in LDS and written out as
// one contiguous run of lane-consecutive 16-byte stores -- and measured it SLOWER: 14.5 us against this kernel's 12.7 for
// the same 350 KB (profiles/r03/compaction_shapes.txt), with the LDS stage on or off and with the list in explicitly
// fine-grained or runtime-default pinned memory alike.  The list does not leave in small pieces because PCIe is short of
// them: 350 KB at the 53 GB/s a kernel reaches into host memory (step_d2h_kernel) is 6.6 us, behind ~2 us of launch and
// two dependent memory latencies; what this shape has and the wide one lacks is 1,024 small workgroups that reach their
// stores at different times, so the drain overlaps the tail of the work.  The same round tried to let the compaction's
// last workgroup raise the host's completion flag (every workgroup: s_waitcnt vmcnt(0), then a device-scope counter):
// the flag overtook the data in 8 of 15 tests under RAFTQ_CYCLE_CHECK -- a store's acknowledgement is not system-wide
// visibility, and stores of different XCDs take different ways out.  Only a kernel boundary (or the system-scope fence
// that cost 127 us in round 2) orders them: the flag is now a one-thread kernel of ours behind this one
// (raftq_capi.hip enqueue_collect) instead of the runtime's 3.4 us write-value kernel 5 us behind it.
template <int GPL, typename Adv>
static __global__ __launch_bounds__(kBlock) void compact_changed_kernel(const uint64_t* changed_bits,
                                                                 const uint4* __restrict__ partials,
                                                                 const uint64_t* old_commit,
                                                                 const uint64_t* new_commit,
                                                                 Adv* out, uint64_t cap, uint64_t* total,
                                                                 const uint64_t* __restrict__ wave_offsets) {
  constexpr int kTile = kBlock * GPL;
  constexpr int kRounds = GPL / 2;
  __shared__ uint64_t red[kWaves];
  __shared__ uint32_t mine[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // exclusive prefix of the change counts of all waves of earlier workgroups.  Summing them here costs
  // O(workgroups^2) loads over the launch, fine up to a few million groups per handle; the host hands in
  // scan_partials_kernel's offsets instead when the handle is larger (wave_offsets != nullptr).
  const uint64_t first_wave = (uint64_t)blockIdx.x * kWaves;
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(partials);  // .x of entry i = word 4 i
  uint64_t acc = 0;
  if (wave_offsets == nullptr) {
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint64_t i = tid;
    for (; i + 3 * kBlock < first_wave; i += 4 * kBlock) {
      a0 += cnt[4 * i];
      a1 += cnt[4 * (i + kBlock)];
      a2 += cnt[4 * (i + 2 * kBlock)];
      a3 += cnt[4 * (i + 3 * kBlock)];
    }
    for (; i < first_wave; i += kBlock) a0 += cnt[4 * i];
    acc = (uint64_t)a0 + a1 + a2 + a3;  // per-thread partial sums stay below 2^32 on this path (< 2^24 waves, <= 512 each)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  } else if (lane == 0 && wave == 0) {
    acc = wave_offsets[first_wave];
  }
  if (lane == 0) red[wave] = acc;
  if (tid < kWaves) mine[tid] = cnt[4 * (first_wave + tid)];
  __syncthreads();
  uint64_t pos = 0;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) pos += red[k];
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    uint64_t t = pos;
    for (int k = 0; k < kWaves; ++k) t += mine[k];
    *total = t;
  }
  for (uint32_t k = 0; k < wave; ++k) pos += mine[k];
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint64_t g0 = (uint64_t)blockIdx.x * kTile + (uint64_t)wave * (64 * GPL) + (uint64_t)j * 128;
    const uint64_t b0 = changed_bits[(g0 >> 6)];
    const uint64_t b1 = changed_bits[(g0 >> 6) + 1];
    const bool e = (b0 >> lane) & 1, o = (b1 >> lane) & 1;
    const uint64_t rank = __popcll(b0 & below) + __popcll(b1 & below);
    if (e) {
      const uint64_t g = g0 + 2 * lane, s = pos + rank;
      if (s < cap) out[s] = make_advance((Adv*)nullptr, g, old_commit[g], new_commit[g]);
    }
    if (o) {
      const uint64_t g = g0 + 2 * lane + 1, s = pos + rank + (e ? 1 : 0);
      if (s < cap) out[s] = make_advance((Adv*)nullptr, g, old_commit[g], new_commit[g]);
    }
    pos += __popcll(b0) + __popcll(b1);
  }
}

// The host's completion flag for a turn: one thread, one store into pinned host memory.  It runs behind the compaction on
// the handle's stream, so the kernel boundary in front of it has made every store of the turn visible to the host.
static __global__ void raise_flag_kernel(uint64_t* flag, uint64_t epoch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// The batching turn's sweep writing its own advance list, in SEGMENTS (RAFTQ_CYCLE_SEGMENTED; raft.go:227-235).
// Round 3's turn ran the sweep (9.5 us), then -- a kernel boundary later -- the compaction above (13.2 us, of which the
// 350 KB of records on their way out are ~7), then the flag.  A workgroup that has decided its tile holds everything a
// record needs in registers (old and new commit index of its changed groups, the ballots); written from there the records
// leave while the rest of the sweep is still streaming and cost the kernel nothing (15 us for sweep + list against 22.7).
// What they cannot have for free is ONE contiguous list: a tile's position in it is the number of changed groups of every
// tile before it, and while the sweep saturates HBM an agent-scope round trip between XCDs is 3-5 us -- learning it cost
// 13-20 us however it was arranged (profiles/r04/sweep_emit_experiment.md).  So tile t writes its records, ascending, at
// list[t * tile groups ...] -- a tile cannot have more changed groups than groups -- and its count at counts[t]: the list
// is ascending when its segments are walked in order (raftq_last_advance_segments).  The changed bitmap and the per-wave
// counts are written as by sweep_kernel: raftq_collect_changed still works afterwards.
// VERDICT r05 item 6, candidate (b): the completion word WITHOUT a kernel of its own.  Every workgroup makes its stores -- records
// and counts in page-locked host memory among them -- performed system-wide (`__threadfence_system()` by every thread, then the
// workgroup's barrier) BEFORE it counts itself in (one agent-scope atomic); the workgroup that finds itself last adds up what the
// others left (agent-scope loads: they may sit in another XCD's L2) and stores the word with system-scope release, and hands the
// counter back zero.  Round 3's attempt had the compaction's last workgroup raise the flag with NO fence in the other workgroups:
// the word overtook the data in 8 of 15 tests.  RAFTQ_CYCLE_FLAG=arrive selects this form; profiles/r06/flag_ab.txt has what
// RAFTQ_CYCLE_CHECK made of it over 16,000 turns and what it costs beside the one-thread kernel and the runtime's write-value packet.
struct Arrival {
  unsigned long long* flag;  // the completion word as the device addresses it, or nullptr: the flag is somebody else's (a kernel behind this one)
  unsigned int* count;       // device word, zero between kernels
  uint32_t epoch;
};
// ALL threads of the workgroup call it, behind their last store.  -> true in every thread of the LAST workgroup to arrive.
__device__ __forceinline__ bool arrive_last(const Arrival& ar, uint32_t* slot /*LDS*/) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int seen = __hip_atomic_fetch_add(ar.count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    *slot = seen == gridDim.x * gridDim.y - 1 ? 1u : 0u;
  }
  __syncthreads();
  return *slot != 0;
}

template <int N, int GPL, bool GATED, int POLICY>
static __global__ __launch_bounds__(kBlock) void sweep_segments_kernel(SweepArgs a, Advance16* list, unsigned int* counts_host, unsigned int* counts_dev,
                                                                       Arrival ar) {
  constexpr bool STNT = (POLICY & kStNT) != 0;
  constexpr int kTile = kBlock * GPL;
  constexpr int kRounds = GPL / 2;
  __shared__ uint32_t wave_cnt[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
  TileRegs<N, GPL, true, GATED, false> r;
  tile_load<N, GPL, true, GATED, false, POLICY>(r, a, tile);
  const uint64_t tile0 = (uint64_t)tile * kTile;
  u64x2 nw[kRounds];
  uint64_t even[kRounds], odd[kRounds];  // wave-uniform ballots
  uint32_t n_changed = 0;
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint64_t g = tile0 + (uint64_t)wave * (64 * GPL) + (uint64_t)j * 128 + 2 * lane;
    uint64_t v0[N], v1[N];
#pragma unroll
    for (int p = 0; p < N; ++p) {
      v0[p] = r.m[j][p].x;
      v1[p] = r.m[j][p].y;
    }
    const uint64_t mci0 = N == 1 ? v0[0] : select_quorum_network<N>(v0);
    const uint64_t mci1 = N == 1 ? v1[0] : select_quorum_network<N>(v1);
    u64x2 o;
    o.x = maybe_commit<GATED>(mci0, r.c[j].x, GATED ? r.f[j].x : 0);
    o.y = maybe_commit<GATED>(mci1, r.c[j].y, GATED ? r.f[j].y : 0);
    even[j] = __ballot(o.x != r.c[j].x);
    odd[j] = __ballot(o.y != r.c[j].y);
    n_changed += __popcll(even[j]) + __popcll(odd[j]);
    nw[j] = o;
    if (a.changed_bits != nullptr && lane == 0) {
      u64x2 w;
      w.x = even[j];
      w.y = odd[j];
      stg<false>(reinterpret_cast<u64x2*>(a.changed_bits + (g >> 6)), w);
    }
    stg<STNT>(reinterpret_cast<u64x2*>(a.committed_out + g), o);
  }
  if (lane == 0) {
    uint4 t;
    t.x = n_changed;
    t.y = t.z = t.w = 0;
    stg_u4(a.partials + ((uint64_t)tile * kWaves + wave), t);
    wave_cnt[wave] = n_changed;
  }
  __syncthreads();
  uint64_t pos = tile0;  // the tile's segment
  for (uint32_t k = 0; k < wave; ++k) pos += wave_cnt[k];
  if (tid == 0) {
    uint32_t tile_cnt = 0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) tile_cnt += wave_cnt[k];
    counts_host[tile] = tile_cnt;
    counts_dev[tile] = tile_cnt;
  }
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {
    const uint64_t g0 = tile0 + (uint64_t)wave * (64 * GPL) + (uint64_t)j * 128;
    const bool e = (even[j] >> lane) & 1, o = (odd[j] >> lane) & 1;
    const uint64_t rank = __popcll(even[j] & below) + __popcll(odd[j] & below);
    if (e) list[pos + rank] = make_advance((Advance16*)nullptr, g0 + 2 * lane, r.c[j].x, nw[j].x);
    if (o) list[pos + rank + (e ? 1 : 0)] = make_advance((Advance16*)nullptr, g0 + 2 * lane + 1, r.c[j].y, nw[j].y);
    pos += __popcll(even[j]) + __popcll(odd[j]);
  }
  if (ar.flag != nullptr) {  // RAFTQ_CYCLE_FLAG=arrive: the last workgroup to arrive is the flag kernel
    __shared__ uint32_t last_slot;
    __shared__ uint32_t red[kWaves];
    if (!arrive_last(ar, &last_slot)) return;
    uint32_t acc = 0;
    for (uint32_t i = tid; i < gridDim.x; i += kBlock) acc += __hip_atomic_load(counts_dev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      uint32_t total = 0;
#pragma unroll
      for (int k = 0; k < kWaves; ++k) total += red[k];
      __hip_atomic_store(ar.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ar.flag, ((unsigned long long)ar.epoch << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The segmented turn's flag: one workgroup behind the sweep (the kernel boundary has made every record and count visible to
// the host) adds up the segments' counts and stores ONE word: epoch << 32 | total.
static __global__ __launch_bounds__(kBlock) void raise_flag_segments_kernel(uint64_t* flag, uint32_t epoch, const unsigned int* counts_dev, uint32_t n_tiles) {
  __shared__ uint32_t red[kWaves];
  uint32_t acc = 0;
  for (uint32_t i = threadIdx.x; i < n_tiles; i += kBlock) acc += counts_dev[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) total += red[k];
    __hip_atomic_store(flag, ((uint64_t)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// Batched Tick: rc.node.Tick() (raft.go:223-224) for every
// group at once -- etcd raft.tickElection / tickHeartbeat / isElectionTimeout.
// A lane owns 4 consecutive groups: one 16-byte load/store of `elapsed`, one
// 4-byte load of `role`, one 4-byte store of `action`.  10 B per group: HBM /
// launch bound.  MsgHup groups are flagged in a lane-ordered bitmap (word k of
// a wave's 4 = its groups 4*lane + k) that compact_hups_kernel turns into an
// ascending list, with the same scan as the commit compaction.
struct TickArgs {
  const uint8_t* role;   // 0 follower, 1 candidate, 2 leader
  uint32_t* elapsed;     // in/out
  uint8_t* action;       // 0 none, 1 MsgHup, 2 MsgBeat
  uint64_t* hup_bits;    // [gpad/64]
  uint64_t* beat_bits;   // [gpad/64] same layout, MsgBeat groups
  uint4* partials;       // [gpad/256] {hup, beat, 0, 0} per wave
  uint64_t n_groups;     // padding groups never act
  uint64_t seed, tick_no;
  uint32_t election_tick, heartbeat_tick;
  uint32_t et_magic;     // floor(2^32 / election_tick): x % election_tick without a division (tick_mod)
};

// The timeout draw (the stream is this repo's own definition -- Go's math/rand cannot be matched; include/raftq.h
// "batched Tick" states it).  tick_key: one splitmix64 finaliser of (seed, tick number) -- wave-uniform, the scalar unit
// computes it once per kernel.  tick_rand: murmur3's 32-bit finaliser of (group ^ key.lo), xored with key.hi: two
// v_mul_lo_u32 per group where round 5's splitmix64 of (seed, tick, group) cost three 64-bit multiplies (a dozen
// quarter-rate instructions) -- with every follower past its base timeout the Tick was VALU-bound at 0.58 of HBM.
__device__ __forceinline__ uint64_t tick_key(uint64_t seed, uint64_t tick_no) {
  uint64_t z = seed ^ (tick_no * 0xD1B54A32D192ED03ull);
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t tick_rand(uint64_t key, uint32_t group /* a handle holds <= 2^30 groups: the oracle's fold of the high word is a no-op */) {
  uint32_t x = group ^ (uint32_t)key;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x ^ (uint32_t)(key >> 32);
}

// x % d with magic = floor(2^32 / d): the quotient estimate is at most 2 short
__device__ __forceinline__ uint32_t tick_mod(uint32_t x, uint32_t d, uint32_t magic) {
  uint32_t r = x - __umulhi(x, magic) * d;
  r = r >= d ? r - d : r;
  return r >= d ? r - d : r;
}

// v_writelane_b32: a wave-uniform value into lane L of a vector register (this toolchain has no clang builtin for it)
template <int L>
__device__ __forceinline__ uint32_t write_lane(uint32_t v, uint32_t x /*wave-uniform*/) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(x), "n"(L));
  return v;
}

// What a wave leaves for one 256-group chunk beside the elapsed / action bytes: its four MsgHup and four MsgBeat bitmap words
// and its {n_hup, n_beat, 0, 0} counts -- 20 dwords, all wave-uniform (ballots and popcounts live in SGPRs).  They are moved
// into lanes 0..19 of ONE vector register with v_writelane and leave in ONE masked store instruction.  Round 4 had lane 0
// store them under `if (lane == 0)`: five flat stores in a branch, behind which the compiler can only wait with vmcnt(0) --
// i.e. for the round's elapsed store to come back from memory -- before it touches the next round's registers.
__device__ __forceinline__ void tick_chunk_out(const TickArgs& a, uint64_t chunk /* blk * kWaves + c */, uint32_t lane, const uint64_t (&hb)[4],
                                               const uint64_t (&bb)[4], uint32_t n_hup, uint32_t n_beat) {
  uint32_t v = 0;
  v = write_lane<0>(v, (uint32_t)hb[0]);  v = write_lane<1>(v, (uint32_t)(hb[0] >> 32));
  v = write_lane<2>(v, (uint32_t)hb[1]);  v = write_lane<3>(v, (uint32_t)(hb[1] >> 32));
  v = write_lane<4>(v, (uint32_t)hb[2]);  v = write_lane<5>(v, (uint32_t)(hb[2] >> 32));
  v = write_lane<6>(v, (uint32_t)hb[3]);  v = write_lane<7>(v, (uint32_t)(hb[3] >> 32));
  v = write_lane<8>(v, (uint32_t)bb[0]);  v = write_lane<9>(v, (uint32_t)(bb[0] >> 32));
  v = write_lane<10>(v, (uint32_t)bb[1]); v = write_lane<11>(v, (uint32_t)(bb[1] >> 32));
  v = write_lane<12>(v, (uint32_t)bb[2]); v = write_lane<13>(v, (uint32_t)(bb[2] >> 32));
  v = write_lane<14>(v, (uint32_t)bb[3]); v = write_lane<15>(v, (uint32_t)(bb[3] >> 32));
  v = write_lane<16>(v, n_hup);
  v = write_lane<17>(v, n_beat);
  // (addresses as integers: selects, not branches)
  const uint64_t at_hup = (uint64_t)(uintptr_t)a.hup_bits + chunk * 32 + 4ull * lane;
  const uint64_t at_beat = (uint64_t)(uintptr_t)a.beat_bits + chunk * 32 + 4ull * lane - 32;
  const uint64_t at_part = (uint64_t)(uintptr_t)a.partials + chunk * 16 + 4ull * lane - 64;
  const uint64_t at = lane < 8 ? at_hup : (lane < 16 ? at_beat : at_part);
  if (lane < 20) stg<false>(reinterpret_cast<uint32_t*>((uintptr_t)at), v);
}

// R consecutive 1,024-group blocks per workgroup (the set dispatch: every load of the R rounds issued before the first
// compare, a quarter of the workgroups); block = what one workgroup of the R = 1 kernel owns, so bitmaps, action bytes
// and per-wave counts have one layout whatever R is.  n_blocks: 1,024-group blocks of the handle (gpad / 1024).
template <int R, bool NT = false>
__device__ __forceinline__ void tick_tile(const TickArgs& a, uint64_t n_blocks) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t key = tick_key(a.seed, a.tick_no);  // wave-uniform
  uint32_t roles[R];
  uint4 el[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t blk = (uint64_t)blockIdx.x * R + r;
    if (blk < n_blocks) {
      const uint64_t g = (blk * kBlock + tid) * 4;
      roles[r] = ldg<NT>(reinterpret_cast<const uint32_t*>(a.role + g));
      const u32x4 v = ldg<NT>(reinterpret_cast<const u32x4*>(a.elapsed + g));
      el[r].x = v.x; el[r].y = v.y; el[r].z = v.z; el[r].w = v.w;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t blk = (uint64_t)blockIdx.x * R + r;
    if (blk >= n_blocks) break;  // workgroup-uniform
    const uint64_t g = (blk * kBlock + tid) * 4;
    uint32_t e[4] = {el[r].x, el[r].y, el[r].z, el[r].w};
    uint32_t acts = 0;
    uint32_t n_hup = 0, n_beat = 0;  // wave-uniform
    uint64_t hb[4], bb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t role = (roles[r] >> (8 * k)) & 0xffu;
      const bool valid = g + k < a.n_groups;
      uint32_t v = e[k] + 1;
      const bool beat = role == 2u && v >= a.heartbeat_tick;
      const int64_t d = (int64_t)v - (int64_t)a.election_tick;
      // the draw (three 64-bit multiplies and a 32-bit modulo per group) only where a timer is past its base timeout: the
      // Tick was VALU-bound with it computed for every group -- 84 MB in 22 us on a chip that streams them in 11
      bool hup = role != 2u && d >= 0;
      if (__ballot(hup) != 0) hup = hup && d > (int64_t)tick_mod(tick_rand(key, (uint32_t)g + k), a.election_tick, a.et_magic);
      const uint32_t act = !valid ? 0u : (hup ? 1u : (beat ? 2u : 0u));
      e[k] = !valid ? e[k] : (act ? 0u : v);
      acts |= act << (8 * k);
      hb[k] = __ballot(act == 1u);
      bb[k] = __ballot(act == 2u);
      n_hup += __popcll(hb[k]);
      n_beat += __popcll(bb[k]);
    }
    u32x4 out;
    out.x = e[0]; out.y = e[1]; out.z = e[2]; out.w = e[3];
    stg<NT>(reinterpret_cast<u32x4*>(a.elapsed + g), out);
    stg<NT>(reinterpret_cast<uint32_t*>(a.action + g), acts);
    tick_chunk_out(a, blk * kWaves + wave, lane, hb, bb, n_hup, n_beat);
  }
}
static __global__ __launch_bounds__(kBlock) void tick_kernel(TickArgs a) { tick_tile<1>(a, gridDim.x); }

// The set dispatch at the sweep's access shape: a WAVE owns whole 1,024-group blocks -- the four
// 256-group chunks tick_tile gives to four waves -- so that every global access of a lane is 16 bytes and every wave
// instruction 1 KB: ONE 16-byte role load and ONE 16-byte action store per lane for the block's 1,024 groups (16 groups per
// lane) beside the four 16-byte elapsed loads / stores, where tick_tile issues four 4-byte ones (256 B per wave instruction).
// A lane's role quad covers groups 16l .. 16l+15; the chunk it ticks in round c is groups c*256 + 4l .. +3: the block's
// 1 KB of role bytes goes through a wave-private LDS kilobyte (one ds_write_b128, four conflict-free ds_read_b32) and the
// action bytes come back the same way.  Chunk c of block blk leaves exactly what wave c of tick_tile's workgroup leaves:
// bitmaps, per-wave counts and action bytes keep ONE layout (tick_lists_kernel, compact_hups_kernel and the oracle
// comparison read either).  R blocks per wave, every load of the R blocks issued before the first compare.
template <int R, bool NT>
__device__ __forceinline__ void tick_tile_wide(const TickArgs& a, uint64_t n_blocks) {
  __shared__ uint32_t xpose[kWaves][256];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t key = tick_key(a.seed, a.tick_no);  // wave-uniform
  uint32_t* const xp = xpose[wave];
  u32x4 rq[R];
  u32x4 el[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // (a block past the end loads the last block again and is never processed: unconditional loads keep the compiler's count of
    // what is in flight exact -- behind a branch it has to assume the fewest and waits for nearly everything at the first use)
    const uint64_t blk_r = ((uint64_t)blockIdx.x * kWaves + wave) * R + r;
    const uint64_t blk = blk_r < n_blocks ? blk_r : n_blocks - 1;
    const uint64_t g0 = blk * 1024;
    rq[r] = ldg<NT>(reinterpret_cast<const u32x4*>(a.role + g0 + 16 * lane));
#pragma unroll
    for (int c = 0; c < 4; ++c) el[r][c] = ldg<NT>(reinterpret_cast<const u32x4*>(a.elapsed + g0 + c * 256 + 4 * lane));
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t blk = ((uint64_t)blockIdx.x * kWaves + wave) * R + r;
    if (blk >= n_blocks) break;  // wave-uniform; blocks ascend with r
    const uint64_t g0 = blk * 1024;
    *reinterpret_cast<u32x4*>(xp + 4 * lane) = rq[r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t roles[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) roles[c] = xp[c * 64 + lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint64_t g = g0 + c * 256 + 4 * lane;
      uint32_t e[4] = {el[r][c].x, el[r][c].y, el[r][c].z, el[r][c].w};
      uint32_t acts = 0;
      uint32_t n_hup = 0, n_beat = 0;  // wave-uniform
      uint64_t hb[4], bb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t role = (roles[c] >> (8 * k)) & 0xffu;
        const bool valid = g + k < a.n_groups;
        uint32_t v = e[k] + 1;
        const bool beat = role == 2u && v >= a.heartbeat_tick;
        const int64_t d = (int64_t)v - (int64_t)a.election_tick;
        bool hup = role != 2u && d >= 0;
        if (__ballot(hup) != 0) hup = hup && d > (int64_t)tick_mod(tick_rand(key, (uint32_t)g + k), a.election_tick, a.et_magic);
        const uint32_t act = !valid ? 0u : (hup ? 1u : (beat ? 2u : 0u));
        e[k] = !valid ? e[k] : (act ? 0u : v);
        acts |= act << (8 * k);
        hb[k] = __ballot(act == 1u);
        bb[k] = __ballot(act == 2u);
        n_hup += __popcll(hb[k]);
        n_beat += __popcll(bb[k]);
      }
      u32x4 out;
      out.x = e[0]; out.y = e[1]; out.z = e[2]; out.w = e[3];
      stg<NT>(reinterpret_cast<u32x4*>(a.elapsed + g), out);
      xp[c * 64 + lane] = acts;
      tick_chunk_out(a, blk * kWaves + c, lane, hb, bb, n_hup, n_beat);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32x4 aq = *reinterpret_cast<const u32x4*>(xp + 4 * lane);
    stg<NT>(reinterpret_cast<u32x4*>(a.action + g0 + 16 * lane), aq);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}


static __global__ __launch_bounds__(kBlock) void compact_hups_kernel(const uint64_t* hup_bits, const uint64_t* offsets,
                                                              uint64_t* out, uint64_t cap) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t wv = (uint64_t)blockIdx.x * kWaves + wave;
  const uint64_t pos = offsets[wv];
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint64_t b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = hup_bits[wv * 4 + k];
  uint64_t rank = pos + __popcll(b[0] & below) + __popcll(b[1] & below) + __popcll(b[2] & below) + __popcll(b[3] & below);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if ((b[k] >> lane) & 1) {
      if (rank < cap) out[rank] = wv * 256 + 4ull * lane + k;
      ++rank;
    }
  }
}

// Both lists of a Tick in ONE launch, no scan launch in front of it: a workgroup owns the four waves (1,024 groups) of one
// tick_kernel workgroup and computes the exclusive offsets of its first wave itself, as the sums of the per-wave MsgHup /
// MsgBeat counts of every wave before it (L2-resident words, as compact_changed_kernel does for the advance list), then
// ranks its own bits.  Ascending group ids in both lists; the last workgroup publishes the two totals.  The host hands in
// scan_partials_kernel's offsets instead when the handle has more than 16K waves (wave_off_* != nullptr).
static __global__ __launch_bounds__(kBlock) void tick_lists_kernel(const uint64_t* __restrict__ hup_bits, const uint64_t* __restrict__ beat_bits,
                                                                    const uint4* __restrict__ partials, uint64_t* hup_out, uint64_t hup_cap,
                                                                    uint64_t* beat_out, uint64_t beat_cap, uint64_t* totals /*[2]*/,
                                                                    const uint64_t* __restrict__ wave_off_hup,
                                                                    const uint64_t* __restrict__ wave_off_beat) {
  __shared__ uint64_t red[2][kWaves];
  __shared__ uint32_t mine[2][kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t first_wave = (uint64_t)blockIdx.x * kWaves;
  uint64_t acc_h = 0, acc_b = 0;
  if (wave_off_hup == nullptr) {
    uint32_t h0 = 0, h1 = 0, b0 = 0, b1 = 0;
    uint64_t i = tid;
    for (; i + kBlock < first_wave; i += 2 * kBlock) {
      const uint4 p0 = partials[i], p1 = partials[i + kBlock];
      h0 += p0.x; b0 += p0.y;
      h1 += p1.x; b1 += p1.y;
    }
    if (i < first_wave) {
      const uint4 p0 = partials[i];
      h0 += p0.x; b0 += p0.y;
    }
    acc_h = (uint64_t)h0 + h1;
    acc_b = (uint64_t)b0 + b1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      acc_h += __shfl_xor(acc_h, o, 64);
      acc_b += __shfl_xor(acc_b, o, 64);
    }
  } else if (lane == 0 && wave == 0) {
    acc_h = wave_off_hup[first_wave];
    acc_b = wave_off_beat[first_wave];
  }
  if (lane == 0) {
    red[0][wave] = acc_h;
    red[1][wave] = acc_b;
  }
  if (tid < kWaves) {
    const uint4 p = partials[first_wave + tid];
    mine[0][tid] = p.x;
    mine[1][tid] = p.y;
  }
  __syncthreads();
  uint64_t pos_h = 0, pos_b = 0;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) {
    pos_h += red[0][k];
    pos_b += red[1][k];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    uint64_t th = pos_h, tb = pos_b;
    for (int k = 0; k < kWaves; ++k) {
      th += mine[0][k];
      tb += mine[1][k];
    }
    totals[0] = th;
    totals[1] = tb;
  }
  for (uint32_t k = 0; k < wave; ++k) {
    pos_h += mine[0][k];
    pos_b += mine[1][k];
  }
  const uint64_t wv = first_wave + wave;
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint64_t hb[4], bb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hb[k] = hup_bits[wv * 4 + k];
    bb[k] = beat_bits[wv * 4 + k];
  }
  uint64_t rh = pos_h + __popcll(hb[0] & below) + __popcll(hb[1] & below) + __popcll(hb[2] & below) + __popcll(hb[3] & below);
  uint64_t rb = pos_b + __popcll(bb[0] & below) + __popcll(bb[1] & below) + __popcll(bb[2] & below) + __popcll(bb[3] & below);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint64_t g = wv * 256 + 4ull * lane + k;
    if ((hb[k] >> lane) & 1) {
      if (rh < hup_cap) hup_out[rh] = g;
      ++rh;
    }
    if ((bb[k] >> lane) & 1) {
      if (rb < beat_cap) beat_out[rb] = g;
      ++rb;
    }
  }
}

// The same two lists as they are meant to be read: LEFT IN PLACE in page-locked memory, 4-byte group ids (a handle holds at
// most 2^30 groups), and -- BEAT_BITMAP -- the MsgBeat groups as a bitmap in GROUP order (bit g % 64 of word g / 64) instead
// of a list: with HeartbeatTick 1 (raft.go:155) every leader beats on every tick, so the beat "list" is the leader set --
// 128 KB as a bitmap against 1.4 MB of ids per 1M groups a third of which lead.  A workgroup compacts its 1,024 groups' ids
// in LDS and writes them out as runs of consecutive words (full lines over the link, not a lane's four words at a time).
// A workgroup's run of ids, LDS -> its place in a list in page-locked host memory: whole 16-byte quads of the DESTINATION (1 KB per
// wave instruction over the link), the few words before the first and behind the last quad one by one.
__device__ __forceinline__ void run_out(uint32_t* out, uint64_t cap, uint64_t pos, const uint32_t* ids /*LDS*/, uint32_t n, uint32_t tid) {
  if (pos >= cap) return;
  if (pos + n > cap) n = (uint32_t)(cap - pos);
  const uint32_t head = (uint32_t)((4 - (pos & 3)) & 3) < n ? (uint32_t)((4 - (pos & 3)) & 3) : n;  // words before the first whole quad
  const uint32_t quads = (n - head) >> 2;
  if (tid < head) out[pos + tid] = ids[tid];
  for (uint32_t q = tid; q < quads; q += kBlock) {
    const uint32_t i = head + 4 * q;
    u32x4 v;
    v.x = ids[i]; v.y = ids[i + 1]; v.z = ids[i + 2]; v.w = ids[i + 3];
    *reinterpret_cast<u32x4*>(out + pos + i) = v;
  }
  const uint32_t done = head + 4 * quads;
  if (tid < n - done) out[pos + done + tid] = ids[done + tid];
}

// BPW: 1,024-group blocks per workgroup.  A workgroup's ids leave as ONE run per list; longer runs (BPW = 4) were measured no
// faster than BPW = 1 (1M groups, 400K ids: 34 us of kernel either way -- 1.6 MB at the link's 47 GB/s for this pattern).
template <bool BEAT_BITMAP, int BPW>
static __global__ __launch_bounds__(kBlock) void tick_lists32_kernel(const uint64_t* __restrict__ hup_bits, const uint64_t* __restrict__ beat_bits,
                                                                      const uint4* __restrict__ partials, uint64_t n_chunks /* gpad / 256 */,
                                                                      uint32_t* hup_out, uint64_t hup_cap, uint32_t* beat_out, uint64_t beat_cap,
                                                                      uint64_t* beat_map, uint64_t* totals /*[2]*/,
                                                                      const uint64_t* __restrict__ wave_off_hup, const uint64_t* __restrict__ wave_off_beat,
                                                                      Arrival ar) {
  constexpr int kC = kWaves * BPW;  // 256-group chunks (a tick wave's share) per workgroup
  __shared__ uint64_t red[2][kWaves];
  __shared__ uint32_t mine[2][kC];
  __shared__ uint32_t ids[2][kBlock * 4 * BPW];
  __shared__ uint64_t map_words[kC * 4];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t first_wave = (uint64_t)blockIdx.x * kC;
  uint64_t acc_h = 0, acc_b = 0;
  if (wave_off_hup == nullptr) {
    uint32_t h0 = 0, h1 = 0, b0 = 0, b1 = 0;
    uint64_t i = tid;
    for (; i + kBlock < first_wave; i += 2 * kBlock) {
      const uint4 p0 = partials[i], p1 = partials[i + kBlock];
      h0 += p0.x; b0 += p0.y;
      h1 += p1.x; b1 += p1.y;
    }
    if (i < first_wave) {
      const uint4 p0 = partials[i];
      h0 += p0.x; b0 += p0.y;
    }
    acc_h = (uint64_t)h0 + h1;
    acc_b = (uint64_t)b0 + b1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      acc_h += __shfl_xor(acc_h, o, 64);
      acc_b += __shfl_xor(acc_b, o, 64);
    }
  } else if (lane == 0 && wave == 0) {
    acc_h = wave_off_hup[first_wave];
    acc_b = wave_off_beat[first_wave];
  }
  if (lane == 0) {
    red[0][wave] = acc_h;
    red[1][wave] = acc_b;
  }
  if (tid < kC) {
    uint4 p;
    p.x = p.y = p.z = p.w = 0;
    if (first_wave + tid < n_chunks) p = partials[first_wave + tid];
    mine[0][tid] = p.x;
    mine[1][tid] = p.y;
  }
  __syncthreads();
  uint64_t pos_h = 0, pos_b = 0;  // where this workgroup's ids start in the two lists
  uint32_t tot_h = 0, tot_b = 0;  // the lengths of its two runs
#pragma unroll
  for (int k = 0; k < kWaves; ++k) {
    pos_h += red[0][k];
    pos_b += red[1][k];
  }
#pragma unroll
  for (int k = 0; k < kC; ++k) {
    tot_h += mine[0][k];
    tot_b += mine[1][k];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    totals[0] = pos_h + tot_h;
    totals[1] = pos_b + tot_b;
  }
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const uint32_t ci = (uint32_t)b * kWaves + wave;  // this wave's chunk of the round
    const uint64_t wv = first_wave + ci;
    if (wv >= n_chunks) continue;  // wave-uniform
    uint32_t loc_h = 0, loc_b = 0;  // this chunk's first id inside the workgroup's runs
#pragma unroll
    for (int k = 0; k < kC; ++k) {
      loc_h += (uint32_t)k < ci ? mine[0][k] : 0u;
      loc_b += (uint32_t)k < ci ? mine[1][k] : 0u;
    }
    uint64_t hb[4], bb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hb[k] = hup_bits[wv * 4 + k];
      bb[k] = beat_bits[wv * 4 + k];
    }
    uint32_t rh = loc_h + __popcll(hb[0] & below) + __popcll(hb[1] & below) + __popcll(hb[2] & below) + __popcll(hb[3] & below);
    uint32_t rb = loc_b + __popcll(bb[0] & below) + __popcll(bb[1] & below) + __popcll(bb[2] & below) + __popcll(bb[3] & below);
    uint32_t nib = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t g = (uint32_t)(wv * 256 + 4ull * lane + k);
      if ((hb[k] >> lane) & 1) ids[0][rh++] = g;
      if (!BEAT_BITMAP) {
        if ((bb[k] >> lane) & 1) ids[1][rb++] = g;
      } else {
        nib |= (uint32_t)((bb[k] >> lane) & 1) << k;
      }
    }
    if (BEAT_BITMAP) {
      // lane l holds the four bits of groups 4l .. 4l+3 of the chunk's 256: group-order word j is lanes 16j .. 16j+15
      uint64_t v = (uint64_t)nib << (4 * (lane & 15));
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v |= __shfl_xor(v, o, 64);
      if ((lane & 15) == 0) map_words[ci * 4 + (lane >> 4)] = v;
    }
  }
  __syncthreads();
  run_out(hup_out, hup_cap, pos_h, ids[0], tot_h, tid);
  if (!BEAT_BITMAP) {
    run_out(beat_out, beat_cap, pos_b, ids[1], tot_b, tid);
  } else if (tid < kC * 4 && first_wave * 4 + tid < n_chunks * 4) {
    beat_map[first_wave * 4 + tid] = map_words[tid];
  }
  if (ar.flag != nullptr) {  // RAFTQ_CYCLE_FLAG=arrive (see sweep_segments_kernel): the totals are among the stores fenced before the count
    __shared__ uint32_t last_slot;
    if (arrive_last(ar, &last_slot) && tid == 0) {
      __hip_atomic_store(ar.count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ar.flag, (unsigned long long)ar.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// A Tick of every member of a sweep set in ONE dispatch (blockIdx.y = member): the members' TickArgs come from a device
// table; tick_no advances by `ticks_since` from what the table holds (the host rebuilds the table when a member was
// ticked on its own in between).  1M groups per launch is launch-bound (10 MB: 4.4 us, 0.29 of the HBM peak); eight
// members per dispatch move 80 MB behind one launch boundary.
constexpr int kTickSetRounds = 4;
static __global__ __launch_bounds__(kBlock) void tick_set_kernel(const TickArgs* __restrict__ tab, uint64_t ticks_since, uint64_t n_blocks) {
  TickArgs a = tab[blockIdx.y];
  a.tick_no += ticks_since;
  tick_tile<kTickSetRounds, true>(a, n_blocks);
}
// ... at 16 groups per lane (tick_tile_wide): a workgroup's four waves own kTickWideBlocks 1,024-group blocks each
constexpr int kTickWideBlocks = 1;  // measured: 1 block per wave 14.9 us, 2: 16.1, 4: 17.2 per 8 x 1M groups (profiles/r05)
template <int R>
static __global__ __launch_bounds__(kBlock) void tick_set_wide_kernel(const TickArgs* __restrict__ tab, uint64_t ticks_since, uint64_t n_blocks) {
  TickArgs a = tab[blockIdx.y];
  a.tick_no += ticks_since;
  tick_tile_wide<R, true>(a, n_blocks);
}

// becomeCandidate for a list of (distinct) groups: role = candidate, elapsed = 0,
// every vote slot cleared, the candidate's own slot granted: one store of the group's vote word.
static __global__ __launch_bounds__(kBlock) void campaign_kernel(uint8_t* role, uint32_t* elapsed, uint8_t* votes,
                                                          uint32_t n_peers, uint32_t self_peer,
                                                          const uint64_t* __restrict__ groups, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint64_t g = groups[i];
  role[g] = 1;
  elapsed[g] = 0;
  // every slot cleared, the candidate's own granted: the whole vote word in one store
  if (n_peers <= 8) reinterpret_cast<uint16_t*>(votes)[g] = (uint16_t)(1u << (2 * self_peer));
  else reinterpret_cast<uint32_t*>(votes)[g] = 1u << (2 * self_peer);
}

}  // namespace raftqk
