// raftq_sort_kernels.hpp -- stable LSD radix sort of (u64 key, u32 value) pairs on the low `end_bit` bits of the key,
// hand-written for gfx950.  One user: the sorted walk of the batched Step (raftq_step.hip: enqueue_sorted_walk), the
// path that takes a batch in which some raft group has a run of messages longer than the list walk holds.  Round 1
// used hipcub::DeviceRadixSort there (7 launches, 55 us at 64K messages).
//
// The sorted walk exists for SKEWED batches (a few very hot groups), so nothing below may degrade when every key of a
// tile has the same digit: there is no per-key LDS atomic; lanes of a wave that hold the same digit find each other
// with eight __ballot()s (one per digit bit) and only the first of them touches the counters.
//
// One pass = one 8-bit digit, least significant first:
//   radix_hist_kernel     a workgroup owns `rounds` x 256 consecutive keys (its tile) and counts the tile's digits;
//   (offsets)             small sorts (<= kRadixSelfScanBlocks tiles): none -- the scatter kernel of each tile adds up
//                         the counts of the tiles before it itself (block-major counts, coalesced, L2-resident);
//                         larger: exclusive_sum_u64 (raftq_wire_kernels.hpp) over the digit-major counts, 2 launches;
//   radix_scatter_kernel  re-reads the tile round by round (256 consecutive keys per round, lane order = key order) and
//                         writes every pair to  offset(digit, tile) + rank of the key among the tile's keys of that
//                         digit, where rank = keys of that digit in earlier rounds (LDS running count) + in earlier
//                         waves of this round + in lower lanes of this wave: equal digits keep their order -- stable.
// Passes = ceil(end_bit / 8): 2 for <= 64K groups, 3 up to 16M.  Buffers ping-pong A -> B -> A ...; the caller is told
// which pair holds the result.  HBM traffic per pass: 12 B read twice + 12 B written per pair; at the batch sizes of
// this path (10^4..10^6 pairs) the sort is launch-latency-bound, not bandwidth-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raftq_wire_kernels.hpp"  // kBlock, kWaves, exclusive_sum_u64

namespace raftqk {

constexpr int kRadix = 256;                 // == kBlock: thread t also stands for digit t
constexpr uint32_t kRadixRounds = 8;        // keys per thread of a tile (tile = 2048 keys) unless the sort is huge
constexpr uint32_t kRadixMaxBlocks = 4096;  // above that many tiles the tiles grow instead (bounds the counts array)
constexpr uint32_t kRadixSelfScanBlocks = 128;
static_assert(kRadix == kBlock, "one thread per digit");

struct RadixPlan {
  uint32_t rounds, nb;
  bool scanned;                      // counts are digit-major and prefix-summed by exclusive_sum_u64
  size_t off_scan, off_tot, bytes;   // scratch layout: [counts | scanned counts | scan tile totals]
};

static inline RadixPlan radix_plan(uint64_t n) {
  RadixPlan p;
  p.rounds = kRadixRounds;
  uint64_t nb = (n + (uint64_t)kBlock * p.rounds - 1) / ((uint64_t)kBlock * p.rounds);
  if (nb > kRadixMaxBlocks) {
    p.rounds = (uint32_t)((n + (uint64_t)kBlock * kRadixMaxBlocks - 1) / ((uint64_t)kBlock * kRadixMaxBlocks));
    nb = (n + (uint64_t)kBlock * p.rounds - 1) / ((uint64_t)kBlock * p.rounds);
  }
  p.nb = (uint32_t)(nb ? nb : 1);
  p.scanned = p.nb > kRadixSelfScanBlocks;
  const size_t items = (size_t)kRadix * p.nb;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  p.off_scan = up(items * 8);
  p.off_tot = p.off_scan + (p.scanned ? up(items * 8) : 0);
  p.bytes = p.off_tot + (p.scanned ? up(scan_sum_scratch_bytes(items)) : 0);
  return p;
}

// lanes of this wave that are valid and hold the same digit as this lane (meaningless on invalid lanes)
__device__ __forceinline__ uint64_t radix_peers(bool valid, uint32_t d) {
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool s = (d >> b) & 1u;
    const uint64_t bal = __ballot(valid && s);
    peers &= s ? bal : ~bal;
  }
  return peers;
}

// R > 0: `rounds` == R, known at compile time -- the tile's loads are all issued before the first key is looked at
// (a round per memory latency would make a 2048-key tile cost eight of them); R == 0: any `rounds`, one at a time.
template <int R>
static __global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int shift,
                                                                   uint32_t rounds, uint64_t* __restrict__ counts,
                                                                   bool digit_major) {
  __shared__ uint32_t cnt[kRadix];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  cnt[tid] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * rounds * kBlock;
  auto count = [&](bool valid, uint64_t key) {
    const uint32_t d = valid ? (uint32_t)(key >> shift) & 255u : 0u;
    const uint64_t peers = radix_peers(valid, d);
    if (valid && (peers & ((1ull << lane) - 1)) == 0) atomicAdd(&cnt[d], (uint32_t)__popcll(peers));  // <= 4 adds per digit and round
  };
  if constexpr (R > 0) {
    uint64_t key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t i = base + (uint64_t)r * kBlock + tid;
      key[r] = i < n ? keys[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) count(base + (uint64_t)r * kBlock + tid < n, key[r]);
  } else {
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint64_t i0 = base + (uint64_t)r * kBlock;
      if (i0 >= n) break;  // uniform
      const uint64_t i = i0 + tid;
      count(i < n, i < n ? keys[i] : 0);
    }
  }
  __syncthreads();
  counts[digit_major ? (uint64_t)tid * gridDim.x + blockIdx.x : (uint64_t)blockIdx.x * kRadix + tid] = cnt[tid];
}

template <int R>
static __global__ __launch_bounds__(kBlock) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                                      const uint32_t* __restrict__ vals_in,
                                                                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                      uint64_t n, int shift, uint32_t rounds,
                                                                      const uint64_t* __restrict__ counts, bool scanned) {
  __shared__ uint64_t goff[kRadix];         // where this tile's first key of digit d goes
  __shared__ uint32_t run[kRadix];          // keys of digit d in this tile's earlier rounds
  __shared__ uint32_t wcnt[kWaves][kRadix];  // keys of digit d per wave, this round
  __shared__ uint64_t red[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nb = gridDim.x, b = blockIdx.x;
  if (scanned) {
    goff[tid] = counts[(uint64_t)tid * nb + b];
  } else {
    // block-major raw counts: digit `tid` of every tile (coalesced rows); pre = this digit in earlier tiles,
    // tot = this digit everywhere; then an exclusive scan of tot over the 256 digits
    uint64_t pre = 0, tot = 0;
    for (uint32_t k = 0; k < nb; ++k) {
      const uint64_t c = counts[(uint64_t)k * kRadix + tid];
      tot += c;
      if (k < b) pre += c;
    }
    uint64_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t y = __shfl_up(incl, o, 64);
      if (lane >= (uint32_t)o) incl += y;
    }
    if (lane == 63) red[w] = incl;
    __syncthreads();
    uint64_t before = incl - tot;
    for (uint32_t k = 0; k < w; ++k) before += red[k];
    goff[tid] = before + pre;
  }
  run[tid] = 0;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) wcnt[k][tid] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)b * rounds * kBlock;
  // one round: 256 consecutive pairs, lane order = input order
  auto place = [&](bool valid, uint64_t key, uint32_t val) {
    const uint32_t d = valid ? (uint32_t)(key >> shift) & 255u : 0u;
    const uint64_t peers = radix_peers(valid, d);
    const uint32_t lrank = (uint32_t)__popcll(peers & ((1ull << lane) - 1));
    if (valid && lrank == 0) wcnt[w][d] = (uint32_t)__popcll(peers);
    __syncthreads();
    uint32_t pos = 0;
    if (valid) {
      pos = run[d] + lrank;
      for (uint32_t k = 0; k < w; ++k) pos += wcnt[k][d];
    }
    __syncthreads();
    {  // thread t folds digit t's counts of this round into the running count and clears them for the next round
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < kWaves; ++k) {
        s += wcnt[k][tid];
        wcnt[k][tid] = 0;
      }
      run[tid] += s;
    }
    if (valid) {
      const uint64_t o = goff[d] + pos;
      keys_out[o] = key;
      vals_out[o] = val;
    }
    __syncthreads();
  };
  if constexpr (R > 0) {
    uint64_t key[R];
    uint32_t val[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t i = base + (uint64_t)r * kBlock + tid;
      key[r] = i < n ? keys_in[i] : 0;
      val[r] = i < n ? vals_in[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (base + (uint64_t)r * kBlock >= n) break;  // uniform
      place(base + (uint64_t)r * kBlock + tid < n, key[r], val[r]);
    }
  } else {
    for (uint32_t r = 0; r < rounds; ++r) {
      const uint64_t i0 = base + (uint64_t)r * kBlock;
      if (i0 >= n) break;  // uniform
      const uint64_t i = i0 + tid;
      place(i < n, i < n ? keys_in[i] : 0, i < n ? vals_in[i] : 0);
    }
  }
}

// Sorts n pairs by the low end_bit bits of the key (stable).  (kA, vA) is the input and is overwritten; the result is
// in (kB, vB) when *in_b comes back 1, else in (kA, vA).  scratch: radix_plan(n).bytes, 256-byte aligned.
static inline hipError_t radix_sort_pairs(hipStream_t st, void* scratch, uint64_t* kA, uint32_t* vA, uint64_t* kB, uint32_t* vB,
                                          uint64_t n, int end_bit, int* in_b) {
  *in_b = 0;
  if (n == 0) return hipSuccess;
  const RadixPlan p = radix_plan(n);
  uint64_t* counts = (uint64_t*)scratch;
  uint64_t* scanned = (uint64_t*)((uint8_t*)scratch + p.off_scan);
  uint64_t* tile_tot = (uint64_t*)((uint8_t*)scratch + p.off_tot);
  for (int shift = 0; shift < end_bit; shift += 8) {
    const uint64_t* ki = *in_b ? kB : kA;
    const uint32_t* vi = *in_b ? vB : vA;
    uint64_t* ko = *in_b ? kA : kB;
    uint32_t* vo = *in_b ? vA : vB;
    if (p.rounds == kRadixRounds)
      hipLaunchKernelGGL(radix_hist_kernel<(int)kRadixRounds>, dim3(p.nb), dim3(kBlock), 0, st, ki, n, shift, p.rounds, counts, p.scanned);
    else
      hipLaunchKernelGGL(radix_hist_kernel<0>, dim3(p.nb), dim3(kBlock), 0, st, ki, n, shift, p.rounds, counts, p.scanned);
    if (p.scanned) {
      if (hipError_t e = exclusive_sum_u64(counts, scanned, (uint64_t)kRadix * p.nb, tile_tot, st)) return e;
    }
    const uint64_t* offs = p.scanned ? scanned : counts;
    if (p.rounds == kRadixRounds)
      hipLaunchKernelGGL(radix_scatter_kernel<(int)kRadixRounds>, dim3(p.nb), dim3(kBlock), 0, st, ki, vi, ko, vo, n, shift, p.rounds, offs, p.scanned);
    else
      hipLaunchKernelGGL(radix_scatter_kernel<0>, dim3(p.nb), dim3(kBlock), 0, st, ki, vi, ko, vo, n, shift, p.rounds, offs, p.scanned);
    *in_b ^= 1;
  }
  return hipGetLastError();
}

}  // namespace raftqk
