// raftq_wire_kernels.hpp -- device code of the batched wire / WAL codecs (include/raftq_wire.h).
//
// The reference marshals one raftpb.Message per rafthttp send (raft.go:230) and one walpb.Record
// per wal.Save / ReadAll step (raft.go:228, :124), each inside its group's goroutine.  Here a whole
// batch is one launch chain on the GPU:
//   * protobuf fields: one lane per message / record.  Sizes first, an exclusive scan turns them
//     into frame offsets, then every lane writes its own frame.  Output bytes go through an 8-byte
//     register accumulator (one unaligned 8-byte store per 8 bytes produced); input varints are
//     cut out of one unaligned 8-byte load with a SWAR compress, not a byte loop.
//   * entry payloads never pass through a lane's byte loop: a wave per message / record copies them
//     (16 B per lane per step) and CRCs the long ones cooperatively.
//   * CRC-32C: per-record CRCs are independent (LDS table, one lane per record; 64 lanes per long
//     record).  The running CRC that chains the records of a WAL segment is a scan over affine maps
//     x -> x * X^(8 len) + crc in GF(2)[X]/P: an associative, non-commutative operator (a hand-written
//     two-launch scan, below).  A crcType record that re-seeds the chain is the constant map (multiplier 0).
// Bound: PCIe (the ABI takes and returns host buffers) -- these kernels are a few us per 64K
// records; DESIGN.md 4.10 has the measured split.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raftq_kernels.hpp"
#include "raftq_wire_parse.hpp"

namespace raftqk {

constexpr uint32_t kCoopBytes = 512;  // payloads longer than this are CRC'd by a whole wave

// ---- CRC-32C (Castagnoli, reflected 0x82f63b78; bit 31 of a word is x^0) -------------------------

constexpr uint32_t kCastagnoli = 0x82f63b78u;

// a(x) * b(x) mod P
__host__ __device__ constexpr uint32_t crc_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; ++i) {
    p ^= b & (0u - (a >> 31));
    a <<= 1;
    b = (b >> 1) ^ (kCastagnoli & (0u - (b & 1u)));
  }
  return p;
}

struct XpowTable {
  uint32_t v[40];  // v[k] = x^(8 * 2^k) mod P
  constexpr XpowTable() : v{} {
    uint32_t s = 0x00800000u;  // x^8
    for (int k = 0; k < 40; ++k) {
      v[k] = s;
      s = crc_mulmod(s, s);
    }
  }
};
__device__ const XpowTable kXpow8{};

// x^(8 n) mod P for n < 256 and x^(8 * 256 n) mod P for n < 256: a record's Data is tens of bytes, a payload hundreds --
// the multiplier of nearly every record is ONE table read, of the rest one read more and one product.  (Round 2 multiplied
// the x^(8 * 2^k) of every set bit of n together: 5-6 products of 32 shift-xor steps each per record -- more instructions than
// the record's whole parse, and most of what wal_dec_kernel's 25.6 us and wal_enc_crc_kernel's 14 us were made of.)
struct XpowBytes {
  uint32_t lo[256], hi[256];
  constexpr XpowBytes() : lo{}, hi{} {
    uint32_t s = 0x80000000u;  // x^0
    for (int i = 0; i < 256; ++i) {
      lo[i] = s;
      s = crc_mulmod(s, 0x00800000u);  // * x^8
    }
    const uint32_t step = s;  // x^(8 * 256)
    s = 0x80000000u;
    for (int i = 0; i < 256; ++i) {
      hi[i] = s;
      s = crc_mulmod(s, step);
    }
  }
};
__device__ const XpowBytes kXpowBytes{};

// x^(8 n) mod P
__device__ __forceinline__ uint32_t crc_xpow8(uint64_t n) {
  if (n < 65536) {
    const uint32_t lo = kXpowBytes.lo[n & 255u];
    return n < 256 ? lo : crc_mulmod(lo, kXpowBytes.hi[n >> 8]);
  }
  uint32_t r = 0x80000000u;
  for (int k = 0; n != 0 && k < 40; ++k, n >>= 1)
    if (n & 1u) r = crc_mulmod(r, kXpow8.v[k]);
  return r;
}

// the affine map  x -> x * p + c  one record applies to the running CRC
struct CrcPair {
  uint32_t c, p;
};
struct CrcCompose {  // (l then r)
  __host__ __device__ __forceinline__ CrcPair operator()(const CrcPair& l, const CrcPair& r) const {
    CrcPair o;
    o.c = crc_mulmod(r.p, l.c) ^ r.c;
    o.p = crc_mulmod(l.p, r.p);
    return o;
  }
};

// ---- inclusive scan of CrcPair under CrcCompose (the running CRC of a WAL segment) ------------------
// Two launches: every 256-item block scans itself (shuffle scan per wave, the four wave totals through LDS) and
// publishes its total; then every block composes its predecessors' totals and folds them into its items.
// The generic library scan leaves 16 workgroups of 4K items for a 64K-record batch and spends 40 us there
// (the operator is two 32-step shift-xor multiplications); this shape spreads the same work over 256 workgroups.
constexpr CrcPair kCrcIdentity = {0u, 0x80000000u};

__device__ __forceinline__ CrcPair crc_wave_inclusive(CrcPair v) {
  const uint32_t lane = threadIdx.x & 63;
  CrcCompose op;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    CrcPair left;
    left.c = __shfl_up(v.c, d);
    left.p = __shfl_up(v.p, d);
    if (lane >= (uint32_t)d) v = op(left, v);
  }
  return v;
}

// block-wide inclusive scan of one item per thread (kBlock threads); *total = the block's composition
__device__ __forceinline__ CrcPair crc_block_inclusive(CrcPair v, CrcPair* wave_tot /*LDS [kWaves]*/, CrcPair* total) {
  CrcCompose op;
  v = crc_wave_inclusive(v);
  const uint32_t w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) wave_tot[w] = v;
  __syncthreads();
  CrcPair pre = kCrcIdentity;
  for (uint32_t k = 0; k < w; ++k) pre = op(pre, wave_tot[k]);
  if (w) v = op(pre, v);
  if (total) {
    CrcPair t = wave_tot[0];
    for (uint32_t k = 1; k < kWaves; ++k) t = op(t, wave_tot[k]);
    *total = t;
  }
  return v;
}

static __global__ __launch_bounds__(kBlock) void crc_scan_blocks_kernel(const CrcPair* __restrict__ in, CrcPair* __restrict__ out,
                                                                        uint64_t n, CrcPair* __restrict__ block_tot) {
  __shared__ CrcPair wave_tot[kWaves];
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  CrcPair total;
  const CrcPair v = crc_block_inclusive(i < n ? in[i] : kCrcIdentity, wave_tot, &total);
  if (i < n) out[i] = v;
  if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}

// Every block composes the totals of the blocks before it ITSELF (in order: the operator is not commutative) and
// folds the result into its items.  This replaces the single-block pass over the totals, which took 23 us on its
// own for 256 totals (profiles/r02/wire_kernel_stats.csv) -- one block's worth of the two-multiplication operator
// with the whole chip waiting for it.  block_tot holds the raw per-block totals of crc_scan_blocks_kernel.
static __global__ __launch_bounds__(kBlock) void crc_scan_apply_kernel(CrcPair* __restrict__ out, uint64_t n,
                                                                       const CrcPair* __restrict__ block_tot) {
  if (blockIdx.x == 0) return;  // nothing precedes the first block
  __shared__ CrcPair wave_tot[kWaves];
  CrcCompose op;
  CrcPair carry = kCrcIdentity;
  for (uint32_t base = 0; base < blockIdx.x; base += kBlock) {
    const uint32_t j = base + threadIdx.x;
    CrcPair total;
    (void)crc_block_inclusive(j < blockIdx.x ? block_tot[j] : kCrcIdentity, wave_tot, &total);
    carry = op(carry, total);
    __syncthreads();  // wave_tot is reused by the next chunk
  }
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  out[i] = op(carry, out[i]);
}

// ---------------------------------------------------------------------------
// Exclusive prefix sum of u64 (frame sizes -> frame offsets, entry counts -> entry bases), hand-written for the
// same reason as the CRC scan above: the library scan's default tile leaves a 64K-item batch to 16 workgroups
// and costs 15 us of launch + operator latency for 0.5 MB of data.  Two launches:
//   scan_sum_local_kernel: a workgroup owns kScanTile = 2048 consecutive items (8 per thread: two 32-byte loads),
//     scans them in registers + wave shuffles + one LDS hop, writes the tile-local exclusive sums and its total;
//   scan_sum_add_kernel:   every workgroup adds up the totals of the tiles before it (<= a few thousand
//     L2-resident words) and adds that to its tile -- no third launch to scan the totals.
// out[n_items - 1] ends up holding the sum of everything before the last item; callers pass n + 1 items with a
// zero last item so that out[n] is the grand total (as they did with the library scan).
constexpr int kScanPer = 8;
constexpr int kScanTile = kBlock * kScanPer;

static __global__ __launch_bounds__(kBlock) void scan_sum_local_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                        uint64_t n, uint64_t* __restrict__ tile_tot) {
  __shared__ uint64_t wave_tot[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)tid * kScanPer;
  uint64_t v[kScanPer];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) v[k] = base + k < n ? in[base + k] : 0;
  uint64_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) sum += v[k];
  uint64_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = __shfl_up(incl, o, 64);
    if (lane >= (uint32_t)o) incl += y;
  }
  if (lane == 63) wave_tot[w] = incl;
  __syncthreads();
  uint64_t run = incl - sum;
  for (uint32_t k = 0; k < w; ++k) run += wave_tot[k];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (tid == kBlock - 1) tile_tot[blockIdx.x] = run;
}

static __global__ __launch_bounds__(kBlock) void scan_sum_add_kernel(uint64_t* __restrict__ out, uint64_t n,
                                                                      const uint64_t* __restrict__ tile_tot) {
  if (blockIdx.x == 0) return;  // nothing precedes the first tile
  __shared__ uint64_t red[kWaves];
  const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint64_t acc = 0;
  for (uint32_t i = tid; i < blockIdx.x; i += kBlock) acc += tile_tot[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) red[w] = acc;
  __syncthreads();
  uint64_t pre = 0;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) pre += red[k];
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)tid * kScanPer;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k)
    if (base + k < n) out[base + k] += pre;
}

// The end of a codec call's device chain: the two words the host waits for (a total the scan left in device memory, the
// call's malformed counter / bad flag) go straight into the handle's page-locked block, and the counter is zero again
// for the next call.  One 3 us launch instead of two 12 us copies (the runtime moves 8 bytes with a blit kernel each)
// and a 3 us memset (profiles/r03/codec_call_probe.txt).
static __global__ void wire_tail_kernel(const uint64_t* __restrict__ total, unsigned long long* flag, uint64_t* __restrict__ pin) {
  if (threadIdx.x == 0) {
    pin[0] = *total;
    pin[1] = *flag;
    *flag = 0;
  }
}

// host side of the scan: bytes of scratch for the tile totals, and the two launches
static inline size_t scan_sum_scratch_bytes(uint64_t n_items) { return (size_t)((n_items + kScanTile - 1) / kScanTile) * 8 + 8; }
static inline hipError_t exclusive_sum_u64(const uint64_t* in, uint64_t* out, uint64_t n_items, uint64_t* tile_tot, hipStream_t st) {
  if (n_items == 0) return hipSuccess;
  const unsigned nb = (unsigned)((n_items + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(scan_sum_local_kernel, dim3(nb), dim3(kBlock), 0, st, in, out, n_items, tile_tot);
  if (nb > 1) hipLaunchKernelGGL(scan_sum_add_kernel, dim3(nb), dim3(kBlock), 0, st, out, n_items, (const uint64_t*)tile_tot);
  return hipGetLastError();
}

// the slicing-by-8 tables (raftq_wire_parse.hpp) in LDS, built by the block (256 threads): t[k * 256 + i]
__device__ __forceinline__ void crc_table_init(uint32_t* tab /*[kCrcTabs * 256]*/) {
  const uint32_t i = threadIdx.x;  // blockDim.x == 256
  uint32_t c = 0;
  if (i < 256) {
    crc_tables_entry(i, &c);
    tab[i] = c;
  }
  __syncthreads();
  if (i < 256)
    for (int k = 1; k < kCrcTabs; ++k) {
      c = (c >> 8) ^ tab[c & 0xffu];
      tab[k * 256 + i] = c;
    }
  __syncthreads();
}

// raw (un-inverted) register update
__device__ __forceinline__ uint32_t crc_byte(const uint32_t* tab, uint32_t raw, uint8_t b) {
  return tab[(raw ^ b) & 0xffu] ^ (raw >> 8);
}
__device__ __forceinline__ uint32_t crc_varint(const uint32_t* tab, uint32_t raw, uint64_t v) {
  while (v >= 0x80) {
    raw = crc_byte(tab, raw, (uint8_t)(v | 0x80));
    v >>= 7;
  }
  return crc_byte(tab, raw, (uint8_t)v);
}
__device__ __forceinline__ uint32_t crc_field(const uint32_t* tab, uint32_t raw, uint8_t tag, uint64_t v) {
  return crc_varint(tab, crc_byte(tab, raw, tag), v);
}
__device__ inline uint32_t crc_span(const uint32_t* tab, uint32_t raw, const uint8_t* p, uint64_t n) {
  uint64_t i = 0;
  for (; i + 8 <= n; i += 8) raw = crc_step8(tab, raw, load_u64(p + i));
  for (; i < n; ++i) raw = crc_byte(tab, raw, p[i]);
  return raw;
}

// CRC-32C (init 0, standard inversions) of p[0, n) by the 64 lanes of a wave: each lane a contiguous
// chunk, then an ordered tree of compositions.  Valid in lane 0.
__device__ inline uint32_t wave_crc(const uint32_t* tab, const uint8_t* p, uint64_t n) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t chunk = (n + 63) / 64;
  const uint64_t lo = lane * chunk < n ? lane * chunk : n;
  const uint64_t hi = lo + chunk < n ? lo + chunk : n;
  CrcPair me;
  me.c = ~crc_span(tab, 0xffffffffu, p + lo, hi - lo);
  me.p = crc_xpow8(hi - lo);
  if (hi == lo) me.c = 0;  // crc of nothing
  CrcCompose op;
  for (int s = 1; s < 64; s <<= 1) {
    CrcPair right;
    right.c = __shfl_down(me.c, s);
    right.p = __shfl_down(me.p, s);
    if ((lane & (2 * s - 1)) == 0) me = op(me, right);
  }
  return me.c;
}

// dst[0, n) = src[0, n) by a wave, 16 B per lane per step (both sides may be unaligned)
__device__ inline void wave_copy(uint8_t* dst, const uint8_t* src, uint64_t n) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t chunks = n >> 4;
  for (uint64_t c = lane; c < chunks; c += 64) {
    uint4 v;
    __builtin_memcpy(&v, src + (c << 4), 16);
    __builtin_memcpy(dst + (c << 4), &v, 16);
  }
  for (uint64_t b = (chunks << 4) + lane; b < n; b += 64) dst[b] = src[b];
}

// dst[0, n) = src[0, n) by the eight lanes of a sub-group (sub = lane & 7), 16 B per lane per step
__device__ inline void sub8_copy(uint8_t* dst, const uint8_t* src, uint64_t n, uint32_t sub) {
  const uint64_t chunks = n >> 4;
  for (uint64_t c = sub; c < chunks; c += 8) {
    uint4 v;
    __builtin_memcpy(&v, src + (c << 4), 16);
    __builtin_memcpy(dst + (c << 4), &v, 16);
  }
  for (uint64_t b = (chunks << 4) + sub; b < n; b += 8) dst[b] = src[b];
}

// ---- protobuf primitives ----------------------------------------------------------------------------

// sovRaft: encoded size of a varint
__device__ __forceinline__ uint32_t sov(uint64_t v) { return v ? (uint32_t)(70 - __clzll((long long)v)) / 7u : 1u; }

// byte sink over global memory: bytes collect in a register, 8 at a time go out in one store
struct Sink {
  uint8_t* p;
  uint64_t acc = 0;
  uint32_t k = 0;  // bytes in acc
  __device__ explicit Sink(uint8_t* dst) : p(dst) {}
  __device__ __forceinline__ void byte(uint8_t b) {
    acc |= (uint64_t)b << (8 * k);
    if (++k == 8) {
      __builtin_memcpy(p, &acc, 8);
      p += 8;
      acc = 0;
      k = 0;
    }
  }
  __device__ __forceinline__ void varint(uint64_t v) {
    while (v >= 0x80) {
      byte((uint8_t)(v | 0x80));
      v >>= 7;
    }
    byte((uint8_t)v);
  }
  __device__ __forceinline__ void field(uint8_t tag, uint64_t v) {
    byte(tag);
    varint(v);
  }
  __device__ __forceinline__ void flush() {
    for (uint32_t i = 0; i < k; ++i) p[i] = (uint8_t)(acc >> (8 * i));
    p += k;
    acc = 0;
    k = 0;
  }
  __device__ __forceinline__ void skip(uint64_t n) {  // someone else writes these bytes
    flush();
    p += n;
  }
  __device__ __forceinline__ void u64_be(uint64_t v) {
    for (int i = 56; i >= 0; i -= 8) byte((uint8_t)(v >> i));
  }
  __device__ __forceinline__ void u64_le(uint64_t v) {
    for (int i = 0; i < 64; i += 8) byte((uint8_t)(v >> i));
  }
};

// ---- raftpb.Entry -----------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t entry_size(uint32_t type, uint64_t term, uint64_t index, uint32_t data_len) {
  uint64_t n = 3 + sov(type) + sov(term) + sov(index);
  if (data_len) n += 1 + sov(data_len) + data_len;
  return n;
}

// ---- raftpb.Message ---------------------------------------------------------------------------------

// bytes of the six fields in front of the entries
__device__ __forceinline__ uint64_t msg_head_size(const WireMsg& m) {
  return 6 + sov(m.type) + sov((uint64_t)m.to + 1) + sov((uint64_t)m.from + 1) + sov(m.term) + sov(m.log_term) +
         sov(m.index);
}
// commit, the empty snapshot (10 bytes), reject (2), rejectHint, group
__device__ __forceinline__ uint64_t msg_tail_size(const WireMsg& m) {
  return 1 + sov(m.commit) + 10 + 2 + 1 + sov(m.reject_hint) + 1 + sov(m.group);
}

// ---- kernels: raftpb.Message -> rafthttp stream frames -----------------------------------------

// sizes[i] = 8 + Message.Size(); sizes[n] = 0 (so the exclusive scan yields n + 1 offsets)
static __global__ __launch_bounds__(kBlock) void wire_enc_size_kernel(const WireMsg* __restrict__ msgs, uint64_t n,
                                                                      const WireEnt* __restrict__ ents,
                                                                      uint64_t n_ents, uint64_t pool_bytes,
                                                                      uint64_t* __restrict__ sizes,
                                                                      unsigned int* bad) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool is_bad = false;
  if (i < n) {
    const WireMsg m = msgs[i];
    is_bad = m.to >= 255 || m.from >= 255 || (m.n_ents != 0 && (uint64_t)m.ent_first + m.n_ents > n_ents);
    uint64_t sz = 8 + msg_head_size(m) + msg_tail_size(m);
    if (!is_bad) {
      for (uint32_t k = 0; k < m.n_ents; ++k) {
        const WireEnt e = ents[m.ent_first + k];
        if (e.data_len != 0 && (e.data_off > pool_bytes || e.data_len > pool_bytes - e.data_off)) is_bad = true;
        const uint64_t es = entry_size(e.type, e.term, e.index, e.data_len);
        sz += 1 + sov(es) + es;
      }
    }
    sizes[i] = is_bad ? 0 : sz;
  } else if (i == n) {
    sizes[i] = 0;
  }
  if (__ballot(is_bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// every byte of frame i except the entry payloads
static __global__ __launch_bounds__(kBlock) void wire_enc_write_kernel(const WireMsg* __restrict__ msgs, uint64_t n,
                                                                       const WireEnt* __restrict__ ents,
                                                                       const uint64_t* __restrict__ off,
                                                                       uint8_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const WireMsg m = msgs[i];
  const uint64_t a = off[i];
  Sink s(out + a);
  s.u64_be(off[i + 1] - a - 8);
  s.field(0x08, m.type);
  s.field(0x10, (uint64_t)m.to + 1);
  s.field(0x18, (uint64_t)m.from + 1);
  s.field(0x20, m.term);
  s.field(0x28, m.log_term);
  s.field(0x30, m.index);
  for (uint32_t k = 0; k < m.n_ents; ++k) {
    const WireEnt e = ents[m.ent_first + k];
    s.field(0x3a, entry_size(e.type, e.term, e.index, e.data_len));
    s.field(0x08, e.type);
    s.field(0x10, e.term);
    s.field(0x18, e.index);
    if (e.data_len) {
      s.field(0x22, e.data_len);
      s.skip(e.data_len);
    }
  }
  s.field(0x40, m.commit);
  s.u64_le(0x0010000a0612084aull);  // 4a 08 12 06 0a 00 10 00 | 18 00: the empty Snapshot
  s.byte(0x18);
  s.byte(0x00);
  s.field(0x50, m.reject ? 1 : 0);
  s.field(0x58, m.reject_hint);
  s.field(0x60, m.group);
  s.flush();
}

// one wave per message: its entries' payloads, pool -> frame
static __global__ __launch_bounds__(kBlock) void wire_enc_payload_kernel(const WireMsg* __restrict__ msgs, uint64_t n,
                                                                         const WireEnt* __restrict__ ents,
                                                                         const uint64_t* __restrict__ off,
                                                                         const uint8_t* __restrict__ pool,
                                                                         uint8_t* __restrict__ out) {
  const uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (i >= n) return;
  const uint32_t n_ents = msgs[i].n_ents;
  if (n_ents == 0) return;
  const WireMsg m = msgs[i];
  uint64_t pos = off[i] + 8 + msg_head_size(m);
  for (uint32_t k = 0; k < n_ents; ++k) {
    const WireEnt e = ents[m.ent_first + k];
    const uint64_t es = entry_size(e.type, e.term, e.index, e.data_len);
    pos += 1 + sov(es) + 3 + sov(e.type) + sov(e.term) + sov(e.index);
    if (e.data_len) {
      pos += 1 + sov(e.data_len);
      wave_copy(out + pos, pool + e.data_off, e.data_len);
      pos += e.data_len;
    }
  }
}

// ---- kernels: stream frames -> raftpb.Message headers ----------------------------------------------

// The scalar fields of the frame a lane is parsing, in LDS: slot-major, one 8-byte column per lane (consecutive lanes,
// consecutive banks -- a wave filing the same field is conflict-free, a wave filing different fields nearly so).
template <int STRIDE>
struct LdsFileT {
  uint64_t* col;  // &file[0][tid]
  __device__ __forceinline__ void put(uint32_t slot, uint64_t v) { col[slot * STRIDE] = v; }
  __device__ __forceinline__ uint64_t get(uint32_t slot) const { return col[slot * STRIDE]; }
};
using LdsFile = LdsFileT<kBlock>;

// A wave's 64 frames are one contiguous run of the stream: the wave copies it into LDS (16 bytes per lane per step)
// and its lanes then read their frames from there -- ~64 cycles per dependent read instead of ~600 from L2.  Runs longer
// than the wave's stage (frames with large payloads), frame offsets that are not monotonic (garbage input) and streams
// that are not 16-byte aligned stay in global memory, per wave / per lane; the bytes are the same either way.
constexpr uint32_t kStageBytes = 8192;  // per wave
struct WaveStage {
  const uint32_t* words = nullptr;  // the wave's LDS stage when the run [lo16, hi) was copied
  uint64_t lo16 = 0, lo = 0, hi = 0;
};
// a, b: this lane's frame extent (off[i], off[i + 1]; 0, 0 for lanes past the batch) -- the wave's run is read off its
// first and last lane's extents, so staging costs no load of its own in front of the stream's
__device__ __forceinline__ uint64_t wave_bcast_u64(uint64_t x, int lane) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)x, lane), hi = __builtin_amdgcn_readlane((uint32_t)(x >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ WaveStage stage_wave_frames(const uint8_t* stream, uint64_t nbytes, uint64_t a, uint64_t b, bool live,
                                                       uint32_t* lds /* this wave's kStageBytes */) {
  WaveStage st;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t alive = __ballot(live);
  if (alive == 0 || ((uintptr_t)stream & 15) != 0) return st;
  const uint64_t lo = wave_bcast_u64(a, 0);  // lane 0 is live whenever any lane is (lanes fill from the front)
  const uint64_t hi = wave_bcast_u64(b, 63 - __builtin_clzll(alive));
  const uint64_t lo16 = lo & ~15ull;
  if (!(lo <= hi && hi <= nbytes) || hi - lo16 + 16 > kStageBytes) return st;  // wave-uniform
  const uint64_t full = (hi - lo16) >> 4;  // whole 16-byte chunks inside the buffer
  for (uint64_t c = lane; c < full; c += 64) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(stream + lo16 + (c << 4));
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(lds) + (c << 4)) = v;
  }
  const uint64_t tail0 = lo16 + (full << 4);  // < 16 bytes up to hi, byte by byte (the buffer may end at hi)
  if (lane < hi - tail0) reinterpret_cast<uint8_t*>(lds)[(full << 4) + lane] = stream[tail0 + lane];
  st.words = lds;
  st.lo16 = lo16;
  st.lo = lo;
  st.hi = hi;
  return st;
}
__device__ __forceinline__ ByteSrc frame_src(const WaveStage& st, const uint8_t* stream, uint64_t nbytes, uint64_t a, uint64_t b) {
  ByteSrc src;
  src.p = stream + a + 8;
  src.safe = nbytes - a - 8;
  src.words = nullptr;
  src.shift = 0;
  if (st.words && a >= st.lo && b <= st.hi) {  // this lane's frame lies inside the staged run
    src.words = st.words;
    src.shift = (uint32_t)(a + 8 - st.lo16);
    src.safe = ~0ull;  // windows past the run's end read stale LDS bytes inside the wave's stage: never used (scope checks)
  }
  return src;
}
// frame_body() with the length word taken from the staged copy when the frame lies in it
__device__ __forceinline__ bool frame_body_staged(const ByteSrc& src, const uint8_t* stream, uint64_t nbytes, uint64_t a, uint64_t b,
                                                  bool big_endian) {
  if (!(a <= b && b <= nbytes && b - a >= 8)) return false;
  uint64_t w;
  if (src.words) {
    ByteSrc head = src;
    head.shift = src.shift - 8;
    w = head.ld8(0);
  } else {
    w = load_u64(stream + a);
  }
  if (big_endian) w = __builtin_bswap64(w);
  return w == b - a - 8;
}

// pass 1: parse every frame, count its entries.  ent_cnt[n] = 0.
// (Round 2's form of this kernel ran 4,400 instructions and 44 dependent loads per wave: 17 us for 64K frames without
// entries, 45 us with 15 % MsgApp; raftq_wire_parse.hpp has what changed.)
static __global__ __launch_bounds__(kBlock) void wire_dec_kernel(const uint8_t* __restrict__ stream, uint64_t nbytes,
                                                                 const uint64_t* __restrict__ off, uint64_t n,
                                                                 WireMsg* __restrict__ msgs,
                                                                 uint64_t* __restrict__ ent_cnt,
                                                                 unsigned long long* n_bad) {
  __shared__ uint64_t file[kFileSlots * kBlock];
  __shared__ __attribute__((aligned(16))) uint32_t stage[kWaves][kStageBytes / 4];
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t a = i < n ? off[i] : 0, b = i < n ? off[i + 1] : 0;
  const WaveStage st = stage_wave_frames(stream, nbytes, a, b, i < n, stage[threadIdx.x >> 6]);
  __syncthreads();
  bool malformed = false;
  if (i < n) {
    WireMsg m;
    const ByteSrc src = frame_src(st, stream, nbytes, a, b);
    bool ok = frame_body_staged(src, stream, nbytes, a, b, true);
    if (ok) {
      LdsFile f{file + threadIdx.x};
      ok = parse_msg<false>(src, b - a - 8, a + 8, f, m, nullptr, 0, 0, 0);
    }
    if (!ok) {
      m.group = m.term = m.log_term = m.index = m.commit = m.reject_hint = 0;
      m.from = 0;
      m.type = m.reject = m.to = 0;
      m.ent_first = m.n_ents = 0;
      m.flags = kWireMalformed;
      malformed = true;
    }
    msgs[i] = m;
    ent_cnt[i] = m.n_ents;
  } else if (i == n) {
    ent_cnt[i] = 0;
  }
  const uint64_t mb = __ballot(malformed);
  if (mb != 0 && (threadIdx.x & 63) == 0) atomicAdd(n_bad, (unsigned long long)__popcll(mb));
}

// pass 2: the entry headers, in message order (ent_base = exclusive scan of ent_cnt).  Only frames that carry entries are
// walked, and only as far as their last entry: what follows was validated by pass 1.
static __global__ __launch_bounds__(kBlock) void wire_dec_ents_kernel(const uint8_t* __restrict__ stream, uint64_t nbytes,
                                                                      const uint64_t* __restrict__ off, uint64_t n,
                                                                      WireMsg* __restrict__ msgs,
                                                                      const uint64_t* __restrict__ ent_base,
                                                                      WireEnt* __restrict__ ents, uint64_t ents_cap) {
  __shared__ uint64_t file[kFileSlots * kBlock];
  __shared__ __attribute__((aligned(16))) uint32_t stage[kWaves][kStageBytes / 4];
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t first = i < n ? ent_base[i] : 0;
  const uint32_t cnt = i < n ? (uint32_t)(ent_base[i + 1] - first) : 0u;
  const uint64_t a = i < n ? off[i] : 0, b = i < n ? off[i + 1] : 0;
  WaveStage st;
  if (__ballot(cnt != 0) != 0) st = stage_wave_frames(stream, nbytes, a, b, i < n, stage[threadIdx.x >> 6]);  // wave-uniform
  __syncthreads();
  if (cnt == 0) return;
  WireMsg m;
  LdsFile f{file + threadIdx.x};
  (void)parse_msg<true>(frame_src(st, stream, nbytes, a, b), b - a - 8, a + 8, f, m, ents, first, ents_cap, cnt);
  msgs[i].ent_first = (uint32_t)first;
}

// ---- the streaming form of a codec call (round 4) --------------------------------------------------------------------
// On page-locked caller buffers round 3's call was copy-in kernel -> 3-5 small kernels -> copy-out kernel: the link's two
// directions never worked at the same time (245 us to decode 64K frames whose inbound and outbound bytes are 77 us of link
// each).  What the link takes (profiles/r04/pcie_duplex_probe.jsonl): a FEW workgroups walking host memory in order pull
// 55 GB/s, many workgroups each pulling a piece of their own 42; workgroups push 55; ONE kernel whose workgroups do both
// moves 38-41 GB/s each way at once; two kernels on two streams do not overlap at all when one of them writes host memory
// (the round-1 finding still holds on ROCm 7.2).  So a call is ONE kernel with two roles:
//   readers  the first few workgroups copy the caller's input arrays, chunk by chunk and in order, into device scratch
//            (agent-scope write-through stores) and raise a per-chunk flag -- they depend on nothing;
//   workers  the others take tiles of 256 frames / records by ticket, wait for the chunks that hold the tile's input,
//            stage it in LDS (DMA from the scratch), parse, learn where their output goes from a decoupled look-back over
//            the tiles before them (one 64-bit status word per tile: epoch | flag | value, relaxed agent-scope atomics --
//            value and flag travel together, no fence), and push their records out through LDS, 4 KB of consecutive host
//            memory per store instruction.
// The first tile's records leave ~25 us into the call and from then on both directions are busy.  (A first form had every
// worker pull its own tile's bytes straight from the host, with a window on how many tiles might have pulls in flight:
// 166-181 us for 64K frames against this form's 175-193 on the same boxes, but nothing in it can feed the encoders' indirect
// reads -- entry ranges, payload ranges -- from host memory; this form serves all four codecs.)
// Tickets (not blockIdx) order the tiles, and readers are the launch's first workgroups: whatever a worker waits for has
// been claimed by a running workgroup, so nothing deadlocks however few workgroups are resident.
constexpr int kLbValueBits = 46, kLbFlagShift = 46, kLbEpochShift = 48;
constexpr int kLbArrays = 6;  // [0..3] running sums of a kernel, [4] spare (measurement builds: phase stamps), [5] the readers' chunk flags
constexpr int kLbFlags = 5, kLbSpare = 4;
constexpr uint64_t kLbValueMask = (1ull << kLbValueBits) - 1;
constexpr uint32_t kLbAggregate = 1, kLbInclusive = 2;

struct TileCtl {
  unsigned int* ticket;            // monotonic across calls: tile = ticket - ticket_base; ticket[1] = "a wait gave up"
  uint32_t ticket_base;
  uint32_t epoch;                  // 1 .. 0xffff: status words of older calls read as "not yet"
  uint32_t ablate;                 // measurement builds (RAFTQ_WIRE_TRACE) only: bit 0 = records stay, bit 1 = entry headers stay
  unsigned long long* status[kLbArrays];  // [n_tiles] each: independent running sums (a kernel uses the first two or three)
};

__host__ __device__ __forceinline__ uint64_t lb_word(uint32_t epoch, uint32_t flag, uint64_t v) {
  return ((uint64_t)epoch << kLbEpochShift) | ((uint64_t)flag << kLbFlagShift) | (v & kLbValueMask);
}
// A word of THIS call, waited for.  The wait is bounded (a second or so of polling: five orders of magnitude above a
// tile's life) so that no fault -- a lost workgroup, a corrupted control block -- can hang the queue: the waiter gives up,
// raises *stuck (the call then fails with RAFTQ_EHIP) and continues with zero.
__device__ __forceinline__ uint64_t lb_wait(const unsigned long long* w, uint32_t epoch, unsigned int* stuck) {
  for (uint32_t spin = 0; spin < (1u << 23); ++spin) {
    const uint64_t s = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(s >> kLbEpochShift) == epoch && ((s >> kLbFlagShift) & 3u) != 0) return s;
    __builtin_amdgcn_s_sleep(2);
  }
  atomicOr(stuck, 1u);
  return lb_word(epoch, kLbInclusive, 0);
}
// ONE lane per workgroup: publish this tile's aggregate, add up the tiles before it, publish the inclusive sum.
// -> the exclusive prefix of `tile`.
__device__ inline uint64_t lb_exclusive(unsigned long long* status, uint32_t epoch, uint32_t tile, uint64_t aggregate, unsigned int* stuck) {
  if (tile == 0) {
    __hip_atomic_store(status, lb_word(epoch, kLbInclusive, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  __hip_atomic_store(status + tile, lb_word(epoch, kLbAggregate, aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t prefix = 0;
  for (uint32_t j = tile; j-- > 0;) {
    const uint64_t s = lb_wait(status + j, epoch, stuck);
    prefix += s & kLbValueMask;
    if (((s >> kLbFlagShift) & 3u) == kLbInclusive) break;
  }
  __hip_atomic_store(status + tile, lb_word(epoch, kLbInclusive, prefix + aggregate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return prefix;
}

// A worker's tile: the next ticket (wave-uniform result through LDS).
__device__ __forceinline__ uint32_t next_tile(const TileCtl& ctl, uint32_t* slot /*LDS*/) {
  __syncthreads();  // the previous tile's readers of *slot (and of every other LDS array of the loop) are done
  if (threadIdx.x == 0) *slot = atomicAdd(ctl.ticket, 1u) - ctl.ticket_base;
  __syncthreads();
  return *slot;
}

// ---- readers: the caller's input arrays -> device scratch, in order -------------------------------------------------------
// Up to three arrays travel together: chunk c of the call is bytes [c * per_chunk, (c + 1) * per_chunk) of EVERY array
// (per_chunk = the array's size / chunks, rounded up to 256 bytes: a chunk boundary is a cache-line boundary of the
// scratch, so no line of it is ever read before it is final), so that arrays consumed side by side -- frame boundaries and
// frames, records and payload pool -- arrive side by side.  flag[c] = this call's epoch once chunk c is in the scratch.
struct FeedSeg {
  const uint8_t* src;   // the caller's array as the device addresses it (16-byte aligned)
  uint8_t* dst;         // scratch (256-byte aligned, 16 bytes of slack behind it)
  uint64_t bytes, per_chunk;
};
struct InFeed {
  FeedSeg seg[3];
  unsigned long long* flag;  // [chunks]
  uint32_t chunks, readers;  // workgroups [0, readers) of the launch are readers
  // Chunks are CLAIMED, in order, by whichever reader workgroup is running (chunk = ticket - chunk_base; monotonic across calls
  // like the tile ticket).  Round 4 gave chunk c to reader c % readers: a reader that is not resident yet then owns chunks its
  // launch's workers spin for -- with ONE such kernel on the chip that cannot happen (readers are the launch's first
  // workgroups), with several handles' kernels resident together (the nodes of one process, 102 KB of LDS a decoder workgroup:
  // one per CU) it can, XCD by XCD, and two launches can then wait for each other's readers until the bounded waits give up.
  // Claimed chunks are always in the hands of a running workgroup; and where a launch has a worker resident on an XCD, its
  // readers there were dispatched before it.
  unsigned int* chunk_ticket;
  uint32_t chunk_base;
  uint32_t no_serve;  // the chunks are not the kernel's to bring in at all (RAFTQ_WIRE_SDMA: the runtime's copy engine does): workers only wait
};

// Agent-scope write-through of one 16-byte quad as ONE store instruction (`global_store_dwordx4 ... sc1`).  Round 4 split the
// quad into two 8-byte agent-scope atomic stores -- the widest store the atomic builtins express -- and every 8 bytes became a
// 32-byte write request: the scratch hop's HBM traffic was x2.5-x3 of its bytes (profiles/pmc_traffic_legs.json, r04).  Four
// lanes now fill a 64-byte request.  The compiler does not count an asm store in its vmcnt bookkeeping: whoever publishes the
// bytes waits with an explicit s_waitcnt vmcnt(0) (reader_role does, before the chunk's flag).  RAFTQ_WIRE_SC1_SPLIT builds
// the old form for an A/B.
__device__ __forceinline__ void sc1_store16(uint8_t* dst, u32x4 v) {
#if defined(RAFTQ_WIRE_SC1_SPLIT)
  unsigned long long lo = ((unsigned long long)v.y << 32) | v.x, hi = ((unsigned long long)v.w << 32) | v.z;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
#endif
}

// chunk c of the call, host -> scratch, by the whole workgroup; its flag once every byte has reached the coherence point
template <int TB = kBlock>  // threads of the workgroup
__device__ inline void feed_copy_chunk(const InFeed& in, uint32_t c, uint32_t epoch) {
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const FeedSeg sg = in.seg[k];
    const uint64_t lo = (uint64_t)c * sg.per_chunk;
    if (sg.bytes == 0 || lo >= sg.bytes) continue;
    const uint64_t len = sg.bytes - lo < sg.per_chunk ? sg.bytes - lo : sg.per_chunk;
    const u32x4* s16 = reinterpret_cast<const u32x4*>(sg.src + lo);
    uint8_t* d = sg.dst + lo;
    const uint64_t quads = len >> 4;
    uint64_t q = tid;
    for (; q + 3 * TB < quads; q += 4 * TB) {  // four pulls in flight per lane
      const u32x4 v0 = __builtin_nontemporal_load(s16 + q), v1 = __builtin_nontemporal_load(s16 + q + TB),
                  v2 = __builtin_nontemporal_load(s16 + q + 2 * TB), v3 = __builtin_nontemporal_load(s16 + q + 3 * TB);
      sc1_store16(d + (q << 4), v0);
      sc1_store16(d + ((q + TB) << 4), v1);
      sc1_store16(d + ((q + 2 * TB) << 4), v2);
      sc1_store16(d + ((q + 3 * TB) << 4), v3);
    }
    for (; q < quads; q += TB) sc1_store16(d + (q << 4), __builtin_nontemporal_load(s16 + q));
    const uint64_t done = quads << 4;  // (only an array's last chunk has a tail)
    if (tid < len - done) __hip_atomic_store(d + done + tid, sg.src[lo + done + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the coherence point ...
  __syncthreads();                                   // ... and so have the other waves'
  if (tid == 0) __hip_atomic_store(in.flag + c, lb_word(epoch, kLbInclusive, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int TB = kBlock>
__device__ inline void reader_role(const InFeed& in, uint32_t epoch) {
  __shared__ uint32_t chunk_slot;
  const uint32_t tid = threadIdx.x;
#if defined(RAFTQ_WIRE_STATIC_CHUNKS)  // round 4's assignment, for the A/B that shows what it does under several resident launches
  for (uint32_t c = blockIdx.x;; c += in.readers) {
    if (c >= in.chunks) return;
    (void)chunk_slot;  // (the ticket word is not used in this build)
#else
  for (;;) {
    __syncthreads();  // (the previous round's readers of chunk_slot are done)
    if (tid == 0) chunk_slot = atomicAdd(in.chunk_ticket, 1u) - in.chunk_base;
    __syncthreads();
    const uint32_t c = chunk_slot;
    if (c >= in.chunks) return;  // every reader workgroup draws exactly one ticket beyond the chunks
#endif
    feed_copy_chunk<TB>(in, c, epoch);
  }
}

// ---- a worker's wait for its input (round 6: liveness is structural) -------------------------------------------------------
// Round 5's workers only ever WAITED for chunks.  That is live as long as every unclaimed chunk will be claimed by a reader
// workgroup that gets to run -- a statement about the dispatcher (a launch's first workgroups become resident before its later
// ones; several launches resident together do not starve each other's readers), not about this code: with chunks owned by
// position two launches waited for each other's readers in 2 of 40 runs of the bench's node legs; with tickets 0 of 40 -- a
// bound of ~7 % on the failure rate, no more.  Now a worker that waits for chunk c watches the chunk
// ticket as well as the chunk's flag: when c is still UNCLAIMED and the ticket has not moved for kFeedStall looks (~100 us: a
// running reader claims a chunk every ~0.15 us) nobody resident is bringing input in, and the worker's whole workgroup takes
// the reader's role itself -- it claims the next unclaimed chunk (compare-and-swap: a worker never draws a ticket beyond the
// chunks, the host's count of the launch's draws stays chunks + readers), copies it, raises its flag, and goes on claiming
// until its own chunk is claimed.  So: a chunk is either claimed by a RUNNING workgroup (a reader, or a worker that serves
// itself; copying waits for nothing) or will be claimed by whoever waits for it; a tile is claimed by a running worker whose
// waits are for chunks and for tiles below its own -- by induction over the tile index no worker waits for work that no
// resident workgroup holds, whatever the dispatcher does, down to a launch with NO reader workgroups at all
// (RAFTQ_WIRE_READERS=0, tests/test_wire_gpu.py::test_streaming_codecs_without_readers).  The normal path is unchanged: the
// ticket moves all the time, no worker ever serves.  (RAFTQ_WIRE_STATIC_CHUNKS -- round 4's ownership by position, kept for
// the soak's A/B -- has no ticket to claim: it waits as it did, and still fails the soak.)
constexpr uint32_t kFeedLookEvery = 16, kFeedStall = 64;
constexpr uint32_t kFeedReady = 0, kFeedGaveUp = 1, kFeedServe = 2;  // kFeedServe + c: copy chunk c, then ask again
// ONE lane
__device__ inline uint32_t feed_poll(const InFeed& in, uint32_t epoch, uint32_t c, bool eager, unsigned int* stuck) {
  const unsigned long long* w = in.flag + c;
  uint32_t last_t = ~0u, still = 0;
  for (uint32_t spin = 0; spin < (1u << 23); ++spin) {
    const uint64_t s = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(s >> kLbEpochShift) == epoch && ((s >> kLbFlagShift) & 3u) != 0) return kFeedReady;
#if !defined(RAFTQ_WIRE_STATIC_CHUNKS)
    if (!in.no_serve && (eager || (spin & (kFeedLookEvery - 1)) == kFeedLookEvery - 1)) {
      const uint32_t raw = __hip_atomic_load(in.chunk_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t t = raw - in.chunk_base;  // chunks claimed so far (readers draw past the end: may exceed in.chunks)
      if (t > c) {
        still = 0;  // mine is in the hands of a running workgroup
        eager = false;
      } else if (!eager && t != last_t) {
        last_t = t;  // readers are at work
        still = 0;
      } else if (eager || ++still >= kFeedStall) {
        unsigned int expect = raw;
        if (__hip_atomic_compare_exchange_strong(in.chunk_ticket, &expect, raw + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          return kFeedServe + t;
        still = 0;  // somebody else moved it
      }
    }
#endif
    __builtin_amdgcn_s_sleep(2);
  }
  atomicOr(stuck, 1u);
  return kFeedGaveUp;
}
// The WHOLE workgroup (lo, hi workgroup-uniform): bytes [lo, hi) of array k are in the scratch.  Ends on a barrier.
// -> false (workgroup-uniform): a wait gave up (a fault: a lost workgroup, a corrupted control block) -- the bytes are NOT there,
// *stuck is raised, the call fails with RAFTQ_EHIP; the decoders then treat the tile's frames as unreadable instead of parsing
// whatever the scratch holds (raftq_step_frames would step from it)
template <int TB = kBlock>
__device__ inline bool feed_wait_chunks(const InFeed& in, uint32_t epoch, uint32_t c0, uint32_t c1, unsigned int* stuck, uint32_t* slot /*LDS*/) {
  bool eager = false, ok = true;  // (workgroup-uniform: set from *slot)
  for (uint32_t c = c0; c <= c1; ++c) {
    for (;;) {
      if (threadIdx.x == 0) *slot = feed_poll(in, epoch, c, eager, stuck);
      __syncthreads();
      const uint32_t s = *slot;
      __syncthreads();  // (*slot is written again)
      if (s == kFeedGaveUp) ok = false;
      if (s < kFeedServe) break;
      feed_copy_chunk<TB>(in, s - kFeedServe, epoch);  // ends on a barrier
      eager = true;  // nobody else is bringing input in: keep claiming until this wait's chunks are claimed
    }
  }
  return ok;
}
template <int TB = kBlock>
__device__ inline bool feed_wait(const InFeed& in, uint32_t epoch, int k, uint64_t lo, uint64_t hi, unsigned int* stuck, uint32_t* slot) {
  // (an array may go on, in the scratch, behind what the READERS bring: records an earlier kernel of the stream left there --
  // raftq_propose_frames -- are simply there)
  if (hi > in.seg[k].bytes) hi = in.seg[k].bytes;
  if (hi <= lo) {
    __syncthreads();
    return true;
  }
  const uint64_t per = in.seg[k].per_chunk;
  uint32_t c1 = (uint32_t)((hi - 1) / per);
  if (c1 >= in.chunks) c1 = in.chunks - 1;
  return feed_wait_chunks<TB>(in, epoch, (uint32_t)(lo / per), c1, stuck, slot);
}
template <int TB = kBlock>
__device__ inline bool feed_wait_all(const InFeed& in, uint32_t epoch, unsigned int* stuck, uint32_t* slot) {
  return feed_wait_chunks<TB>(in, epoch, 0, in.chunks - 1, stuck, slot);
}

// block-wide exclusive sum of one u32 per thread (kBlock threads) -> this thread's prefix; *total = the block's sum
template <int W = kWaves>
__device__ __forceinline__ uint64_t block_exclusive_u32(uint32_t v, uint64_t* wave_tot /*LDS [W]*/, uint64_t* total) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = __shfl_up(incl, o, 64);
    if (lane >= (uint32_t)o) incl += y;
  }
  if (lane == 63) wave_tot[w] = incl;
  __syncthreads();
  uint64_t pre = 0, all = 0;
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)W; ++k) {
    const uint64_t t = wave_tot[k];
    pre += k < w ? t : 0;
    all += t;
  }
  *total = all;
  return pre + incl - v;
}

// the tile's records leave through LDS: lane-consecutive 16-byte stores, 4 KB of consecutive host memory per instruction
template <typename Rec, int TB = kBlock>
__device__ __forceinline__ void tile_records_out(const Rec& mine, bool live, u32x4* lds /*[TB * sizeof(Rec) / 16]*/, Rec* out_h,
                                                 uint64_t tile0, uint64_t n, Rec* out_d = nullptr /* a copy that stays in HBM */) {
  static_assert(sizeof(Rec) % 16 == 0, "records are whole quads");
  constexpr uint32_t kQ = sizeof(Rec) / 16;
  if (live) {
    u32x4 q[kQ];
    __builtin_memcpy(q, &mine, sizeof(Rec));
#pragma unroll
    for (uint32_t k = 0; k < kQ; ++k) lds[threadIdx.x * kQ + k] = q[k];
  }
  __syncthreads();
  const uint64_t live_recs = n - tile0 < (uint64_t)TB ? n - tile0 : (uint64_t)TB;
  const uint32_t quads = (uint32_t)live_recs * kQ;
  u32x4* dst = reinterpret_cast<u32x4*>(out_h + tile0);
#pragma unroll
  for (uint32_t k = 0; k < kQ; ++k) {
    const uint32_t q = k * TB + threadIdx.x;
    if (q < quads) __builtin_nontemporal_store(lds[q], dst + q);
  }
  if (out_d != nullptr) {
    u32x4* dd = reinterpret_cast<u32x4*>(out_d + tile0);
#pragma unroll
    for (uint32_t k = 0; k < kQ; ++k) {
      const uint32_t q = k * TB + threadIdx.x;
      if (q < quads) dd[q] = lds[q];
    }
  }
}

// A wave's run of frames on its way from the scratch into LDS with every request issued before the first one is waited
// for: global -> LDS DMA (global_load_lds, 16 bytes per lane, 1 KB of consecutive stream per instruction, no VGPR round
// trip, agent-scope so that no stale line of this XCD's L2 can answer); the caller waits with s_waitcnt vmcnt(0).
// `readable`: bytes of the scratch copy that may be read (the stream rounded up to whole quads: the scratch has the slack).
constexpr int kAuxSc1 = 16;  // cache-policy operand of the DMA builtin: sc1 (agent scope) on gfx94x / gfx950
constexpr uint32_t kEntQ = 4;  // entry headers a lane keeps in LDS from its one walk of a frame (more: the frame is walked again)
__device__ __forceinline__ WaveStage stage_wave_frames_dma(const uint8_t* stream, uint64_t nbytes, uint64_t readable, uint64_t a, uint64_t b,
                                                           bool live, uint32_t* lds /* this wave's kStageBytes */) {
  WaveStage st;
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t alive = __ballot(live);
  if (alive == 0) return st;
  const uint64_t lo = wave_bcast_u64(a, 0);  // lane 0 is live whenever any lane is (lanes fill from the front)
  const uint64_t hi = wave_bcast_u64(b, 63 - __builtin_clzll(alive));
  const uint64_t lo16 = lo & ~15ull;
  if (!(lo <= hi && hi <= nbytes) || hi - lo16 + 16 > kStageBytes) return st;  // wave-uniform
  const uint64_t full = (hi - lo16 + 15) >> 4;  // quads that cover the run
  if (lo16 + (full << 4) > readable) return st;
  const uint32_t chunks = (uint32_t)((full + 63) >> 6);  // instructions of 64 quads; lanes past the run re-read its first quad
  for (uint32_t c = 0; c < chunks; ++c) {
    const uint64_t q = (uint64_t)c * 64 + lane;
    const uint8_t* g = stream + (q < full ? lo16 + (q << 4) : lo16);
    __builtin_amdgcn_global_load_lds((global_cvoid_t*)g, (lds_void_t*)(reinterpret_cast<uint8_t*>(lds) + c * 1024), 16, 0, kAuxSc1);
  }
  st.words = lds;
  st.lo16 = lo16;
  st.lo = lo;
  st.hi = hi;
  return st;
}

// -DRAFTQ_WIRE_TRACE (measurement builds only): lane 0 of every tile leaves wall-clock stamps (100 MHz) of its phases in
// the spare status array, raftq_wire.hip prints them after the call; RAFTQ_WIRE_ABLATE drops outputs
#if defined(RAFTQ_WIRE_TRACE)
#define RAFTQ_TRACE_STAMP(ctl, tile, k) \
  do { if (threadIdx.x == 0) (ctl).status[kLbSpare][(uint64_t)(tile) * 8 + (k)] = wall_clock64(); } while (0)
#define RAFTQ_ABLATE(ctl, bit) (((ctl).ablate >> (bit)) & 1u)
#else
#define RAFTQ_TRACE_STAMP(ctl, tile, k) do { } while (0)
#define RAFTQ_ABLATE(ctl, bit) false
#endif

// raftq_wire_decode on page-locked buffers: everything in one launch (see above).  Readers bring the frame boundaries
// (array 0) and the stream (array 1) into the scratch; msgs_h / ents_h are the caller's result arrays as the device
// addresses them.  pin[0] = entries found, pin[1] = malformed frames (written by the worker of the last tile, whose
// inclusive sums are the totals), pin[3] = a wait gave up.
//
// raftq_step_frames (a node's inbound half-turn as one submission): `ff.on` -- the decoder also makes the checks a node makes on
// what it received and says in the record's flag byte what Step is to do with it (RAFTQ_MSGF_*, raftq_step.h): a frame that did
// not parse, is of a kind a peer never sends, names no group / sender of this cluster or is addressed to another slot is
// RAFTQ_MSGF_SKIP; a MsgProp (the log owner's) RAFTQ_MSGF_HOLD; a MsgApp a barrier that says what it carries (reject_hint, which
// MsgApp does not use = the last entry's term).  msgs_d: the same records once more, in HBM, where Step's kernels -- enqueued
// right behind this one -- read them.
struct FrameFilter {
  uint32_t on, n_peers, self, tail_appends;
  uint64_t n_groups;
  unsigned long long* zero2;  // two words this kernel leaves zero for the kernels behind it (Step's {touched groups, bad | skipped}), or nullptr
};
constexpr uint8_t kFrameSkip = 0x10, kFrameHold = 0x20, kFrameBarrier = 0x40, kFrameEntries = 0x80;  // == RAFTQ_MSGF_*
// TB (round 6): threads of a workgroup = frames of a tile.  Round 5's workgroup was 256 frames with 102 KB of LDS -- the scalar
// fields' file (34 KB), the frames' stage (32 KB), four entry headers per lane (32 KB): ONE workgroup per CU, one wave per SIMD,
// and every reader workgroup of the launch paid for the same 102 KB.  Now the entry headers a lane meets on its one walk go to
// a slot of its own in device scratch (`ent_spill`: 15 % of the lanes write one to three 32-byte headers; they are read back,
// L2-resident, when the tile's run is gathered behind the look-back), and the tile is a template parameter: 128 frames =
// 34.3 KB (four workgroups per CU, every tile of a 64K-frame call resident at once, each waiting for its own bytes);
// 256 frames = 68.5 KB (two per CU).  RAFTQ_WIRE_TILE picks.  MEASURED (profiles/r06/wire_tile_ab.jsonl, 64K frames, one box, one
// process): 256 frames 172 us a call, 128 frames 186 -- residency is not what bounds a call (every tile has a worker waiting
// for its bytes either way; the call is its input over the link plus ONE tile's chain), and twice the tiles are twice the
// look-back words, barriers and status traffic.  256 is the default; both are tested.
template <int TB>
static __global__ __launch_bounds__(TB) void wire_dec_fused_kernel(InFeed in, uint64_t nbytes, uint64_t n, WireMsg* msgs_h, WireEnt* ents_h,
                                                                   uint64_t ents_cap, TileCtl ctl, uint64_t* __restrict__ pin,
                                                                   WireMsg* msgs_d, FrameFilter ff, WireEnt* __restrict__ ent_spill) {
  constexpr int W = TB / 64;
  if (blockIdx.x < in.readers) {
    reader_role<TB>(in, ctl.epoch);
    return;
  }
  __shared__ __attribute__((aligned(16))) uint64_t file[kFileSlots * TB];  // 136 B per frame; the records' way out afterwards (64 B per frame)
  __shared__ __attribute__((aligned(16))) uint32_t stage[W][kStageBytes / 4];  // the frames; the entry headers' way out afterwards
  __shared__ uint64_t offs[TB + 1];
  __shared__ uint64_t wave_tot[W];
  __shared__ uint64_t prefix[2];
  __shared__ uint32_t wave_bad[W];
  __shared__ uint32_t tile_slot, feed_slot;
#if defined(RAFTQ_WIRE_LDS_PAD)
  // measurement builds only (tools/gpurun_trip.sh soakpad*): round 5's LDS footprint back -- 102 KB, ONE workgroup per CU -- so that
  // several handles' launches oversubscribe the chip again and the soak meets the residency that starved round 4's readers
  __shared__ uint32_t lds_pad[RAFTQ_WIRE_LDS_PAD / 4];
  if (n == ~0ull) {  // (never true: a store and a load the compiler cannot fold keep the array)
    lds_pad[(threadIdx.x * 977u + ctl.epoch) % (RAFTQ_WIRE_LDS_PAD / 4)] = threadIdx.x;
    __syncthreads();
    pin[4 + (threadIdx.x & 1)] = lds_pad[(threadIdx.x * 331u + ctl.ticket_base) % (RAFTQ_WIRE_LDS_PAD / 4)];
  }
#endif
  const uint64_t* off = reinterpret_cast<const uint64_t*>(in.seg[0].dst);
  const uint8_t* stream = in.seg[1].dst;
  const uint64_t readable = (nbytes + 15) & ~15ull;
  const uint32_t n_tiles = (uint32_t)((n + TB - 1) / TB);
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  unsigned int* stuck = ctl.ticket + 1;
  if (ff.zero2 != nullptr && blockIdx.x == in.readers && tid < 2) ff.zero2[tid] = 0;  // (saves Step a launch of its own in front of this one)
  for (;;) {
    const uint32_t cur = next_tile(ctl, &tile_slot);
    if (cur >= n_tiles) return;
    RAFTQ_TRACE_STAMP(ctl, cur, 0);
    const uint64_t tile0 = (uint64_t)cur * TB, i = tile0 + tid;
    const uint64_t last = tile0 + TB < n ? tile0 + TB : n;
    bool fed = feed_wait<TB>(in, ctl.epoch, 0, tile0 * 8, (last + 1) * 8, stuck, &feed_slot);  // the tile's TB + 1 boundaries are in the scratch
    if (i <= n) offs[tid] = __hip_atomic_load(off + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) offs[TB] = __hip_atomic_load(off + last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    RAFTQ_TRACE_STAMP(ctl, cur, 1);
    const bool live = i < n;
    const uint64_t a = live ? offs[tid] : 0, b = live ? offs[tid + 1] : 0;
    // ... and so are its frames: [first boundary, last boundary) when the boundaries ascend inside the buffer, else (garbage
    // boundaries: every lane may look anywhere) the whole stream
    const bool ordered = __syncthreads_and(!live || (a <= b && b <= nbytes));
    if (ordered) fed &= feed_wait<TB>(in, ctl.epoch, 1, offs[0], offs[last - tile0], stuck, &feed_slot);
    else fed &= feed_wait_all<TB>(in, ctl.epoch, stuck, &feed_slot);
    const WaveStage st = stage_wave_frames_dma(stream, nbytes, readable, a, b, live && fed, stage[wave]);
    RAFTQ_TRACE_STAMP(ctl, cur, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAFTQ_TRACE_STAMP(ctl, cur, 3);
    WireMsg m;
    bool malformed = false;
    LdsFileT<TB> f{file + tid};
    ByteSrc src = {stream, 0, nullptr, 0};
    WireEnt* my_ents = ent_spill + ((uint64_t)cur * TB + tid) * kEntQ;
    if (live) {
      src = frame_src(st, stream, nbytes, a, b);
      bool ok = fed && frame_body_staged(src, stream, nbytes, a, b, true);  // (a wait that gave up: the bytes are not there -- nothing is parsed)
      // ONE walk: the first kEntQ entry headers of the frame are left in the lane's scratch slot on the way (a second walk cost every tile 11 us)
      if (ok) ok = parse_msg<true>(src, b - a - 8, a + 8, f, m, my_ents, 0, kEntQ, 0xffffffffu);
      if (!ok) {
        m.group = m.term = m.log_term = m.index = m.commit = m.reject_hint = 0;
        m.from = 0;
        m.type = m.reject = m.to = 0;
        m.ent_first = m.n_ents = 0;
        m.flags = kWireMalformed;
        malformed = true;
      }
      if (ff.on) {
        const uint8_t t = m.type;  // MsgProp 2, MsgApp 3, MsgAppResp 4, MsgVote 5, MsgVoteResp 6, MsgHeartbeat 8, MsgHeartbeatResp 9
        const bool kind_ok = t == 2 || t == 3 || t == 4 || t == 5 || t == 6 || t == 8 || t == 9;
        if (malformed || !kind_ok || m.group >= ff.n_groups || m.from >= ff.n_peers || m.to != ff.self) {
          m.flags |= kFrameSkip;
        } else if (t == 2) {
          m.flags |= kFrameHold;
        } else if (t == 3) {
          m.flags |= kFrameBarrier | (ff.tail_appends ? kFrameEntries : 0);
          m.reject_hint = m.n_ents ? f.get(14) : 0;  // (slot 14: Term of the Entry walked last)
        }
      }
    }
    const uint32_t cnt = live ? m.n_ents : 0u;
    const uint64_t mb = __ballot(malformed);
    if ((tid & 63) == 0) wave_bad[wave] = (uint32_t)__popcll(mb);
    uint64_t tile_ents;
    const uint64_t local = block_exclusive_u32<W>(cnt, wave_tot, &tile_ents);  // (its barrier publishes wave_bad too)
    RAFTQ_TRACE_STAMP(ctl, cur, 4);
    if (tid == 0) {
      uint32_t tile_bad = 0;
      for (int k = 0; k < W; ++k) tile_bad += wave_bad[k];
      const uint64_t pe = lb_exclusive(ctl.status[0], ctl.epoch, cur, tile_ents, stuck);
      const uint64_t pb = lb_exclusive(ctl.status[1], ctl.epoch, cur, tile_bad, stuck);
      prefix[0] = pe;
      if (cur == n_tiles - 1) {
        pin[0] = pe + tile_ents;
        pin[1] = pb + tile_bad;
        pin[3] = __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a give-up elsewhere after this shows in the next call at the latest)
      }
    }
    __syncthreads();
    RAFTQ_TRACE_STAMP(ctl, cur, 5);
    const uint64_t first = prefix[0] + local;
    if (cnt != 0) m.ent_first = (uint32_t)first;
    if (cnt > kEntQ && ents_h != nullptr) {  // a frame with more entries than a lane keeps: walked again, headers straight out
      WireMsg again;
      (void)parse_msg<true>(src, b - a - 8, a + 8, f, again, ents_h, first, ents_cap, cnt);
    }
    // The tile's entry headers are one contiguous run of the caller's array: gathered in LDS (the frames' stage is free now)
    // and pushed out like the records, whole 64-byte lines.  A tile with a frame that was walked twice writes them lane by lane.
    constexpr uint32_t kEntStage = W * kStageBytes / sizeof(WireEnt);
    static_assert(kEntStage >= TB * kEntQ, "the stage holds every header a tile's lanes can keep");
    WireEnt* ent_run = reinterpret_cast<WireEnt*>(&stage[0][0]);
    const bool mine_kept = cnt != 0 && cnt <= kEntQ && ents_h != nullptr;
    const bool run_staged = __syncthreads_and(cnt <= kEntQ);  // (the barrier: nobody files fields or reads frames any more)
    if (mine_kept && run_staged) {
#pragma unroll
      for (uint32_t k = 0; k < kEntQ; ++k)
        if (k < cnt) ent_run[local + k] = my_ents[k];
    }
    tile_records_out<WireMsg, TB>(m, live && !RAFTQ_ABLATE(ctl, 0), reinterpret_cast<u32x4*>(file), RAFTQ_ABLATE(ctl, 0) ? msgs_h - tile0 : msgs_h, tile0,
                                  RAFTQ_ABLATE(ctl, 0) ? tile0 + 1 : n, msgs_d);  // (its barrier publishes ent_run too)
    RAFTQ_TRACE_STAMP(ctl, cur, 6);
    if (ents_h != nullptr && tile_ents != 0 && !RAFTQ_ABLATE(ctl, 1)) {
      const uint64_t run0 = prefix[0];
      if (run_staged) {
        const uint64_t room = run0 < ents_cap ? ents_cap - run0 : 0;
        const uint32_t quads = (uint32_t)(tile_ents < room ? tile_ents : room) * 2;
        const u32x4* src_q = reinterpret_cast<const u32x4*>(ent_run);
        u32x4* dst = reinterpret_cast<u32x4*>(ents_h + run0);
        for (uint32_t q = tid; q < quads; q += TB) __builtin_nontemporal_store(src_q[q], dst + q);
      } else if (mine_kept) {
        for (uint32_t k = 0; k < cnt; ++k)
          if (first + k < ents_cap) ents_h[first + k] = my_ents[k];
      }
    }
    RAFTQ_TRACE_STAMP(ctl, cur, 7);
  }
}

// ---- kernels: walpb.Record ---------------------------------------------------------------------------

__device__ __forceinline__ bool wal_has_payload(uint8_t kind) { return kind == kWalEntry || kind == kWalMetadata; }

// bytes of Record.data
__device__ __forceinline__ uint64_t wal_data_size(const WalRec& r) {
  switch (r.kind) {
    case kWalEntry: return entry_size(r.entry_type, r.term, r.index, r.data_len) + 1 + sov(r.group);
    case kWalState: return 4 + sov(r.term) + sov(r.vote) + sov(r.index) + sov(r.group);
    case kWalSnapshot: return 2 + sov(r.index) + sov(r.term);
    case kWalMetadata: return r.data_len;
    default: return 0;
  }
}

// encode, step 0: one wave per record with a long payload -> pcrc[i] = crc32c(payload)
static __global__ __launch_bounds__(kBlock) void wal_enc_payload_crc_kernel(const WalRec* __restrict__ recs, uint64_t n,
                                                                            const uint8_t* __restrict__ pool,
                                                                            uint64_t pool_bytes,
                                                                            uint32_t* __restrict__ pcrc) {
  __shared__ uint32_t tab[kCrcTabs * 256];
  const uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const uint32_t len = i < n ? recs[i].data_len : 0u;
  const uint64_t o = i < n ? recs[i].data_off : 0ull;
  // (flagged by wal_enc_crc_kernel when the payload lies outside the pool)
  const bool work = i < n && len > kCoopBytes && wal_has_payload(recs[i].kind) && !(o > pool_bytes || len > pool_bytes - o);
  if (!__syncthreads_or(work)) return;  // no wave of this workgroup has a long payload: not even the tables are built
  crc_table_init(tab);
  if (!work) return;
  const uint32_t c = wave_crc(tab, pool + o, len);
  if ((threadIdx.x & 63) == 0) pcrc[i] = c;
}

// encode, step 1: pair[i] = the map record i applies to the running CRC (record 0 has prev_crc folded in)
static __global__ __launch_bounds__(kBlock) void wal_enc_crc_kernel(const WalRec* __restrict__ recs, uint64_t n,
                                                                    const uint8_t* __restrict__ pool,
                                                                    uint64_t pool_bytes,
                                                                    const uint32_t* __restrict__ pcrc,
                                                                    uint32_t prev_crc, CrcPair* __restrict__ pair,
                                                                    unsigned int* bad) {
  __shared__ uint32_t tab[kCrcTabs * 256];
  crc_table_init(tab);
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool is_bad = false;
  if (i < n) {
    const WalRec r = recs[i];
    is_bad = r.kind < 1 || r.kind > 5;
    const bool payload = !is_bad && wal_has_payload(r.kind) && r.data_len != 0;
    if (payload && (r.data_off > pool_bytes || r.data_len > pool_bytes - r.data_off)) is_bad = true;
    CrcPair me = {0u, 0x80000000u};
    if (!is_bad) {
      uint32_t raw = 0xffffffffu;
      // bytes in front of the payload
      if (r.kind == kWalEntry) {
        raw = crc_field(tab, raw, 0x08, r.entry_type);
        raw = crc_field(tab, raw, 0x10, r.term);
        raw = crc_field(tab, raw, 0x18, r.index);
        if (r.data_len) raw = crc_field(tab, raw, 0x22, r.data_len);
      } else if (r.kind == kWalState) {
        raw = crc_field(tab, raw, 0x08, r.term);
        raw = crc_field(tab, raw, 0x10, r.vote);
        raw = crc_field(tab, raw, 0x18, r.index);
        raw = crc_field(tab, raw, 0x20, r.group);
      } else if (r.kind == kWalSnapshot) {
        raw = crc_field(tab, raw, 0x08, r.index);
        raw = crc_field(tab, raw, 0x10, r.term);
      }
      if (payload) {
        if (r.data_len > kCoopBytes) {
          const uint32_t joined = crc_mulmod(crc_xpow8(r.data_len), ~raw) ^ pcrc[i];  // crc(front || payload)
          raw = ~joined;
        } else {
          raw = crc_span(tab, raw, pool + r.data_off, r.data_len);
        }
      }
      if (r.kind == kWalEntry) raw = crc_field(tab, raw, 0x28, r.group);
      const uint64_t dsz = wal_data_size(r);
      me.c = dsz ? ~raw : 0u;
      me.p = crc_xpow8(dsz);
      if (i == 0) me.c ^= crc_mulmod(me.p, prev_crc);
    }
    pair[i] = me;
  }
  if (__ballot(is_bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// Record.Size() given the chained crc
__device__ __forceinline__ uint64_t wal_rec_size(const WalRec& r, uint32_t crc, uint64_t dsz) {
  const bool has_data = r.kind != kWalCrc && !(r.kind == kWalMetadata && dsz == 0);
  return 2 + sov(r.kind) + sov(crc) + (has_data ? 1 + sov(dsz) + dsz : 0);
}

// encode, step 2: frame sizes (they depend on the chained CRC's varint length); sizes[n] = 0
static __global__ __launch_bounds__(kBlock) void wal_enc_size_kernel(const WalRec* __restrict__ recs, uint64_t n,
                                                                     const CrcPair* __restrict__ chain,
                                                                     uint64_t* __restrict__ sizes) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    const WalRec r = recs[i];
    sizes[i] = 8 + wal_rec_size(r, chain[i].c, wal_data_size(r));
  } else if (i == n) {
    sizes[i] = 0;
  }
}

// encode, step 3: every byte except the payloads; *last_crc = the chain after the last record
static __global__ __launch_bounds__(kBlock) void wal_enc_write_kernel(const WalRec* __restrict__ recs, uint64_t n,
                                                                      const CrcPair* __restrict__ chain,
                                                                      const uint64_t* __restrict__ off,
                                                                      uint8_t* __restrict__ out,
                                                                      uint32_t* __restrict__ last_crc) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const WalRec r = recs[i];
  const uint32_t crc = chain[i].c;
  if (i == n - 1) *last_crc = crc;
  const uint64_t dsz = wal_data_size(r);
  Sink s(out + off[i]);
  s.u64_le(wal_rec_size(r, crc, dsz));
  s.field(0x08, r.kind);
  s.field(0x10, crc);
  const bool has_data = r.kind != kWalCrc && !(r.kind == kWalMetadata && dsz == 0);
  if (has_data) {
    s.field(0x1a, dsz);
    if (r.kind == kWalEntry) {
      s.field(0x08, r.entry_type);
      s.field(0x10, r.term);
      s.field(0x18, r.index);
      if (r.data_len) {
        s.field(0x22, r.data_len);
        s.skip(r.data_len);
      }
      s.field(0x28, r.group);
    } else if (r.kind == kWalState) {
      s.field(0x08, r.term);
      s.field(0x10, r.vote);
      s.field(0x18, r.index);
      s.field(0x20, r.group);
    } else if (r.kind == kWalSnapshot) {
      s.field(0x08, r.index);
      s.field(0x10, r.term);
    }  // metadata: payload only
  }
  s.flush();
}

// encode, step 4: one wave per record, payload pool -> frame
static __global__ __launch_bounds__(kBlock) void wal_enc_payload_kernel(const WalRec* __restrict__ recs, uint64_t n,
                                                                        const CrcPair* __restrict__ chain,
                                                                        const uint64_t* __restrict__ off,
                                                                        const uint8_t* __restrict__ pool,
                                                                        uint8_t* __restrict__ out) {
  const uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (i >= n) return;
  if (recs[i].data_len == 0 || !wal_has_payload(recs[i].kind)) return;
  const WalRec r = recs[i];
  const uint64_t dsz = wal_data_size(r);
  uint64_t pos = off[i] + 8 + 2 + sov(r.kind) + sov(chain[i].c) + 1 + sov(dsz);
  if (r.kind == kWalEntry) pos += 3 + sov(r.entry_type) + sov(r.term) + sov(r.index) + 1 + sov(r.data_len);
  wave_copy(out + pos, pool + r.data_off, r.data_len);
}

struct WalSpan {  // Record.data of record i inside the input buffer
  uint64_t off, len;
};

// decode, step 1: parse; CRC the short Data spans in the lane; pair[i] = the record's map.  Round 3: the frames of a wave
// are staged in LDS (stage_wave_frames), parsed there by the flat one-window-per-field walk and CRC'd there.
static __global__ __launch_bounds__(kBlock) void wal_dec_kernel(const uint8_t* __restrict__ bytes, uint64_t nbytes,
                                                                const uint64_t* __restrict__ off, uint64_t n,
                                                                uint32_t prev_crc, WalRec* __restrict__ recs,
                                                                WalSpan* __restrict__ span,
                                                                CrcPair* __restrict__ pair) {
  __shared__ uint32_t tab[kCrcTabs * 256];
  __shared__ __attribute__((aligned(16))) uint32_t stage[kWaves][kStageBytes / 4];
  crc_table_init(tab);
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const uint64_t a = i < n ? off[i] : 0, b = i < n ? off[i + 1] : 0;
  const WaveStage st = stage_wave_frames(bytes, nbytes, a, b, i < n, stage[threadIdx.x >> 6]);
  __syncthreads();
  if (i >= n) return;
  WalRec r;
  uint64_t d_off = 0, d_len = 0;
  const ByteSrc src = frame_src(st, bytes, nbytes, a, b);
  bool ok = frame_body_staged(src, bytes, nbytes, a, b, false);
  if (ok) ok = parse_wal_rec(src, b - a - 8, a + 8, r, d_off, d_len);
  CrcPair me = {0u, 0x80000000u};  // a record that does not parse leaves the chain alone
  WalSpan sp = {0, 0};
  if (!ok) {
    r.group = r.term = r.index = r.data_off = 0;
    r.data_len = r.vote = r.crc = 0;
    r.kind = r.entry_type = r.pad = 0;
    r.flags = kWalMalformed;
  } else if (r.kind == kWalCrc) {
    me.c = r.crc;  // re-seed: the constant map
    me.p = 0;
  } else {
    sp.off = a + 8 + d_off;
    sp.len = d_len;
    me.p = crc_xpow8(d_len);
    if (d_len != 0 && d_len <= kCoopBytes) me.c = ~crc_span8(tab, 0xffffffffu, src, d_off, d_len);
  }
  if (i == 0) me.c ^= crc_mulmod(me.p, prev_crc);
  recs[i] = r;
  span[i] = sp;
  pair[i] = me;
}

// decode, step 2: one wave per record whose Data is long
static __global__ __launch_bounds__(kBlock) void wal_dec_long_crc_kernel(const uint8_t* __restrict__ bytes, uint64_t n,
                                                                         const WalSpan* __restrict__ span,
                                                                         uint32_t prev_crc,
                                                                         CrcPair* __restrict__ pair) {
  __shared__ uint32_t tab[kCrcTabs * 256];
  const uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  WalSpan sp = {0, 0};
  if (i < n) sp = span[i];
  const bool work = sp.len > kCoopBytes;
  if (!__syncthreads_or(work)) return;  // no wave of this workgroup has a long Data span
  crc_table_init(tab);
  if (!work) return;
  uint32_t c = wave_crc(tab, bytes + sp.off, sp.len);
  if ((threadIdx.x & 63) == 0) {
    if (i == 0) c ^= crc_mulmod(pair[0].p, prev_crc);
    pair[i].c = c;
  }
}

// decode, step 3: compare every stored crc with the chain; first_bad = min index of a bad record
static __global__ __launch_bounds__(kBlock) void wal_dec_check_kernel(WalRec* __restrict__ recs, uint64_t n,
                                                                      const CrcPair* __restrict__ chain,
                                                                      uint32_t prev_crc,
                                                                      unsigned long long* first_bad) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const uint8_t flags = recs[i].flags;
    bad = (flags & kWalMalformed) != 0;
    if (!bad) {
      const uint32_t stored = recs[i].crc;
      if (recs[i].kind == kWalCrc) {
        const uint32_t before = i == 0 ? prev_crc : chain[i - 1].c;  // decoder.crc.Sum32() when the record arrives
        bad = before != 0 && stored != before;
      } else {
        bad = stored != chain[i].c;
      }
      if (bad) recs[i].flags = flags | kWalBadCrc;
    }
  }
  const uint64_t bb = __ballot(bad);
  if (bad && (uint64_t)__ffsll((long long)bb) - 1 == (threadIdx.x & 63)) atomicMin(first_bad, (unsigned long long)i);
}

// decode, step 4: {n_valid, last_crc} behind the records
static __global__ void wal_dec_tail_kernel(const CrcPair* __restrict__ chain, uint64_t n, uint32_t prev_crc,
                                           const unsigned long long* __restrict__ first_bad,
                                           uint64_t* __restrict__ tail) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint64_t fb = *first_bad < n ? *first_bad : n;
  tail[0] = fb;
  tail[1] = fb == 0 ? prev_crc : chain[fb - 1].c;
}

// ---- the streaming form of the WAL codecs (see "the streaming form of a codec call") -------------------------------------

// The running CRC across tiles: the look-back's value is an affine map (CrcPair: 64 bits), so a tile publishes it as two
// status words (c and p, each with epoch and flag); a reader takes a pair only when both words carry the same flag -- a
// tile writes each word at most twice (aggregate, then inclusive), so a torn read is seen and repeated.
__device__ inline CrcPair lb_exclusive_crc(unsigned long long* st_c, unsigned long long* st_p, uint32_t epoch, uint32_t tile, CrcPair aggregate,
                                           unsigned int* stuck) {
  CrcCompose op;
  auto put = [&](uint32_t flag, CrcPair v) {
    __hip_atomic_store(st_c + tile, lb_word(epoch, flag, v.c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(st_p + tile, lb_word(epoch, flag, v.p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (tile == 0) {
    put(kLbInclusive, aggregate);
    return kCrcIdentity;
  }
  put(kLbAggregate, aggregate);
  CrcPair prefix = kCrcIdentity;  // the composition of the tiles before `tile` seen so far (they come BEFORE what is in it)
  for (uint32_t j = tile; j-- > 0;) {
    uint64_t wc = 0, wp = 0;
    for (uint32_t tries = 0; tries < (1u << 16); ++tries) {
      wc = lb_wait(st_c + j, epoch, stuck);
      wp = lb_wait(st_p + j, epoch, stuck);
      if (((wc ^ wp) >> kLbFlagShift & 3u) == 0) break;
    }
    const CrcPair v = {(uint32_t)wc, (uint32_t)wp};
    prefix = op(v, prefix);
    if (((wc >> kLbFlagShift) & 3u) == kLbInclusive) break;
  }
  put(kLbInclusive, op(prefix, aggregate));
  return prefix;
}

// the look-back for "index of the first bad record so far" (a running minimum; kLbValueMask = none)
__device__ inline uint64_t lb_exclusive_min(unsigned long long* status, uint32_t epoch, uint32_t tile, uint64_t mine, unsigned int* stuck) {
  if (tile == 0) {
    __hip_atomic_store(status, lb_word(epoch, kLbInclusive, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return kLbValueMask;
  }
  __hip_atomic_store(status + tile, lb_word(epoch, kLbAggregate, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t before = kLbValueMask;
  for (uint32_t j = tile; j-- > 0;) {
    const uint64_t s = lb_wait(status + j, epoch, stuck);
    const uint64_t v = s & kLbValueMask;
    before = v < before ? v : before;
    if (((s >> kLbFlagShift) & 3u) == kLbInclusive) break;
  }
  __hip_atomic_store(status + tile, lb_word(epoch, kLbInclusive, mine < before ? mine : before), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return before;
}

// raftq_wal_decode on page-locked buffers, one launch: readers bring the frame boundaries (array 0) and the WAL bytes
// (array 1) into the scratch; a worker parses its tile's records from LDS, CRCs their Data (short spans in the lane, long
// ones by the wave), chains the CRCs through the look-back, compares, and pushes the 48-byte records out.
// pin[0] = records before the first bad one, pin[1] = the running CRC there, pin[3] = a wait gave up.
static __global__ __launch_bounds__(kBlock) void wal_dec_fused_kernel(InFeed in, uint64_t nbytes, uint64_t n, uint32_t prev_crc, WalRec* recs_h,
                                                                      TileCtl ctl, uint64_t* __restrict__ pin) {
  if (blockIdx.x < in.readers) {
    reader_role(in, ctl.epoch);
    return;
  }
  __shared__ uint32_t tab[kCrcTabs * 256];
  __shared__ __attribute__((aligned(16))) uint32_t stage[kWaves][kStageBytes / 4];  // the frames; the records' way out afterwards (12 KB)
  __shared__ uint64_t offs[kBlock + 1];
  __shared__ uint32_t chain_of[kBlock];
  __shared__ CrcPair wave_tot[kWaves];
  __shared__ CrcPair tile_pre;
  __shared__ unsigned long long wave_min[kWaves];
  __shared__ unsigned long long bad_before;
  __shared__ uint32_t tile_slot, feed_slot;
  const uint64_t* off = reinterpret_cast<const uint64_t*>(in.seg[0].dst);
  const uint8_t* bytes = in.seg[1].dst;
  const uint64_t readable = (nbytes + 15) & ~15ull;
  const uint32_t n_tiles = (uint32_t)((n + kBlock - 1) / kBlock);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned int* stuck = ctl.ticket + 1;
  CrcCompose op;
  crc_table_init(tab);
  for (;;) {
    const uint32_t cur = next_tile(ctl, &tile_slot);
    if (cur >= n_tiles) return;
    const uint64_t tile0 = (uint64_t)cur * kBlock, i = tile0 + tid;
    const uint64_t last = tile0 + kBlock < n ? tile0 + kBlock : n;
    feed_wait(in, ctl.epoch, 0, tile0 * 8, (last + 1) * 8, stuck, &feed_slot);
    if (i <= n) offs[tid] = __hip_atomic_load(off + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) offs[kBlock] = __hip_atomic_load(off + last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool live = i < n;
    const uint64_t a = live ? offs[tid] : 0, b = live ? offs[tid + 1] : 0;
    const bool ordered = __syncthreads_and(!live || (a <= b && b <= nbytes));
    if (ordered) feed_wait(in, ctl.epoch, 1, offs[0], offs[last - tile0], stuck, &feed_slot);
    else feed_wait_all(in, ctl.epoch, stuck, &feed_slot);
    const WaveStage st = stage_wave_frames_dma(bytes, nbytes, readable, a, b, live, stage[wave]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WalRec r;
    uint64_t d_off = 0, d_len = 0, span_off = 0;
    CrcPair me = kCrcIdentity;  // a record that does not parse leaves the chain alone
    bool long_span = false;
    if (live) {
      const ByteSrc src = frame_src(st, bytes, nbytes, a, b);
      bool ok = frame_body_staged(src, bytes, nbytes, a, b, false);
      if (ok) ok = parse_wal_rec(src, b - a - 8, a + 8, r, d_off, d_len);
      if (!ok) {
        r.group = r.term = r.index = r.data_off = 0;
        r.data_len = r.vote = r.crc = 0;
        r.kind = r.entry_type = r.pad = 0;
        r.flags = kWalMalformed;
      } else if (r.kind == kWalCrc) {
        me.c = r.crc;  // re-seed: the constant map
        me.p = 0;
      } else {
        me.p = crc_xpow8(d_len);
        span_off = a + 8 + d_off;
        if (d_len != 0 && d_len <= kCoopBytes) me.c = ~crc_span8(tab, 0xffffffffu, src, d_off, d_len);
        long_span = d_len > kCoopBytes;
      }
    }
    for (uint64_t todo = __ballot(long_span); todo != 0; todo &= todo - 1) {  // long Data spans: the whole wave per span
      const int l = __ffsll((long long)todo) - 1;
      const uint64_t so = wave_bcast_u64(span_off, l), sl = wave_bcast_u64(d_len, l);
      const uint32_t c = wave_crc(tab, bytes + so, sl);  // (valid in lane 0)
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(c);
      if ((int)lane == l) me.c = c0;
    }
    if (i == 0) me.c ^= crc_mulmod(me.p, prev_crc);
    CrcPair tile_tot;
    const CrcPair incl = crc_block_inclusive(live ? me : kCrcIdentity, wave_tot, &tile_tot);
    if (tid == 0) tile_pre = lb_exclusive_crc(ctl.status[0], ctl.status[1], ctl.epoch, cur, tile_tot, stuck);
    __syncthreads();
    const CrcPair pre = tile_pre;
    const uint32_t chain = cur == 0 ? incl.c : op(pre, incl).c;  // (tile 0 has nothing in front of it: prev_crc is folded into record 0)
    chain_of[tid] = chain;
    __syncthreads();
    const uint32_t before = tid != 0 ? chain_of[tid - 1] : (cur == 0 ? prev_crc : pre.c);  // decoder.crc.Sum32() when the record arrives
    bool bad = false;
    if (live) {
      bad = (r.flags & kWalMalformed) != 0;
      if (!bad) {
        bad = r.kind == kWalCrc ? (before != 0 && r.crc != before) : r.crc != chain;
        if (bad) r.flags |= kWalBadCrc;
      }
    }
    // the first bad record: of this tile (block minimum), of the tiles before it (look-back)
    unsigned long long mine = bad ? i : kLbValueMask;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long y = __shfl_xor(mine, o, 64);
      mine = y < mine ? y : mine;
    }
    if (lane == 0) wave_min[wave] = mine;
    __syncthreads();
    if (tid == 0) {
      unsigned long long tmin = wave_min[0];
      for (int k = 1; k < kWaves; ++k) tmin = wave_min[k] < tmin ? wave_min[k] : tmin;
      const uint64_t earlier = lb_exclusive_min(ctl.status[2], ctl.epoch, cur, tmin, stuck);
      bad_before = earlier;
      const uint64_t live_recs = last - tile0;
      if (earlier == kLbValueMask && tmin != kLbValueMask) {  // the call's first bad record is mine: the chain stops in front of it
        const uint64_t k = tmin - tile0;
        pin[1] = k != 0 ? chain_of[k - 1] : (cur == 0 ? prev_crc : pre.c);
      }
      if (cur == n_tiles - 1) {
        const uint64_t fb = earlier < tmin ? earlier : tmin;
        pin[0] = fb < n ? fb : n;
        if (fb == kLbValueMask) pin[1] = chain_of[live_recs - 1];
        pin[3] = __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();  // (chain_of / wave_min are read; the frames in `stage` are dead)
    tile_records_out(r, live, reinterpret_cast<u32x4*>(&stage[0][0]), recs_h, tile0, n);
  }
}

// ---- the streaming form of the encoders ------------------------------------------------------------------------------
// block-wide exclusive sum of one u64 per thread -> this thread's prefix; *total = the block's sum
__device__ __forceinline__ uint64_t block_exclusive_u64(uint64_t v, uint64_t* wave_tot /*LDS [kWaves]*/, uint64_t* total) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = __shfl_up(incl, o, 64);
    if (lane >= (uint32_t)o) incl += y;
  }
  if (lane == 63) wave_tot[w] = incl;
  __syncthreads();
  uint64_t pre = 0, all = 0;
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)kWaves; ++k) {
    const uint64_t t = wave_tot[k];
    pre += k < w ? t : 0;
    all += t;
  }
  *total = all;
  return pre + incl - v;
}
// block-wide [min lo, max hi) over the threads that have a range (lo < hi); an empty result has lo >= hi
__device__ __forceinline__ void block_range(uint64_t& lo, uint64_t& hi, uint64_t* red /*LDS [2 * kWaves]*/) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  __syncthreads();  // (red may still be read from the previous use)
  if (lane == 0) {
    red[w] = lo;
    red[kWaves + w] = hi;
  }
  __syncthreads();
  lo = red[0];
  hi = red[kWaves];
#pragma unroll
  for (int k = 1; k < kWaves; ++k) {
    lo = red[k] < lo ? red[k] : lo;
    hi = red[kWaves + k] > hi ? red[kWaves + k] : hi;
  }
}

// every byte of one stream frame except the entry payloads (what wire_enc_write_kernel writes per lane)
__device__ inline void enc_write_frame(const WireMsg& m, const WireEnt* __restrict__ ents, uint8_t* dst, uint64_t frame_len) {
  Sink s(dst);
  s.u64_be(frame_len - 8);
  s.field(0x08, m.type);
  s.field(0x10, (uint64_t)m.to + 1);
  s.field(0x18, (uint64_t)m.from + 1);
  s.field(0x20, m.term);
  s.field(0x28, m.log_term);
  s.field(0x30, m.index);
  for (uint32_t k = 0; k < m.n_ents; ++k) {
    const WireEnt e = ents[m.ent_first + k];
    s.field(0x3a, entry_size(e.type, e.term, e.index, e.data_len));
    s.field(0x08, e.type);
    s.field(0x10, e.term);
    s.field(0x18, e.index);
    if (e.data_len) {
      s.field(0x22, e.data_len);
      s.skip(e.data_len);
    }
  }
  s.field(0x40, m.commit);
  s.u64_le(0x0010000a0612084aull);  // 4a 08 12 06 0a 00 10 00 | 18 00: the empty Snapshot
  s.byte(0x18);
  s.byte(0x00);
  s.field(0x50, m.reject ? 1 : 0);
  s.field(0x58, m.reject_hint);
  s.field(0x60, m.group);
  s.flush();
}

// A tile's run of output bytes, built in device memory by this workgroup, on its way to the caller's buffer: whole
// 16-byte quads of the DESTINATION as lane-consecutive stores (4 KB of consecutive host memory per instruction), the
// < 16 bytes at either end byte by byte (the neighbouring tiles write the other bytes of those quads).
__device__ inline void tile_bytes_out(const uint8_t* __restrict__ src, uint8_t* dst_h, uint64_t len) {
  if (len == 0) return;
  const uint32_t tid = threadIdx.x;
  uint64_t head = (16 - ((uintptr_t)dst_h & 15)) & 15;
  if (head > len) head = len;
  if (tid < head) dst_h[tid] = src[tid];
  const uint64_t quads = (len - head) >> 4;
  u32x4* d16 = reinterpret_cast<u32x4*>(dst_h + head);
  for (uint64_t q = tid; q < quads; q += kBlock) {
    u32x4 v;
    __builtin_memcpy(&v, src + head + (q << 4), 16);
    __builtin_nontemporal_store(v, d16 + q);
  }
  const uint64_t done = head + (quads << 4);
  if (tid < len - done) dst_h[done + tid] = src[done + tid];
}

// raftq_wire_encode on page-locked buffers, one launch: readers bring records (array 0), entry headers (1) and the payload
// pool (2) into the scratch; a worker sizes its tile's 256 messages, learns the tile's place in the stream from the
// look-back, writes the frames into the device copy of the stream (lane per frame, payloads by the wave) and pushes the
// tile's run of bytes and its frame offsets out.  Nothing is ever written at or behind out_h[cap].
// pin[0] = bytes the stream takes, pin[1] = messages refused (to / from >= 255, ranges outside ents[] / the pool: they
// count as empty frames), pin[3] = a wait gave up.
// raftq_propose_frames: msgs / ents go on in the scratch behind the caller's (in.seg[0] / [1].bytes: what the readers bring) with
// the records propose_apply_kernel wrote there -- n and n_ents count both parts; *ext_bad == ext_stamp (its validation refused the call):
// every message counts as refused, nothing is built.
static __global__ __launch_bounds__(kBlock) void wire_enc_fused_kernel(InFeed in, uint64_t n, uint64_t n_ents, uint64_t pool_bytes, uint8_t* d_out,
                                                                       uint8_t* out_h, uint64_t cap, uint64_t* off_h, TileCtl ctl,
                                                                       uint64_t* __restrict__ pin, const unsigned int* __restrict__ ext_bad, unsigned int ext_stamp) {
  if (blockIdx.x < in.readers) {
    reader_role(in, ctl.epoch);
    return;
  }
  __shared__ uint64_t wave_tot[kWaves];
  __shared__ uint64_t red[2 * kWaves];
  __shared__ uint64_t prefix[2];
  __shared__ uint32_t wave_bad[kWaves];
  __shared__ uint32_t tile_slot, feed_slot;
  const WireMsg* msgs = reinterpret_cast<const WireMsg*>(in.seg[0].dst);
  const WireEnt* ents = reinterpret_cast<const WireEnt*>(in.seg[1].dst);
  const uint8_t* pool = in.seg[2].dst;
  const uint32_t n_tiles = (uint32_t)((n + kBlock - 1) / kBlock);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned int* stuck = ctl.ticket + 1;
  for (;;) {
    const uint32_t cur = next_tile(ctl, &tile_slot);
    if (cur >= n_tiles) return;
    const uint64_t tile0 = (uint64_t)cur * kBlock, i = tile0 + tid;
    const uint64_t last = tile0 + kBlock < n ? tile0 + kBlock : n;
    const bool live = i < n;
    feed_wait(in, ctl.epoch, 0, tile0 * sizeof(WireMsg), last * sizeof(WireMsg), stuck, &feed_slot);
    WireMsg m = {};
    if (live) m = msgs[i];
    bool is_bad = live && (m.to >= 255 || m.from >= 255 || (m.n_ents != 0 && (uint64_t)m.ent_first + m.n_ents > n_ents));
    if (ext_bad != nullptr && *ext_bad == ext_stamp) is_bad = live;
    const bool walks = live && !is_bad && m.n_ents != 0;
    // the entry headers this tile names are in the scratch ...
    uint64_t lo = walks ? (uint64_t)m.ent_first * sizeof(WireEnt) : ~0ull, hi = walks ? ((uint64_t)m.ent_first + m.n_ents) * sizeof(WireEnt) : 0;
    block_range(lo, hi, red);
    feed_wait(in, ctl.epoch, 1, lo, hi, stuck, &feed_slot);
    uint64_t sz = 0;
    lo = ~0ull;
    hi = 0;
    if (live) {
      sz = 8 + msg_head_size(m) + msg_tail_size(m);
      if (walks) {
        for (uint32_t k = 0; k < m.n_ents; ++k) {
          const WireEnt e = ents[m.ent_first + k];
          if (e.data_len != 0) {
            if (e.data_off > pool_bytes || e.data_len > pool_bytes - e.data_off) {
              is_bad = true;
            } else {
              lo = e.data_off < lo ? e.data_off : lo;
              hi = e.data_off + e.data_len > hi ? e.data_off + e.data_len : hi;
            }
          }
          const uint64_t es = entry_size(e.type, e.term, e.index, e.data_len);
          sz += 1 + sov(es) + es;
        }
      }
      if (is_bad) sz = 0;
    }
    if (is_bad) {  // a refused message carries no payload range
      lo = ~0ull;
      hi = 0;
    }
    // ... and so are the payloads they name
    block_range(lo, hi, red);
    feed_wait(in, ctl.epoch, 2, lo, hi, stuck, &feed_slot);
    const uint64_t bb = __ballot(is_bad);
    if (lane == 0) wave_bad[wave] = (uint32_t)__popcll(bb);
    uint64_t tile_bytes;
    const uint64_t local = block_exclusive_u64(sz, wave_tot, &tile_bytes);  // (its barrier publishes wave_bad and the feed wait)
    if (tid == 0) {
      uint32_t tile_bad = 0;
      for (int k = 0; k < kWaves; ++k) tile_bad += wave_bad[k];
      const uint64_t pb = lb_exclusive(ctl.status[0], ctl.epoch, cur, tile_bytes, stuck);
      const uint64_t pr = lb_exclusive(ctl.status[1], ctl.epoch, cur, tile_bad, stuck);
      prefix[0] = pb;
      if (cur == n_tiles - 1) {
        pin[0] = pb + tile_bytes;
        pin[1] = pr + tile_bad;
        pin[3] = __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (off_h != nullptr) off_h[n] = pb + tile_bytes;
      }
    }
    __syncthreads();
    const uint64_t base = prefix[0], at = base + local;
    if (off_h != nullptr && live) off_h[i] = at;
    const bool fits = base + tile_bytes <= cap;  // (workgroup-uniform: a tile that does not fit is not built at all)
    if (fits) {
      if (live && sz != 0) enc_write_frame(m, ents, d_out + at, sz);
      // payloads: EIGHT messages of the wave at a time, eight lanes each (16 bytes per lane per step: 128 bytes of payload a step
      // and message).  Round 5 took a wave's entry-carrying messages one after the other with all 64 lanes -- fine at 15 % MsgApps,
      // but a turn's proposals are ALL MsgApps with a ~80-byte entry: 64 dependent load-store round trips per wave, ~100 us a
      // tile, most of what raftq_propose_frames' marshal took (round 6).
      const uint64_t pos0 = at + 8 + msg_head_size(m);
      const uint64_t walk_bits = __ballot(walks && !is_bad);
      const uint32_t n_walk = (uint32_t)__popcll(walk_bits), sub = lane & 7u, grp = lane >> 3;
      for (uint32_t it = 0; it * 8 < n_walk; ++it) {
        const uint32_t r = it * 8 + grp;  // this sub-group's message: the lane of the r-th set bit of walk_bits
        uint32_t at_bit = 0, rr = r;
        uint64_t w = walk_bits;
#pragma unroll
        for (int sft = 32; sft != 0; sft >>= 1) {
          const uint64_t low = w & ((1ull << sft) - 1);
          const uint32_t c = (uint32_t)__popcll(low);
          if (rr >= c) {
            rr -= c;
            w >>= sft;
            at_bit += (uint32_t)sft;
          } else {
            w = low;
          }
        }
        const bool mine = r < n_walk;
        const int src_lane = mine ? (int)at_bit : (int)lane;
        uint64_t pos = __shfl(pos0, src_lane, 64);
        // (every lane takes part in every shuffle: a shuffle under a branch only sees the lanes that took it)
        const uint32_t first = __shfl(m.ent_first, src_lane, 64), cnt_src = __shfl(m.n_ents, src_lane, 64);
        const uint32_t cnt = mine ? cnt_src : 0u;
        for (uint32_t k = 0; k < cnt; ++k) {
          const WireEnt e = ents[first + k];
          const uint64_t es = entry_size(e.type, e.term, e.index, e.data_len);
          pos += 1 + sov(es) + 3 + sov(e.type) + sov(e.term) + sov(e.index);
          if (e.data_len) {
            pos += 1 + sov(e.data_len);
            sub8_copy(d_out + pos, pool + e.data_off, e.data_len, sub);
            pos += e.data_len;
          }
        }
      }
      __syncthreads();  // the tile's bytes are in the device copy (this workgroup wrote all of them) ...
      tile_bytes_out(d_out + base, out_h + base, tile_bytes);  // ... and leave as one run
    }
  }
}

// one WAL frame except an entry's / the metadata's payload bytes (what wal_enc_write_kernel writes per lane)
__device__ inline void wal_write_frame(const WalRec& r, uint32_t crc, uint64_t dsz, uint8_t* dst) {
  Sink s(dst);
  s.u64_le(wal_rec_size(r, crc, dsz));
  s.field(0x08, r.kind);
  s.field(0x10, crc);
  const bool has_data = r.kind != kWalCrc && !(r.kind == kWalMetadata && dsz == 0);
  if (has_data) {
    s.field(0x1a, dsz);
    if (r.kind == kWalEntry) {
      s.field(0x08, r.entry_type);
      s.field(0x10, r.term);
      s.field(0x18, r.index);
      if (r.data_len) {
        s.field(0x22, r.data_len);
        s.skip(r.data_len);
      }
      s.field(0x28, r.group);
    } else if (r.kind == kWalState) {
      s.field(0x08, r.term);
      s.field(0x10, r.vote);
      s.field(0x18, r.index);
      s.field(0x20, r.group);
    } else if (r.kind == kWalSnapshot) {
      s.field(0x08, r.index);
      s.field(0x10, r.term);
    }  // metadata: payload only
  }
  s.flush();
}

// raftq_wal_encode on page-locked buffers, one launch: readers bring the records (array 0) and the payload pool (1) into
// the scratch; a worker CRCs its tile's records (short payloads in the lane, long ones by the wave), chains the CRCs
// through the look-back -- a frame's size depends on its chained crc's varint length --, sizes, learns the tile's place in
// the segment from a second look-back, writes the frames into the device copy and pushes the tile's run of bytes and its
// frame offsets out.  Nothing is ever written at or behind out_h[cap].
// pin[0] = bytes, pin[1] = records refused (unknown kind, payload outside the pool), pin[2] = the chain's end,
// pin[3] = a wait gave up.
static __global__ __launch_bounds__(kBlock) void wal_enc_fused_kernel(InFeed in, uint64_t n, uint64_t pool_bytes, uint32_t prev_crc, uint8_t* d_out,
                                                                      uint8_t* out_h, uint64_t cap, uint64_t* off_h, TileCtl ctl,
                                                                      uint64_t* __restrict__ pin) {
  if (blockIdx.x < in.readers) {
    reader_role(in, ctl.epoch);
    return;
  }
  __shared__ uint32_t tab[kCrcTabs * 256];
  __shared__ CrcPair crc_tot[kWaves];
  __shared__ CrcPair tile_pre;
  __shared__ uint64_t wave_tot[kWaves];
  __shared__ uint64_t red[2 * kWaves];
  __shared__ uint64_t prefix[2];
  __shared__ uint32_t wave_bad[kWaves];
  __shared__ uint32_t tile_slot, feed_slot;
  const WalRec* recs = reinterpret_cast<const WalRec*>(in.seg[0].dst);
  const uint8_t* pool = in.seg[1].dst;
  const uint32_t n_tiles = (uint32_t)((n + kBlock - 1) / kBlock);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned int* stuck = ctl.ticket + 1;
  CrcCompose op;
  crc_table_init(tab);
  for (;;) {
    const uint32_t cur = next_tile(ctl, &tile_slot);
    if (cur >= n_tiles) return;
    const uint64_t tile0 = (uint64_t)cur * kBlock, i = tile0 + tid;
    const uint64_t last = tile0 + kBlock < n ? tile0 + kBlock : n;
    const bool live = i < n;
    feed_wait(in, ctl.epoch, 0, tile0 * sizeof(WalRec), last * sizeof(WalRec), stuck, &feed_slot);
    WalRec r = {};
    if (live) r = recs[i];
    bool is_bad = live && (r.kind < 1 || r.kind > 5);
    const bool payload = live && !is_bad && wal_has_payload(r.kind) && r.data_len != 0;
    if (payload && (r.data_off > pool_bytes || r.data_len > pool_bytes - r.data_off)) is_bad = true;
    const bool copies = payload && !is_bad;
    uint64_t lo = copies ? r.data_off : ~0ull, hi = copies ? r.data_off + r.data_len : 0;
    block_range(lo, hi, red);
    feed_wait(in, ctl.epoch, 1, lo, hi, stuck, &feed_slot);
    // the record's map on the running CRC (wal_enc_crc_kernel): front fields, payload, the group field behind an entry
    CrcPair me = kCrcIdentity;
    uint32_t raw = 0xffffffffu;
    const bool long_pl = copies && r.data_len > kCoopBytes;
    if (live && !is_bad) {
      if (r.kind == kWalEntry) {
        raw = crc_field(tab, raw, 0x08, r.entry_type);
        raw = crc_field(tab, raw, 0x10, r.term);
        raw = crc_field(tab, raw, 0x18, r.index);
        if (r.data_len) raw = crc_field(tab, raw, 0x22, r.data_len);
      } else if (r.kind == kWalState) {
        raw = crc_field(tab, raw, 0x08, r.term);
        raw = crc_field(tab, raw, 0x10, r.vote);
        raw = crc_field(tab, raw, 0x18, r.index);
        raw = crc_field(tab, raw, 0x20, r.group);
      } else if (r.kind == kWalSnapshot) {
        raw = crc_field(tab, raw, 0x08, r.index);
        raw = crc_field(tab, raw, 0x10, r.term);
      }
      if (copies && !long_pl) raw = crc_span(tab, raw, pool + r.data_off, r.data_len);
    }
    for (uint64_t todo = __ballot(long_pl); todo != 0; todo &= todo - 1) {  // long payloads: the whole wave per payload
      const int l = __ffsll((long long)todo) - 1;
      const uint64_t po = wave_bcast_u64(r.data_off, l);
      const uint32_t pl = __builtin_amdgcn_readlane(r.data_len, l);
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(wave_crc(tab, pool + po, pl));  // (valid in lane 0)
      if ((int)lane == l) raw = ~(crc_mulmod(crc_xpow8(pl), ~raw) ^ c0);  // crc(front || payload)
    }
    uint64_t dsz = 0;
    if (live && !is_bad) {
      if (r.kind == kWalEntry) raw = crc_field(tab, raw, 0x28, r.group);
      dsz = wal_data_size(r);
      me.c = dsz ? ~raw : 0u;
      me.p = crc_xpow8(dsz);
      if (i == 0) me.c ^= crc_mulmod(me.p, prev_crc);
    }
    CrcPair tile_crc;
    const CrcPair incl = crc_block_inclusive(live ? me : kCrcIdentity, crc_tot, &tile_crc);
    if (tid == 0) tile_pre = lb_exclusive_crc(ctl.status[0], ctl.status[1], ctl.epoch, cur, tile_crc, stuck);
    __syncthreads();
    const uint32_t crc = op(tile_pre, incl).c;  // Record.crc of record i: the running CRC through it
    const uint64_t sz = live && !is_bad ? 8 + wal_rec_size(r, crc, dsz) : 0;  // (a refused record takes no bytes: its fields are not to be trusted)
    const uint64_t bb = __ballot(is_bad);
    if (lane == 0) wave_bad[wave] = (uint32_t)__popcll(bb);
    uint64_t tile_bytes;
    const uint64_t local = block_exclusive_u64(sz, wave_tot, &tile_bytes);
    if (tid == 0) {
      uint32_t tile_bad = 0;
      for (int k = 0; k < kWaves; ++k) tile_bad += wave_bad[k];
      const uint64_t pb = lb_exclusive(ctl.status[2], ctl.epoch, cur, tile_bytes, stuck);
      const uint64_t pr = lb_exclusive(ctl.status[3], ctl.epoch, cur, tile_bad, stuck);
      prefix[0] = pb;
      if (cur == n_tiles - 1) {
        pin[0] = pb + tile_bytes;
        pin[1] = pr + tile_bad;
        pin[3] = __hip_atomic_load(stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (off_h != nullptr) off_h[n] = pb + tile_bytes;
      }
    }
    if (i == n - 1) pin[2] = crc;  // the chain's end: prev_crc of the next batch
    __syncthreads();
    const uint64_t base = prefix[0], at = base + local;
    if (off_h != nullptr && live) off_h[i] = at;
    const bool fits = base + tile_bytes <= cap;  // (workgroup-uniform)
    if (fits) {
      if (live && !is_bad) wal_write_frame(r, crc, dsz, d_out + at);
      uint64_t pos0 = at + 8 + 2 + sov(r.kind) + sov(crc) + 1 + sov(dsz);
      if (r.kind == kWalEntry) pos0 += 3 + sov(r.entry_type) + sov(r.term) + sov(r.index) + 1 + sov(r.data_len);
      // payloads, pool -> frame: eight records of the wave at a time, eight lanes each (see wire_enc_fused_kernel: a turn's WAL is
      // nearly all entry records)
      const uint64_t copy_bits = __ballot(copies);
      const uint32_t n_copy = (uint32_t)__popcll(copy_bits), sub = lane & 7u, grp = lane >> 3;
      for (uint32_t it = 0; it * 8 < n_copy; ++it) {
        const uint32_t rk = it * 8 + grp;
        uint32_t at_bit = 0, rr = rk;
        uint64_t w = copy_bits;
#pragma unroll
        for (int sft = 32; sft != 0; sft >>= 1) {
          const uint64_t low = w & ((1ull << sft) - 1);
          const uint32_t c = (uint32_t)__popcll(low);
          if (rr >= c) {
            rr -= c;
            w >>= sft;
            at_bit += (uint32_t)sft;
          } else {
            w = low;
          }
        }
        const bool mine = rk < n_copy;
        const int src_lane = mine ? (int)at_bit : (int)lane;
        const uint64_t dpos = __shfl(pos0, src_lane, 64), spos = __shfl(r.data_off, src_lane, 64);
        const uint32_t len = __shfl(r.data_len, src_lane, 64);
        if (mine) sub8_copy(d_out + dpos, pool + spos, len, sub);
      }
      __syncthreads();
      tile_bytes_out(d_out + base, out_h + base, tile_bytes);
    }
  }
}

}  // namespace raftqk
