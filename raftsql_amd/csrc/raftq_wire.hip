// raftq_wire.hip -- implementation of include/raftq_wire.h: batched raftpb.Message stream frames
// and walpb.Record WAL frames on the GPU (raftq_wire_kernels.hpp).  Host side: move the caller's
// buffers to the device, run the launch chain on the handle's stream, move the results back.
// No CPU path: without the handle's GPU nothing here encodes or decodes a byte
// (raftq_wire_scan_frames, the serial length-word walk, is the one host-only entry point).
#include "raftq_wire.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raftq_internal.hpp"
#include "raftq_propose_kernels.hpp"
#include "raftq_wire_kernels.hpp"

using namespace raftqk;
using raftq_detail::fail;
using raftq_detail::use_device;

static_assert(sizeof(raftq_wire_msg_t) == sizeof(WireMsg) && sizeof(raftq_wire_ent_t) == sizeof(WireEnt) &&
                  sizeof(raftq_wal_rec_t) == sizeof(WalRec) && sizeof(raftq_prop_t) == sizeof(PropRec) && sizeof(raftq_prop_ent_t) == sizeof(PropEnt),
              "ABI struct mismatch");

namespace {

constexpr uint64_t kMaxItems = 0x7ffffffeull;  // batch positions travel as 31-bit values

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) {
    const size_t o = off;
    off += align256(bytes + 16);  // 16 bytes of slack behind every byte buffer
    return o;
  }
};

int grow(raftq_t* h, void** buf, size_t* have, size_t want) {
  if (want <= *have) return RAFTQ_OK;
  if (*buf) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  const size_t bytes = std::max(want + want / 2, (size_t)1 << 20);
  HIPCHK(h, hipMalloc(buf, bytes));
  *have = bytes;
  return RAFTQ_OK;
}

int ensure_pin(raftq_t* h) {
  if (h->wire_pin) return RAFTQ_OK;
  HIPCHK(h, hipHostMalloc((void**)&h->wire_pin, 256, hipHostMallocMapped));
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->wire_pin_d, h->wire_pin, 0));
  HIPCHK(h, hipMalloc((void**)&h->wire_flags, 64));
  HIPCHK(h, hipMemsetAsync(h->wire_flags, 0, 64, h->stream));  // wire_tail_kernel leaves a used word zero again
  return RAFTQ_OK;
}

// the address the device has for caller memory, or nullptr when it has none (pageable memory: the runtime's copies then)
void* dev_view(const void* p) {
  if (!p) return nullptr;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return d;
}

// totals / flags of the call -> wire_pin[0], wire_pin[1] (read after the next hipStreamSynchronize)
int tail_to_pin(raftq_t* h, const uint64_t* total, unsigned long long* flag) {
  hipLaunchKernelGGL(wire_tail_kernel, dim3(1), dim3(64), 0, h->stream, total, flag, h->wire_pin_d);
  HIPCHK(h, hipGetLastError());
  return RAFTQ_OK;
}

unsigned blocks_for(uint64_t lanes) { return (unsigned)((lanes + kBlock - 1) / kBlock); }

// ---- the streaming form (one persistent kernel per call; raftq_wire_kernels.hpp) -----------------------------------
bool streaming_on() {  // RAFTQ_WIRE_STREAMING=0: the copying form even for page-locked buffers (for A/B and the tests)
  const char* e = std::getenv("RAFTQ_WIRE_STREAMING");  // (read per call: the tests switch forms inside one process)
  return !(e && e[0] == '0');
}
// Worker workgroups of a streaming kernel (RAFTQ_WIRE_WGS overrides): a tile is ~25 us of dependent work (flags, scratch
// reads, the parse at one wave per SIMD, the look-back), the link moves a tile every ~0.3 us.
// `fit`: what is resident at once beside the readers when the caller knows better than 208 (the decoder: its LDS per workgroup)
unsigned fused_grid(uint32_t n_tiles, unsigned fit = 208u) {
  const char* e = std::getenv("RAFTQ_WIRE_WGS");  // read per call: the tests drive tiny grids through one process
  const long v = e ? std::strtol(e, nullptr, 10) : 0;
  return std::min<unsigned>(v > 0 && v <= 4096 ? (unsigned)v : fit, n_tiles);
}
// frames per tile = threads per workgroup of the streaming decoder (RAFTQ_WIRE_TILE=128|256; raftq_wire_kernels.hpp wire_dec_fused_kernel)
unsigned dec_tile() {
  const char* e = std::getenv("RAFTQ_WIRE_TILE");  // read per call: the tests run both
  const long v = e ? std::strtol(e, nullptr, 10) : 0;
  return v == 128 ? 128u : 256u;  // measured (profiles/r06/wire_tile_ab.jsonl): 256 frames 172 us a call, 128 frames 186
}

constexpr uint64_t kLbHead = 4;  // words in front of the status arrays
// reader workgroups of a streaming kernel (RAFTQ_WIRE_READERS overrides): 48 pull a caller's array at 55 GB/s, more are slower
// RAFTQ_WIRE_READERS=0: NO reader workgroups -- every chunk is brought in by a worker that found nobody else doing it (the
// liveness argument's limit case, tests/test_wire_gpu.py::test_streaming_codecs_without_readers).  `dflt`: 48 workgroups of 256
// threads; the 128-thread decoder launches 96 for the same bytes in flight.
unsigned fused_readers(uint32_t chunks, unsigned dflt = 48u) {
  const char* e = std::getenv("RAFTQ_WIRE_READERS");
  char* end = nullptr;
  const long v = e ? std::strtol(e, &end, 10) : -1;
  if (e && end != e && v == 0) return 0;
  return std::min<unsigned>(v > 0 && v <= 1024 ? (unsigned)v : dflt, chunks);
}
// bytes of all arrays together that a reader brings in before it raises a flag (RAFTQ_WIRE_CHUNK overrides)
uint64_t sdma_chunk();
uint64_t feed_chunk() {
  if (const uint64_t s = sdma_chunk()) return s;
  const char* e = std::getenv("RAFTQ_WIRE_CHUNK");
  const long v = e ? std::strtol(e, nullptr, 10) : 0;
  return v >= 1024 && v <= (1 << 20) ? (uint64_t)v : 8192;
}

// The readers' plan for up to three caller arrays (device views `src`, all 16-byte aligned): where they go in the scratch
// (carved behind `c`), how many chunks, how many bytes of every array per chunk.  max_chunks: flags available.
struct FeedPlan {
  InFeed in;
  size_t off[3];
};
// extra[k]: bytes of array k that follow, in the scratch, what the readers bring (records a kernel writes there itself)
FeedPlan plan_feed(Carver& c, const void* const src[3], const uint64_t bytes[3], uint64_t max_chunks, unsigned readers_dflt = 48u,
                   const uint64_t* extra = nullptr) {
  FeedPlan p{};
  uint64_t total = 0;
  for (int k = 0; k < 3; ++k) total += bytes[k];
  const uint64_t chunks = std::max<uint64_t>(1, std::min<uint64_t>(max_chunks, (total + feed_chunk() - 1) / feed_chunk()));
  for (int k = 0; k < 3; ++k) {
    p.off[k] = c.take(bytes[k] + (extra ? extra[k] : 0));
    p.in.seg[k].src = (const uint8_t*)src[k];
    p.in.seg[k].bytes = bytes[k];
    p.in.seg[k].per_chunk = std::max<uint64_t>(256, ((bytes[k] + chunks - 1) / chunks + 255) / 256 * 256);
  }
  p.in.chunks = (uint32_t)chunks;
  p.in.readers = fused_readers(p.in.chunks, readers_dflt);
  return p;
}
// (called once per launch, after tile_ctl: the chunk tickets of the launch are accounted for in tile_ctl_launched)
// The chunk ticket is monotonic across calls: a launch with reader workgroups draws exactly chunks + readers tickets whoever
// copies what (a worker that serves itself claims by compare-and-swap and never draws past the end; the readers draw the rest
// and one beyond each), so the next call's base is known without asking the device.  A launch with NO readers (RAFTQ_WIRE_READERS=0,
// RAFTQ_WIRE_SDMA) only claims the chunks somebody waited for -- a chunk past every array's end, or one that holds bytes no tile
// names, stays unclaimed -- so its count is not known: the ticket word is zeroed in front of such a launch and in front of the
// first launch after one (a 4-byte memset in the stream: test and A/B shapes only).
int bind_feed(raftq_t* h, FeedPlan& p, uint8_t* base, unsigned long long* flags) {
  for (int k = 0; k < 3; ++k) p.in.seg[k].dst = base + p.off[k];
  p.in.flag = flags;
  p.in.chunk_ticket = reinterpret_cast<unsigned int*>(h->wire_lb + 3);  // the head's fourth word
  if (p.in.readers == 0 || h->wire_chunk_unknown) {
    HIPCHK(h, hipMemsetAsync(p.in.chunk_ticket, 0, 4, h->stream));
    h->wire_chunk_base = 0;
    h->wire_chunk_unknown = p.in.readers == 0;
  }
  p.in.chunk_base = h->wire_chunk_base;
  p.in.no_serve = 0;
  h->wire_chunk_pending = p.in.readers ? p.in.chunks + p.in.readers : 0;  // every reader workgroup draws exactly one ticket beyond the chunks
  return RAFTQ_OK;
}
// RAFTQ_WIRE_SDMA=<chunk KiB> (A/B only; VERDICT r04 / r05 item 1: "build the SDMA-reader A/B instead of citing the old probe"): the
// decoder's input is brought in by the RUNTIME's copies on a second stream -- one hipMemcpyAsync per array and chunk, the chunk's
// flag raised behind it by hipStreamWriteValue64 -- and the kernel is launched with no reader workgroups and workers that only
// wait.  profiles/r06/wire_tile_ab.jsonl has what it measured (2.0 ms with 64 KB chunks, 0.35 ms with 1 MB chunks, against 0.17 ms).
uint64_t sdma_chunk() {
  const char* e = std::getenv("RAFTQ_WIRE_SDMA");
  const long v = e ? std::strtol(e, nullptr, 10) : 0;
  return v >= 1 && v <= 65536 ? (uint64_t)v << 10 : 0;
}
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ticket word + status arrays for a call of n_tiles tiles; a new call is a new epoch (the words of older calls read as
// "not published yet"), the arrays are zeroed when they are (re)allocated and when the 16-bit epoch wraps
int tile_ctl(raftq_t* h, uint64_t n_tiles, TileCtl* ctl) {
  if (n_tiles > h->wire_lb_tiles) {
    if (h->wire_lb) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      HIPCHK(h, hipFree(h->wire_lb));
      h->wire_lb = nullptr;
      h->wire_lb_tiles = 0;
    }
    const uint64_t tiles = std::max<uint64_t>(n_tiles + n_tiles / 2, 4096);
    HIPCHK(h, hipMalloc((void**)&h->wire_lb, (kLbHead + kLbArrays * tiles) * 8));
    HIPCHK(h, hipMemsetAsync(h->wire_lb, 0, (kLbHead + kLbArrays * tiles) * 8, h->stream));
    h->wire_lb_tiles = tiles;
    h->wire_ticket_base = 0;
    h->wire_chunk_base = 0;
    h->wire_chunk_unknown = false;
    h->wire_epoch = 0;
  }
  if (++h->wire_epoch > 0xffffu) {
    HIPCHK(h, hipMemsetAsync(h->wire_lb + kLbHead, 0, kLbArrays * h->wire_lb_tiles * 8, h->stream));
    h->wire_epoch = 1;
  }
  ctl->ticket = reinterpret_cast<unsigned int*>(h->wire_lb);
  ctl->ticket_base = h->wire_ticket_base;
  ctl->ablate = 0;
#if defined(RAFTQ_WIRE_TRACE)
  if (const char* e = std::getenv("RAFTQ_WIRE_ABLATE")) ctl->ablate = (uint32_t)std::strtol(e, nullptr, 10);
#endif
  ctl->epoch = h->wire_epoch;
  for (int k = 0; k < kLbArrays; ++k) ctl->status[k] = h->wire_lb + kLbHead + (uint64_t)k * h->wire_lb_tiles;
  return RAFTQ_OK;
}
// every worker of a launch draws exactly one ticket beyond the tiles
void tile_ctl_launched(raftq_t* h, uint32_t n_tiles, unsigned workers) {
  h->wire_ticket_base += n_tiles + workers;
  h->wire_chunk_base += h->wire_chunk_pending;
  h->wire_chunk_pending = 0;
}
#if defined(RAFTQ_WIRE_TRACE)
// RAFTQ_TRACE_STAMP's rows of the call just waited for -> stderr (once every 16th call): per tile, microseconds since the
// earliest stamp of the launch
void trace_dump(raftq_t* h, const char* what, uint32_t n_tiles) {
  static int calls = 0;
  static const bool dump = std::getenv("RAFTQ_WIRE_TRACE_DUMP") != nullptr;
  if (!dump || (calls++ & 15) != 15 || n_tiles * 8ull > h->wire_lb_tiles) return;
  std::vector<unsigned long long> t(n_tiles * 8ull);
  if (hipMemcpy(t.data(), h->wire_lb + kLbHead + kLbSpare * h->wire_lb_tiles, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  unsigned long long t0 = ~0ull;
  for (uint32_t i = 0; i < n_tiles; ++i) t0 = std::min(t0, t[i * 8ull]);
  std::fprintf(stderr, "[trace %s] tile: claimed offs_in dma_issued frames_in parsed lookback ents_out recs_out (us)\n", what);
  for (uint32_t i = 0; i < n_tiles; i += (n_tiles > 64 ? n_tiles / 32 : 1)) {
    std::fprintf(stderr, "[trace %s] %5u:", what, i);
    for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %7.1f", (double)(t[i * 8ull + k] - t0) / 100.0);
    std::fprintf(stderr, "\n");
  }
}
#endif
// after the call's wait: did a look-back give up (wire_pin[3], copied from the control block by the last tile)?
int tile_ctl_check(raftq_t* h, const char* who, uint32_t pin_base = 0) {
  if (h->wire_pin[pin_base + 3] == 0) return RAFTQ_OK;
  h->wire_lb_tiles = 0;  // the control block is not trusted any more: the next call allocates a fresh one
  return fail(h, RAFTQ_EHIP, std::string(who) + ": a workgroup waited a second for its predecessor's tile and gave up; the results are not valid");
}

// chain[i] = pair[0] . pair[1] . ... . pair[i]; tot: scratch for ceil(n / kBlock) pairs
int crc_chain_scan(raftq_t* h, const CrcPair* pair, CrcPair* chain, uint64_t n, CrcPair* tot) {
  const unsigned nb = blocks_for(n);
  hipLaunchKernelGGL(crc_scan_blocks_kernel, dim3(nb), dim3(kBlock), 0, h->stream, pair, chain, n, tot);
  if (nb > 1)
    hipLaunchKernelGGL(crc_scan_apply_kernel, dim3(nb), dim3(kBlock), 0, h->stream, chain, n, (const CrcPair*)tot);
  HIPCHK(h, hipGetLastError());
  return RAFTQ_OK;
}

int h2d(raftq_t* h, void* dst, const void* src, size_t bytes) {
  if (bytes) HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  return RAFTQ_OK;
}
int d2h(raftq_t* h, void* dst, const void* src, size_t bytes) {
  if (bytes) HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  return RAFTQ_OK;
}

}  // namespace

void raftq_detail::free_wire_state(raftq_t* h) {
  (void)hipFree(h->wire_dev);
  (void)hipFree(h->wire_out);
  (void)hipFree(h->wire_flags);
  (void)hipFree(h->wire_lb);
  if (h->wire_copy_stream) (void)hipStreamDestroy(h->wire_copy_stream);
  if (h->wire_copy_ev) (void)hipEventDestroy(h->wire_copy_ev);
  h->wire_copy_stream = nullptr;
  h->wire_copy_ev = nullptr;
  h->wire_flags = nullptr;
  h->wire_lb = nullptr;
  h->wire_lb_tiles = 0;
  if (h->wire_pin) (void)hipHostFree(h->wire_pin);
  h->wire_dev = h->wire_out = nullptr;
  h->wire_pin = nullptr;
}

extern "C" {

int raftq_wire_scan_frames(const void* buf, uint64_t nbytes, int big_endian, uint64_t* off, uint64_t cap,
                           uint64_t* n_frames, uint64_t* consumed) {
  if ((!buf && nbytes) || !off || !n_frames || !consumed) return fail(nullptr, RAFTQ_EINVAL, "raftq_wire_scan_frames: null argument");
  const uint8_t* p = (const uint8_t*)buf;
  uint64_t at = 0, k = 0;
  for (; k < cap && nbytes - at >= 8; ++k) {
    uint64_t len;
    std::memcpy(&len, p + at, 8);  // this library only runs on little-endian hosts
    if (big_endian) len = __builtin_bswap64(len);
    if (len > nbytes - at - 8) break;  // the tail is torn: leave it to the caller
    off[k] = at;
    at += 8 + len;
  }
  off[k] = at;
  *n_frames = k;
  *consumed = at;
  return RAFTQ_OK;
}

int raftq_wire_encode(raftq_t* h, const raftq_wire_msg_t* msgs, uint64_t n, const raftq_wire_ent_t* ents,
                      uint64_t n_ents, const void* pool, uint64_t pool_bytes, void* out, uint64_t cap,
                      uint64_t* frame_off, raftq_wire_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (n == 0) {
    if (frame_off) frame_off[0] = 0;
    return RAFTQ_OK;
  }
  if (!msgs || (n_ents && !ents) || (pool_bytes && !pool) || (cap && !out))
    return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: null argument");
  if (n > kMaxItems || n_ents > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: batch too large");
  if (int rc = ensure_pin(h)) return rc;
  // page-locked caller buffers (what a node passes every turn) take the streaming form; anything else the copying form
  void *v_msgs = nullptr, *v_ents = nullptr, *v_pool = nullptr, *v_out = nullptr, *v_off = nullptr;
  const bool mapped = streaming_on() && cap != 0 && cap <= ((uint64_t)1 << 31) && (v_msgs = dev_view(msgs)) != nullptr &&
                      (n_ents == 0 || (v_ents = dev_view(ents)) != nullptr) && (pool_bytes == 0 || (v_pool = dev_view(pool)) != nullptr) &&
                      (v_out = dev_view(out)) != nullptr && (!frame_off || (v_off = dev_view(frame_off)) != nullptr);
  if (mapped && aligned16(v_msgs) && aligned16(v_ents) && aligned16(v_pool) && aligned16(v_out) && aligned16(v_off)) {
    // page-locked caller buffers: the streaming form (readers | workers in one launch; raftq_wire_kernels.hpp)
    const uint32_t n_tiles = blocks_for(n);
    const unsigned workers = fused_grid(n_tiles);
    Carver fc;
    const void* const src[3] = {v_msgs, v_ents, v_pool};
    const uint64_t sizes[3] = {n * sizeof(WireMsg), n_ents * sizeof(WireEnt), pool_bytes};
    TileCtl ctl;
    if (int rc = tile_ctl(h, std::max<uint64_t>(n_tiles, (sizes[0] + sizes[1] + sizes[2]) / feed_chunk() + 1), &ctl)) return rc;
    FeedPlan plan = plan_feed(fc, src, sizes, h->wire_lb_tiles);
    if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, fc.off)) return rc;
    if (int rc = grow(h, &h->wire_out, &h->wire_out_bytes, cap + 16)) return rc;
    if (int rc = bind_feed(h, plan, (uint8_t*)h->wire_dev, ctl.status[kLbFlags])) return rc;
    hipLaunchKernelGGL(wire_enc_fused_kernel, dim3(plan.in.readers + workers), dim3(kBlock), 0, h->stream, plan.in, n, n_ents, pool_bytes,
                       (uint8_t*)h->wire_out, (uint8_t*)v_out, cap, (uint64_t*)v_off, ctl, h->wire_pin_d, (const unsigned int*)nullptr, 0u);
    HIPCHK(h, hipGetLastError());
    tile_ctl_launched(h, n_tiles, workers);
    HIPCHK(h, raftq_detail::wait_call(h));
    if (int rc = tile_ctl_check(h, "raftq_wire_encode")) return rc;
    const uint64_t total = h->wire_pin[0];
    if (h->wire_pin[1])
      return fail(h, RAFTQ_EINVAL,
                  "raftq_wire_encode: a message has to / from >= 255, an entry range outside ents[], or a payload outside "
                  "the pool; the output is not valid");
    if (counts) {
      counts->n_msgs = n;
      counts->n_ents = n_ents;
      counts->bytes = total;
    }
    if (total > cap) return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: out is too small (counts->bytes is the size needed)");
    return RAFTQ_OK;
  }
  // the copying form: the runtime's copies, a wait in the middle to learn the size
  const size_t scan_bytes = scan_sum_scratch_bytes(n + 1);  // tile totals of the hand-written scan
  Carver c;
  const size_t o_msgs = c.take(n * sizeof(WireMsg)), o_ents = c.take(n_ents * sizeof(WireEnt)),
               o_pool = c.take(pool_bytes), o_sizes = c.take((n + 1) * 8), o_off = c.take((n + 1) * 8),
               o_bad = c.take(8), o_scan = c.take(scan_bytes);
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, c.off)) return rc;
  uint8_t* base = (uint8_t*)h->wire_dev;
  WireMsg* d_msgs = (WireMsg*)(base + o_msgs);
  WireEnt* d_ents = (WireEnt*)(base + o_ents);
  uint8_t* d_pool = base + o_pool;
  uint64_t *d_sizes = (uint64_t*)(base + o_sizes), *d_off = (uint64_t*)(base + o_off);
  unsigned int* d_bad = (unsigned int*)(h->wire_flags + 0);  // (o_bad: unused since the flags have a block of their own)
  (void)o_bad;
  if (int rc = h2d(h, d_msgs, msgs, n * sizeof(WireMsg))) return rc;
  if (int rc = h2d(h, d_ents, ents, n_ents * sizeof(WireEnt))) return rc;
  if (int rc = h2d(h, d_pool, pool, pool_bytes)) return rc;
  hipLaunchKernelGGL(wire_enc_size_kernel, dim3(blocks_for(n + 1)), dim3(kBlock), 0, h->stream, (const WireMsg*)d_msgs, n,
                     (const WireEnt*)d_ents, n_ents, pool_bytes, d_sizes, d_bad);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, exclusive_sum_u64((const uint64_t*)d_sizes, d_off, n + 1, (uint64_t*)(base + o_scan), h->stream));
  if (int rc = tail_to_pin(h, d_off + n, h->wire_flags + 0)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint64_t total = h->wire_pin[0];
  if ((uint32_t)h->wire_pin[1])
    return fail(h, RAFTQ_EINVAL,
                "raftq_wire_encode: a message has to / from >= 255, an entry range outside ents[], or a payload outside "
                "the pool; nothing was written");
  if (counts) {
    counts->n_msgs = n;
    counts->n_ents = n_ents;
    counts->bytes = total;
  }
  if (total > cap) return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: out is too small (counts->bytes is the size needed)");
  if (int rc = grow(h, &h->wire_out, &h->wire_out_bytes, total + 16)) return rc;
  uint8_t* d_out = (uint8_t*)h->wire_out;
  hipLaunchKernelGGL(wire_enc_write_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, (const WireMsg*)d_msgs, n,
                     (const WireEnt*)d_ents, (const uint64_t*)d_off, d_out);
  if (n_ents)
    hipLaunchKernelGGL(wire_enc_payload_kernel, dim3(blocks_for(n * 64)), dim3(kBlock), 0, h->stream,
                       (const WireMsg*)d_msgs, n, (const WireEnt*)d_ents, (const uint64_t*)d_off,
                       (const uint8_t*)d_pool, d_out);
  HIPCHK(h, hipGetLastError());
  if (int rc = d2h(h, out, d_out, total)) return rc;
  if (frame_off)
    if (int rc = d2h(h, frame_off, d_off, (n + 1) * 8)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

int raftq_propose_frames(raftq_t* h, const raftq_prop_t* props, uint64_t n_props, const raftq_prop_ent_t* prop_ents, uint64_t n_prop_ents,
                         const raftq_wire_msg_t* msgs, uint64_t n_msgs, const raftq_wire_ent_t* ents, uint64_t n_ents, const void* pool,
                         uint64_t pool_bytes, void* out, uint64_t cap, uint64_t* frame_off, raftq_wire_counts_t* counts) {
  if (int rc = raftq_detail::use_device_idle(h, "raftq_propose_frames")) return rc;
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (n_props == 0)  // nothing proposed: the marshal of what the caller queued
    return raftq_wire_encode(h, msgs, n_msgs, ents, n_ents, pool, pool_bytes, out, cap, frame_off, counts);
  if (!props || !prop_ents || n_prop_ents == 0 || (n_msgs && !msgs) || (n_ents && !ents) || (pool_bytes && !pool) || !out || cap == 0)
    return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: null argument");
  if (h->N < 2)
    return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: a single-peer group commits what it appends -- raftq_apply_log_deltas reports that");
  const uint64_t n_dev = n_props * (h->N - 1), n = n_msgs + n_dev, n_e = n_ents + n_prop_ents;
  if (n > kMaxItems || n_e > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: batch too large");
  if (int rc = ensure_pin(h)) return rc;
  void *v_props = dev_view(props), *v_pe = dev_view(prop_ents), *v_msgs = n_msgs ? dev_view(msgs) : nullptr, *v_ents = n_ents ? dev_view(ents) : nullptr,
       *v_pool = pool_bytes ? dev_view(pool) : nullptr, *v_out = dev_view(out), *v_off = frame_off ? dev_view(frame_off) : nullptr;
  const bool mapped = v_props && v_pe && (!n_msgs || v_msgs) && (!n_ents || v_ents) && (!pool_bytes || v_pool) && v_out && (!frame_off || v_off) &&
                      cap <= ((uint64_t)1 << 31);
  if (!(mapped && aligned16(v_props) && aligned16(v_pe) && aligned16(v_msgs) && aligned16(v_ents) && aligned16(v_pool) && aligned16(v_out) && aligned16(v_off)))
    return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: every array must be page-locked (raftq_host_alloc, hipHostMalloc, hipHostRegister) and "
                                 "16-byte aligned -- append with raftq_apply_log_deltas and marshal with raftq_wire_encode otherwise");
  NodeArrays na;
  if (int rc = raftq_detail::node_arrays_of(h, &na)) return rc;
  const uint32_t n_tiles = blocks_for(n);
  const unsigned workers = fused_grid(n_tiles);
  Carver fc;
  const void* const src[3] = {v_msgs, v_ents, v_pool};
  const uint64_t sizes[3] = {n_msgs * sizeof(WireMsg), n_ents * sizeof(WireEnt), pool_bytes};
  const uint64_t extra[3] = {n_dev * sizeof(WireMsg), n_prop_ents * sizeof(WireEnt), 0};
  TileCtl ctl;
  if (int rc = tile_ctl(h, std::max<uint64_t>(n_tiles, (sizes[0] + sizes[1] + sizes[2]) / feed_chunk() + 1), &ctl)) return rc;
  FeedPlan plan = plan_feed(fc, src, sizes, h->wire_lb_tiles, 48u, extra);
  const size_t o_props = fc.take(n_props * sizeof(PropRec)), o_pe = fc.take(n_prop_ents * sizeof(PropEnt));  // the check kernel's copies of the records
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, fc.off)) return rc;
  if (int rc = grow(h, &h->wire_out, &h->wire_out_bytes, cap + 16)) return rc;
  if (int rc = bind_feed(h, plan, (uint8_t*)h->wire_dev, ctl.status[kLbFlags])) return rc;
  // appendEntry + bcastAppend on the device, INTO the encoder's input (the scratch behind what its readers bring) ...
  // the validation's verdict: a word that holds THIS call's stamp when a record was refused (no memset in the chain: a stamp
  // of an earlier call reads as "fine")
  unsigned int* bad = (unsigned int*)(h->wire_flags + 2);
  if (++h->prop_stamp == 0) h->prop_stamp = 1;
  const unsigned int stamp = h->prop_stamp;
  WireMsg* msgs_dev = (WireMsg*)(plan.in.seg[0].dst + sizes[0]);
  WireEnt* ents_dev = (WireEnt*)(plan.in.seg[1].dst + sizes[1]);
  const dim3 pg((unsigned)((n_props + kBlock - 1) / kBlock)), cg((unsigned)((std::max(n_props, n_prop_ents) + kBlock - 1) / kBlock));
  PropRec* props_d = (PropRec*)((uint8_t*)h->wire_dev + o_props);
  PropEnt* pe_d = (PropEnt*)((uint8_t*)h->wire_dev + o_pe);
  hipLaunchKernelGGL(propose_check_kernel, cg, dim3(kBlock), 0, h->stream, na, (const PropRec*)v_props, n_props, (const PropEnt*)v_pe, n_prop_ents, pool_bytes, bad,
                     stamp, props_d, pe_d, h->wire_flags + 3);
  hipLaunchKernelGGL(propose_apply_kernel, pg, dim3(kBlock), 0, h->stream, na, (const PropRec*)props_d, n_props, (const PropEnt*)pe_d,
                     (const unsigned int*)bad, stamp, msgs_dev, ents_dev, (uint32_t)n_ents);
  // ... and the marshal of everything right behind it: one wait
  hipLaunchKernelGGL(wire_enc_fused_kernel, dim3(plan.in.readers + workers), dim3(kBlock), 0, h->stream, plan.in, n, n_e, pool_bytes, (uint8_t*)h->wire_out,
                     (uint8_t*)v_out, cap, (uint64_t*)v_off, ctl, h->wire_pin_d, (const unsigned int*)bad, stamp);
  HIPCHK(h, hipGetLastError());
  tile_ctl_launched(h, n_tiles, workers);
  HIPCHK(h, raftq_detail::wait_call(h));
  if (int rc = tile_ctl_check(h, "raftq_propose_frames")) return rc;
  const uint64_t total = h->wire_pin[0];
  if (h->wire_pin[1]) {
    // which record, and why (the check kernel left the largest (stamp, reason, record) it met; an older call's stamp: the marshal refused)
    unsigned long long why = 0;
    (void)hipMemcpy(&why, h->wire_flags + 3, 8, hipMemcpyDeviceToHost);
    static const char* const kWhy[] = {"?", "an entry's payload lies outside the pool", "its group is out of range", "it carries no entries (or more than 1024)",
                                       "its entries lie outside prop_ents[]", "this node does not lead its group", "its group is named twice"};
    const uint32_t reason = (uint32_t)(why >> 32) & 0xffu;
    if ((why >> 40) == (stamp & 0xffffffu) && reason >= 1 && reason <= 6)
      return fail(h, RAFTQ_EINVAL, std::string("raftq_propose_frames: record ") + std::to_string((uint32_t)why) + ": " + kWhy[reason] + " -- nothing was appended");
    return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: a queued message has to / from >= 255, an entry range outside ents[] or a payload outside the pool; the "
                                 "proposals WERE appended, the output is not valid");
  }
  if (counts) {
    counts->n_msgs = n;
    counts->n_ents = n_e;
    counts->bytes = total;
  }
  if (total > cap) return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: out is too small (counts->bytes is the size needed); the proposals WERE appended");
  return RAFTQ_OK;
}

// The streaming decode (raftq_wire_kernels.hpp "the streaming form"), enqueued on the handle's stream and NOT waited for;
// v_*: the caller's arrays as the device addresses them.  msgs_d / ff: see wire_dec_fused_kernel (raftq_step_frames).
static int decode_streaming_enqueue(raftq_t* h, const void* v_stream, uint64_t nbytes, const void* v_off, uint64_t n, void* v_msgs, void* v_ents,
                                    uint64_t ents_cap, WireMsg* msgs_d, FrameFilter ff) {
  const unsigned tb = dec_tile();
  const uint32_t n_tiles = (uint32_t)((n + tb - 1) / tb);
  // 128-frame tiles: every workgroup that fits (34 KB of LDS: four per CU, the readers' share taken off).  256-frame tiles: 208, one per
  // CU beside the readers as in round 5 -- at 68 KB two fit, but 464 workers measured SLOWER than 208 (173.9 against 167.6 us a call,
  // profiles/r06/wire_tile_ab.jsonl): with 256 tiles in a 64K-frame call every tile has its own waiting worker at 208 already, and
  // twice the workgroups are twice the pollers of the chunk flags
  const unsigned workers = fused_grid(n_tiles, tb == 128 ? 1024u - 96u : 208u);
  Carver c;
  const void* const src[3] = {v_off, v_stream, nullptr};
  const uint64_t bytes[3] = {(n + 1) * 8, nbytes, 0};
  TileCtl ctl;
  if (int rc = tile_ctl(h, std::max<uint64_t>(n_tiles, (bytes[0] + bytes[1]) / feed_chunk() + 1), &ctl)) return rc;
  FeedPlan plan = plan_feed(c, src, bytes, h->wire_lb_tiles, tb == 128 ? 96u : 48u);
  const size_t o_spill = c.take((size_t)n_tiles * tb * kEntQ * sizeof(WireEnt));  // a slot of kEntQ entry headers per lane
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, c.off)) return rc;
  const bool sdma = sdma_chunk() != 0;
  if (sdma) plan.in.readers = 0;
  if (int rc = bind_feed(h, plan, (uint8_t*)h->wire_dev, ctl.status[kLbFlags])) return rc;
  WireEnt* spill = (WireEnt*)((uint8_t*)h->wire_dev + o_spill);
  if (sdma) {
    plan.in.no_serve = 1;  // nobody draws a chunk ticket
    if (!h->wire_copy_stream) {
      HIPCHK(h, hipStreamCreateWithFlags(&h->wire_copy_stream, hipStreamNonBlocking));
      HIPCHK(h, hipEventCreateWithFlags(&h->wire_copy_ev, hipEventDisableTiming));
    }
    HIPCHK(h, hipEventRecord(h->wire_copy_ev, h->stream));  // the scratch's previous user is done before a copy lands in it
    HIPCHK(h, hipStreamWaitEvent(h->wire_copy_stream, h->wire_copy_ev, 0));
  }
  if (tb == 128)
    hipLaunchKernelGGL(wire_dec_fused_kernel<128>, dim3(plan.in.readers + workers), dim3(128), 0, h->stream, plan.in, nbytes, n, (WireMsg*)v_msgs,
                       (WireEnt*)v_ents, ents_cap, ctl, h->wire_pin_d, msgs_d, ff, spill);
  else
    hipLaunchKernelGGL(wire_dec_fused_kernel<256>, dim3(plan.in.readers + workers), dim3(256), 0, h->stream, plan.in, nbytes, n, (WireMsg*)v_msgs,
                       (WireEnt*)v_ents, ents_cap, ctl, h->wire_pin_d, msgs_d, ff, spill);
  HIPCHK(h, hipGetLastError());
  tile_ctl_launched(h, n_tiles, workers);
  h->wire_last_tiles = n_tiles;
  if (sdma) {  // behind the launch: the kernel's workers are already waiting for the flags
    const uint64_t word = lb_word(ctl.epoch, kLbInclusive, 0);
    for (uint32_t ck = 0; ck < plan.in.chunks; ++ck) {
      for (int k = 0; k < 2; ++k) {
        const FeedSeg& sg = plan.in.seg[k];
        const uint64_t lo = (uint64_t)ck * sg.per_chunk;
        if (sg.bytes == 0 || lo >= sg.bytes) continue;
        const uint64_t len = std::min<uint64_t>(sg.per_chunk, sg.bytes - lo);
        HIPCHK(h, hipMemcpyAsync(sg.dst + lo, sg.src + lo, len, hipMemcpyDefault, h->wire_copy_stream));
      }
      HIPCHK(h, hipStreamWriteValue64(h->wire_copy_stream, (void*)(plan.in.flag + ck), word, 0));
    }
  }
  return RAFTQ_OK;
}
// ... after the wait that covers it.  too_many_is_error: raftq_wire_decode's contract; raftq_step_frames only reports the count.
static int decode_streaming_finish(raftq_t* h, const char* who, const uint64_t* frame_off, uint64_t n, bool have_ents, uint64_t ents_cap,
                                   bool too_many_is_error, raftq_wire_counts_t* counts) {
#if defined(RAFTQ_WIRE_TRACE)
  trace_dump(h, "wire_dec", h->wire_last_tiles);
#endif
  if (int rc = tile_ctl_check(h, who)) return rc;
  const uint64_t total = h->wire_pin[0];
  if (counts) {
    counts->n_msgs = n;
    counts->n_ents = total;
    counts->n_malformed = h->wire_pin[1];
    counts->bytes = frame_off[n] >= frame_off[0] ? frame_off[n] - frame_off[0] : 0;
  }
  if (too_many_is_error && have_ents && total > ents_cap)
    return fail(h, RAFTQ_EINVAL, std::string(who) + ": more entries than ents_cap (counts->n_ents is the number needed)");
  return RAFTQ_OK;
}

}  // extern "C"
int raftq_detail::wire_frames_enqueue(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, void* msgs, void* ents,
                                      uint64_t ents_cap, void* msgs_d, int tail_appends, void* zero2) {
  if (n > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_step_frames: batch too large");
  if (int rc = ensure_pin(h)) return rc;
  void *v_stream = nullptr, *v_off = nullptr, *v_msgs = nullptr, *v_ents = nullptr;
  const bool mapped = (nbytes == 0 || (v_stream = dev_view(stream)) != nullptr) && (v_off = dev_view(frame_off)) != nullptr &&
                      (v_msgs = dev_view(msgs)) != nullptr && (!ents || (v_ents = dev_view(ents)) != nullptr);
  if (!(mapped && nbytes < (1ull << (kLbValueBits - 1)) && aligned16(v_stream) && aligned16(v_off) && aligned16(v_msgs) && aligned16(v_ents)))
    return fail(h, RAFTQ_EINVAL, "raftq_step_frames: the stream, the boundaries and the result arrays must be page-locked (raftq_host_alloc, "
                                 "hipHostMalloc, hipHostRegister) and 16-byte aligned -- decode and step in two calls otherwise");
  const FrameFilter ff{1u, h->N, h->self_peer, tail_appends ? 1u : 0u, h->G, (unsigned long long*)zero2};
  return decode_streaming_enqueue(h, v_stream, nbytes, v_off, n, v_msgs, v_ents, ents ? ents_cap : 0, (WireMsg*)msgs_d, ff);
}
int raftq_detail::wire_frames_finish(raftq_t* h, const uint64_t* frame_off, uint64_t n, bool have_ents, uint64_t ents_cap, raftq_wire_counts_t* counts) {
  return decode_streaming_finish(h, "raftq_step_frames", frame_off, n, have_ents, ents_cap, false, counts);
}
extern "C" {

int raftq_wire_decode(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n,
                      raftq_wire_msg_t* msgs, raftq_wire_ent_t* ents, uint64_t ents_cap, raftq_wire_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (n == 0) return RAFTQ_OK;
  if ((!stream && nbytes) || !frame_off || !msgs) return fail(h, RAFTQ_EINVAL, "raftq_wire_decode: null argument");
  if (n > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_wire_decode: batch too large");
  if (!ents) ents_cap = 0;
  if (int rc = ensure_pin(h)) return rc;
  // an entry costs its message at least two bytes (tag, length), so this many can never be exceeded
  const uint64_t dev_cap = std::min<uint64_t>(ents_cap, nbytes / 2 + 1);
  void *v_stream = nullptr, *v_off = nullptr, *v_msgs = nullptr, *v_ents = nullptr;
  const bool mapped = streaming_on() && (nbytes == 0 || (v_stream = dev_view(stream)) != nullptr) && (v_off = dev_view(frame_off)) != nullptr &&
                      (v_msgs = dev_view(msgs)) != nullptr && (!ents || (v_ents = dev_view(ents)) != nullptr);
  // (every array, inputs and outputs: the workers store whole 16-byte quads into msgs / ents -- raftq_wire.h "odd alignment")
  if (mapped && nbytes < (1ull << (kLbValueBits - 1)) && aligned16(v_stream) && aligned16(v_off) && aligned16(v_msgs) && aligned16(v_ents)) {
    // page-locked caller buffers: ONE kernel -- readers bring boundaries and stream into the scratch in order, workers parse
    // tile by tile behind them and push records and entry headers out (raftq_wire_kernels.hpp "the streaming form")
    if (int rc = decode_streaming_enqueue(h, v_stream, nbytes, v_off, n, v_msgs, v_ents, ents_cap, nullptr, FrameFilter{0, 0, 0, 0, 0, nullptr})) return rc;
    HIPCHK(h, raftq_detail::wait_call(h));
    return decode_streaming_finish(h, "raftq_wire_decode", frame_off, n, ents != nullptr, ents_cap, true, counts);
  }
  const size_t scan_bytes = scan_sum_scratch_bytes(n + 1);  // tile totals of the hand-written scan
  Carver c;
  const size_t o_stream = c.take(nbytes), o_off = c.take((n + 1) * 8), o_msgs = c.take(n * sizeof(WireMsg)),
               o_cnt = c.take((n + 1) * 8), o_base = c.take((n + 1) * 8), o_bad = c.take(8), o_scan = c.take(scan_bytes);
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, c.off)) return rc;
  if (int rc = grow(h, &h->wire_out, &h->wire_out_bytes, dev_cap * sizeof(WireEnt) + 16)) return rc;
  uint8_t* base = (uint8_t*)h->wire_dev;
  uint8_t* d_stream = base + o_stream;
  uint64_t *d_off = (uint64_t*)(base + o_off), *d_cnt = (uint64_t*)(base + o_cnt), *d_base = (uint64_t*)(base + o_base);
  WireMsg* d_msgs = (WireMsg*)(base + o_msgs);
  WireEnt* d_ents = (WireEnt*)h->wire_out;
  unsigned long long* d_bad = h->wire_flags + 1;  // (o_bad: unused since the flags have a block of their own)
  (void)o_bad;
  if (int rc = h2d(h, d_stream, stream, nbytes)) return rc;
  if (int rc = h2d(h, d_off, frame_off, (n + 1) * 8)) return rc;
  
  hipLaunchKernelGGL(wire_dec_kernel, dim3(blocks_for(n + 1)), dim3(kBlock), 0, h->stream, (const uint8_t*)d_stream,
                     nbytes, (const uint64_t*)d_off, n, d_msgs, d_cnt, d_bad);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, exclusive_sum_u64((const uint64_t*)d_cnt, d_base, n + 1, (uint64_t*)(base + o_scan), h->stream));
  hipLaunchKernelGGL(wire_dec_ents_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, (const uint8_t*)d_stream,
                     nbytes, (const uint64_t*)d_off, n, d_msgs, (const uint64_t*)d_base, dev_cap ? d_ents : (WireEnt*)nullptr,
                     dev_cap);
  HIPCHK(h, hipGetLastError());
  if (int rc = tail_to_pin(h, d_base + n, d_bad)) return rc;
  if (int rc = d2h(h, msgs, d_msgs, n * sizeof(WireMsg))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint64_t total = h->wire_pin[0];
  if (counts) {
    counts->n_msgs = n;
    counts->n_ents = total;
    counts->n_malformed = h->wire_pin[1];
    counts->bytes = frame_off[n] >= frame_off[0] ? frame_off[n] - frame_off[0] : 0;
  }
  if (!ents) return RAFTQ_OK;  // headers only
  if (total > ents_cap)
    return fail(h, RAFTQ_EINVAL, "raftq_wire_decode: more entries than ents_cap (counts->n_ents is the number needed)");
  if (total) {
    if (int rc = d2h(h, ents, d_ents, total * sizeof(WireEnt))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return RAFTQ_OK;
}

// The streaming WAL encode, enqueued and NOT waited for; its totals go to wire_pin[pin_base ..] (raftq_wal_encode: 0;
// raftq_wal_encode_begin: 8, so that the call enqueued behind it can use the words at 0).
static int wal_streaming_enqueue(raftq_t* h, const void* v_recs, uint64_t n, const void* v_pool, uint64_t pool_bytes, uint32_t prev_crc, void* v_out,
                                 uint64_t cap, void* v_off, uint32_t pin_base) {
  const uint32_t n_tiles = blocks_for(n);
  const unsigned workers = fused_grid(n_tiles);
  Carver fc;
  const void* const src[3] = {v_recs, v_pool, nullptr};
  const uint64_t sizes[3] = {n * sizeof(WalRec), pool_bytes, 0};
  TileCtl ctl;
  if (int rc = tile_ctl(h, std::max<uint64_t>(n_tiles, (sizes[0] + sizes[1]) / feed_chunk() + 1), &ctl)) return rc;
  FeedPlan plan = plan_feed(fc, src, sizes, h->wire_lb_tiles);
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, fc.off)) return rc;
  if (int rc = grow(h, &h->wire_out, &h->wire_out_bytes, cap + 16)) return rc;
  if (int rc = bind_feed(h, plan, (uint8_t*)h->wire_dev, ctl.status[kLbFlags])) return rc;
  hipLaunchKernelGGL(wal_enc_fused_kernel, dim3(plan.in.readers + workers), dim3(kBlock), 0, h->stream, plan.in, n, pool_bytes, prev_crc,
                     (uint8_t*)h->wire_out, (uint8_t*)v_out, cap, (uint64_t*)v_off, ctl, h->wire_pin_d + pin_base);
  HIPCHK(h, hipGetLastError());
  tile_ctl_launched(h, n_tiles, workers);
  return RAFTQ_OK;
}
static int wal_streaming_finish(raftq_t* h, const char* who, uint64_t n, uint64_t cap, uint32_t prev_crc, uint32_t pin_base, raftq_wal_counts_t* counts) {
  if (counts) {
    *counts = raftq_wal_counts_t{0, 0, 0, 0, 0};
    counts->last_crc = prev_crc;
  }
  if (int rc = tile_ctl_check(h, who, pin_base)) return rc;
  const uint64_t* pin = h->wire_pin + pin_base;
  const uint64_t total = pin[0];
  if (pin[1])
    return fail(h, RAFTQ_EINVAL, std::string(who) + ": a record has an unknown kind or a payload outside the pool; the output is not valid");
  if (counts) {
    counts->n_recs = n;
    counts->bytes = total;
  }
  if (total > cap) return fail(h, RAFTQ_EINVAL, std::string(who) + ": out is too small (counts->bytes is the size needed)");
  if (counts) {
    counts->n_valid = n;
    counts->last_crc = (uint32_t)pin[2];
  }
  return RAFTQ_OK;
}
// a raftq_wal_encode_begin whose _end has not come yet: wait for it and keep what _end will report (called by whatever else
// is about to use its pinned words' neighbours' scratch from the host side)
static int wal_pending_complete(raftq_t* h) {
  if (!h->wal_pending || h->wal_pending_done) return RAFTQ_OK;
  // (the wait of the marshal called in between has usually covered it: then there is nothing to wait for, and nothing to launch)
  if (!h->wal_pending_waited) HIPCHK(h, raftq_detail::wait_call(h));
  h->wal_pending_rc = wal_streaming_finish(h, "raftq_wal_encode_begin", h->wal_pending_n, h->wal_pending_cap, h->wal_pending_prev, 8, &h->wal_pending_counts);
  if (h->wal_pending_rc != RAFTQ_OK) h->wal_pending_err = h->err;
  h->wal_pending_done = true;
  return RAFTQ_OK;
}

int raftq_wal_encode(raftq_t* h, const raftq_wal_rec_t* recs, uint64_t n, const void* pool, uint64_t pool_bytes,
                     uint32_t prev_crc, void* out, uint64_t cap, uint64_t* frame_off, raftq_wal_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) {
    *counts = raftq_wal_counts_t{0, 0, 0, 0, 0};
    counts->last_crc = prev_crc;
  }
  if (n == 0) {
    if (frame_off) frame_off[0] = 0;
    return RAFTQ_OK;
  }
  if (!recs || (pool_bytes && !pool) || (cap && !out)) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: null argument");
  if (n > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: batch too large");
  if (int rc = ensure_pin(h)) return rc;
  // page-locked caller buffers take the streaming form; anything else the copying form
  void *v_recs = nullptr, *v_pool = nullptr, *v_out = nullptr, *v_off = nullptr;
  const bool mapped = streaming_on() && cap != 0 && cap <= ((uint64_t)1 << 31) && (v_recs = dev_view(recs)) != nullptr &&
                      (pool_bytes == 0 || (v_pool = dev_view(pool)) != nullptr) && (v_out = dev_view(out)) != nullptr &&
                      (!frame_off || (v_off = dev_view(frame_off)) != nullptr);
  if (mapped && aligned16(v_recs) && aligned16(v_pool) && aligned16(v_out) && aligned16(v_off)) {
    // page-locked caller buffers: the streaming form (readers | workers in one launch; raftq_wire_kernels.hpp)
    if (int rc = wal_pending_complete(h)) return rc;  // (a raftq_wal_encode_begin nobody ended: its results are kept for its _end)
    if (int rc = wal_streaming_enqueue(h, v_recs, n, v_pool, pool_bytes, prev_crc, v_out, cap, v_off, 0)) return rc;
    HIPCHK(h, raftq_detail::wait_call(h));
    return wal_streaming_finish(h, "raftq_wal_encode", n, cap, prev_crc, 0, counts);
  }
  const size_t scan_bytes = scan_sum_scratch_bytes(n + 1);  // tile totals of the hand-written scan
  Carver c;
  const size_t o_recs = c.take(n * sizeof(WalRec)), o_pool = c.take(pool_bytes), o_pcrc = c.take(n * 4),
               o_pair = c.take(n * 8), o_chain = c.take(n * 8), o_sizes = c.take((n + 1) * 8),
               o_off = c.take((n + 1) * 8), o_flags = c.take(8), o_scan = c.take(scan_bytes),
               o_tot = c.take((size_t)blocks_for(n) * sizeof(CrcPair));
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, c.off)) return rc;
  uint8_t* base = (uint8_t*)h->wire_dev;
  WalRec* d_recs = (WalRec*)(base + o_recs);
  uint8_t* d_pool = base + o_pool;
  uint32_t* d_pcrc = (uint32_t*)(base + o_pcrc);
  CrcPair *d_pair = (CrcPair*)(base + o_pair), *d_chain = (CrcPair*)(base + o_chain);
  uint64_t *d_sizes = (uint64_t*)(base + o_sizes), *d_off = (uint64_t*)(base + o_off);
  uint32_t* d_last = (uint32_t*)(base + o_flags) + 1;
  unsigned int* d_bad = (unsigned int*)(base + o_flags);
  if (int rc = h2d(h, d_recs, recs, n * sizeof(WalRec))) return rc;
  if (int rc = h2d(h, d_pool, pool, pool_bytes)) return rc;
  HIPCHK(h, hipMemsetAsync(d_bad, 0, 8, h->stream));
  
  hipLaunchKernelGGL(wal_enc_payload_crc_kernel, dim3(blocks_for(n * 64)), dim3(kBlock), 0, h->stream,
                     (const WalRec*)d_recs, n, (const uint8_t*)d_pool, pool_bytes, d_pcrc);
  hipLaunchKernelGGL(wal_enc_crc_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, (const WalRec*)d_recs, n,
                     (const uint8_t*)d_pool, pool_bytes, (const uint32_t*)d_pcrc, prev_crc, d_pair, d_bad);
  HIPCHK(h, hipGetLastError());
  if (int rc = crc_chain_scan(h, d_pair, d_chain, n, (CrcPair*)(base + o_tot))) return rc;
  hipLaunchKernelGGL(wal_enc_size_kernel, dim3(blocks_for(n + 1)), dim3(kBlock), 0, h->stream, (const WalRec*)d_recs, n,
                     (const CrcPair*)d_chain, d_sizes);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, exclusive_sum_u64((const uint64_t*)d_sizes, d_off, n + 1, (uint64_t*)(base + o_scan), h->stream));
  if (int rc = d2h(h, &h->wire_pin[0], d_off + n, 8)) return rc;
  if (int rc = d2h(h, &h->wire_pin[1], d_bad, 4)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint64_t total = h->wire_pin[0];
  if ((uint32_t)h->wire_pin[1])
    return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: a record has an unknown kind or a payload outside the pool; nothing was written");
  if (counts) {
    counts->n_recs = n;
    counts->bytes = total;
  }
  if (total > cap) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: out is too small (counts->bytes is the size needed)");
  if (int rc = grow(h, &h->wire_out, &h->wire_out_bytes, total + 16)) return rc;
  uint8_t* d_out = (uint8_t*)h->wire_out;
  hipLaunchKernelGGL(wal_enc_write_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, (const WalRec*)d_recs, n,
                     (const CrcPair*)d_chain, (const uint64_t*)d_off, d_out, d_last);
  if (pool_bytes)
    hipLaunchKernelGGL(wal_enc_payload_kernel, dim3(blocks_for(n * 64)), dim3(kBlock), 0, h->stream,
                       (const WalRec*)d_recs, n, (const CrcPair*)d_chain, (const uint64_t*)d_off,
                       (const uint8_t*)d_pool, d_out);
  HIPCHK(h, hipGetLastError());
  if (int rc = d2h(h, out, d_out, total)) return rc;
  if (frame_off)
    if (int rc = d2h(h, frame_off, d_off, (n + 1) * 8)) return rc;
  if (int rc = d2h(h, &h->wire_pin[2], d_last, 4)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (counts) {
    counts->n_valid = n;
    counts->last_crc = (uint32_t)h->wire_pin[2];
  }
  return RAFTQ_OK;
}

int raftq_wal_encode_begin(raftq_t* h, const raftq_wal_rec_t* recs, uint64_t n, const void* pool, uint64_t pool_bytes, uint32_t prev_crc,
                           void* out, uint64_t cap, uint64_t* frame_off) {
  if (int rc = use_device(h)) return rc;
  if (h->wal_pending) return fail(h, RAFTQ_ESTATE, "raftq_wal_encode_begin: the previous one has not been ended (raftq_wal_encode_end)");
  if (n == 0 || !recs || (pool_bytes && !pool) || !out || cap == 0) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode_begin: null argument or empty batch");
  if (n > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode_begin: batch too large");
  if (int rc = ensure_pin(h)) return rc;
  void *v_recs = nullptr, *v_pool = nullptr, *v_out = nullptr, *v_off = nullptr;
  const bool mapped = cap <= ((uint64_t)1 << 31) && (v_recs = dev_view(recs)) != nullptr && (pool_bytes == 0 || (v_pool = dev_view(pool)) != nullptr) &&
                      (v_out = dev_view(out)) != nullptr && (!frame_off || (v_off = dev_view(frame_off)) != nullptr);
  if (!(mapped && aligned16(v_recs) && aligned16(v_pool) && aligned16(v_out) && aligned16(v_off)))
    return fail(h, RAFTQ_EINVAL, "raftq_wal_encode_begin: the records, the pool and the output must be page-locked (raftq_host_alloc, hipHostMalloc, "
                                 "hipHostRegister) and 16-byte aligned -- raftq_wal_encode otherwise");
  if (int rc = wal_streaming_enqueue(h, v_recs, n, v_pool, pool_bytes, prev_crc, v_out, cap, v_off, 8)) return rc;
  h->wal_pending = true;
  h->wal_pending_done = false;
  h->wal_pending_waited = false;
  h->wal_pending_n = n;
  h->wal_pending_cap = cap;
  h->wal_pending_prev = prev_crc;
  return RAFTQ_OK;
}

int raftq_wal_encode_end(raftq_t* h, raftq_wal_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (!h->wal_pending) return fail(h, RAFTQ_ESTATE, "raftq_wal_encode_end: nothing was begun");
  if (int rc = wal_pending_complete(h)) return rc;  // (no wait left to make when a later call on the handle has waited already)
  h->wal_pending = false;
  if (counts) *counts = h->wal_pending_counts;
  if (h->wal_pending_rc != RAFTQ_OK) return fail(h, h->wal_pending_rc, h->wal_pending_err);
  return RAFTQ_OK;
}

int raftq_wal_decode(raftq_t* h, const void* bytes, uint64_t nbytes, const uint64_t* frame_off, uint64_t n,
                     uint32_t prev_crc, raftq_wal_rec_t* recs, raftq_wal_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) {
    *counts = raftq_wal_counts_t{0, 0, 0, 0, 0};
    counts->last_crc = prev_crc;
  }
  if (n == 0) return RAFTQ_OK;
  if ((!bytes && nbytes) || !frame_off || !recs) return fail(h, RAFTQ_EINVAL, "raftq_wal_decode: null argument");
  if (n > kMaxItems) return fail(h, RAFTQ_EINVAL, "raftq_wal_decode: batch too large");
  if (int rc = ensure_pin(h)) return rc;
  {
    void *f_bytes = nullptr, *f_off = nullptr, *f_recs = nullptr;
    if (streaming_on() && nbytes < (1ull << (kLbValueBits - 1)) && (nbytes == 0 || (f_bytes = dev_view(bytes)) != nullptr) &&
        (f_off = dev_view(frame_off)) != nullptr && (f_recs = dev_view(recs)) != nullptr && aligned16(f_bytes) && aligned16(f_off) && aligned16(f_recs)) {
      // page-locked caller buffers: the streaming form (readers | workers in one launch)
      const uint32_t n_tiles = blocks_for(n);
      const unsigned workers = fused_grid(n_tiles);
      Carver c;
      const void* const src[3] = {f_off, f_bytes, nullptr};
      const uint64_t sizes[3] = {(n + 1) * 8, nbytes, 0};
      TileCtl ctl;
      if (int rc = tile_ctl(h, std::max<uint64_t>(n_tiles, (sizes[0] + sizes[1]) / feed_chunk() + 1), &ctl)) return rc;
      FeedPlan plan = plan_feed(c, src, sizes, h->wire_lb_tiles);
      if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, c.off)) return rc;
      if (int rc = bind_feed(h, plan, (uint8_t*)h->wire_dev, ctl.status[kLbFlags])) return rc;
      hipLaunchKernelGGL(wal_dec_fused_kernel, dim3(plan.in.readers + workers), dim3(kBlock), 0, h->stream, plan.in, nbytes, n, prev_crc,
                         (WalRec*)f_recs, ctl, h->wire_pin_d);
      HIPCHK(h, hipGetLastError());
      tile_ctl_launched(h, n_tiles, workers);
      HIPCHK(h, raftq_detail::wait_call(h));
      if (int rc = tile_ctl_check(h, "raftq_wal_decode")) return rc;
      if (counts) {
        counts->n_recs = n;
        counts->n_valid = h->wire_pin[0];
        counts->bytes = frame_off[n] >= frame_off[0] ? frame_off[n] - frame_off[0] : 0;
        counts->last_crc = (uint32_t)h->wire_pin[1];
      }
      return RAFTQ_OK;
    }
  }
  Carver c;
  const size_t o_bytes = c.take(nbytes), o_off = c.take((n + 1) * 8), o_recs = c.take(n * sizeof(WalRec)),
               o_span = c.take(n * sizeof(WalSpan)), o_pair = c.take(n * 8), o_chain = c.take(n * 8),
               o_tail = c.take(32), o_tot = c.take((size_t)blocks_for(n) * sizeof(CrcPair));
  if (int rc = grow(h, &h->wire_dev, &h->wire_dev_bytes, c.off)) return rc;
  uint8_t* base = (uint8_t*)h->wire_dev;
  uint8_t* d_bytes = base + o_bytes;
  uint64_t* d_off = (uint64_t*)(base + o_off);
  WalRec* d_recs = (WalRec*)(base + o_recs);
  WalSpan* d_span = (WalSpan*)(base + o_span);
  CrcPair *d_pair = (CrcPair*)(base + o_pair), *d_chain = (CrcPair*)(base + o_chain);
  unsigned long long* d_first_bad = (unsigned long long*)(base + o_tail);
  uint64_t* d_tail = (uint64_t*)(base + o_tail) + 1;
  void *v_bytes = nullptr, *v_off = nullptr, *v_recs = nullptr;
  const bool mapped = streaming_on() && (nbytes == 0 || (v_bytes = dev_view(bytes)) != nullptr) && (v_off = dev_view(frame_off)) != nullptr &&
                      (v_recs = dev_view(recs)) != nullptr;
  if (int rc = h2d(h, d_bytes, bytes, nbytes)) return rc;
  if (int rc = h2d(h, d_off, frame_off, (n + 1) * 8)) return rc;
  
  HIPCHK(h, hipMemsetAsync(d_first_bad, 0xff, 8, h->stream));
  hipLaunchKernelGGL(wal_dec_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, (const uint8_t*)d_bytes, nbytes,
                     (const uint64_t*)d_off, n, prev_crc, d_recs, d_span, d_pair);
  hipLaunchKernelGGL(wal_dec_long_crc_kernel, dim3(blocks_for(n * 64)), dim3(kBlock), 0, h->stream,
                     (const uint8_t*)d_bytes, n, (const WalSpan*)d_span, prev_crc, d_pair);
  HIPCHK(h, hipGetLastError());
  if (int rc = crc_chain_scan(h, d_pair, d_chain, n, (CrcPair*)(base + o_tot))) return rc;
  hipLaunchKernelGGL(wal_dec_check_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, h->stream, d_recs, n,
                     (const CrcPair*)d_chain, prev_crc, d_first_bad);
  hipLaunchKernelGGL(wal_dec_tail_kernel, dim3(1), dim3(64), 0, h->stream, (const CrcPair*)d_chain, n, prev_crc,
                     (const unsigned long long*)d_first_bad, d_tail);
  HIPCHK(h, hipGetLastError());
  if (int rc = d2h(h, recs, d_recs, n * sizeof(WalRec))) return rc;
  if (int rc = d2h(h, &h->wire_pin[0], d_tail, 16)) return rc;
  
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (counts) {
    counts->n_recs = n;
    counts->n_valid = h->wire_pin[0];
    counts->bytes = frame_off[n] >= frame_off[0] ? frame_off[n] - frame_off[0] : 0;
    counts->last_crc = (uint32_t)h->wire_pin[1];
  }
  return RAFTQ_OK;
}

}  // extern "C"
