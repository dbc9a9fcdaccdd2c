// raftq_step.hip -- implementation of include/raftq_step.h: the batched raft Step over
// the device-resident group state.  Host side: validate + stage the batch in pinned
// memory, key/sort/walk it on the handle's stream (raftq_step_kernels.hpp), copy the
// result records back.  No CPU path: the state machine itself runs only on the GPU.
#include "raftq_step.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "raftq_internal.hpp"
#include "raftq_step_kernels.hpp"
#include "raftq_wire.h"
#include "raftq_sort_kernels.hpp"
#include "raftq_wire_kernels.hpp"

using namespace raftqk;
using raftq_detail::ensure_staging;
using raftq_detail::ensure_tick_state;
using raftq_detail::fail;
using raftq_detail::use_device;

static_assert(sizeof(raftq_msg_t) == sizeof(MsgRec) && sizeof(raftq_step_out_t) == sizeof(StepOutRec) &&
                  sizeof(raftq_log_delta_t) == sizeof(LogDeltaRec) && sizeof(raftq_step_out_c_t) == sizeof(StepOutC) &&
                  sizeof(raftq_step_out_s_t) == sizeof(raftqk::StepOutS),
              "ABI struct mismatch");

namespace {

// bytes of one result record in the handle's format
uint32_t result_rec_bytes(const raftq_t* h) {
  return h->step_compact == raftqk::kFmtS32 ? (uint32_t)sizeof(raftqk::StepOutS) : h->step_compact ? (uint32_t)sizeof(StepOutC) : (uint32_t)sizeof(StepOutRec);
}

int ensure_node_state(raftq_t* h) {
  if (h->node_rec) return RAFTQ_OK;
  if (int rc = ensure_tick_state(h)) return rc;  // role, elapsed
  auto alloc = [&](void** p, size_t bytes) -> int {
    HIPCHK(h, hipMalloc(p, bytes));
    HIPCHK(h, hipMemsetAsync(*p, 0, bytes, h->stream));
    return RAFTQ_OK;
  };
  if (!h->step_stall)  // (a retry after a failed launch below finds it allocated: ADVICE r05, 256 bytes leaked per failed attempt)
    if (int rc = alloc((void**)&h->step_stall, 256)) return rc;
  if (const char* w = std::getenv("RAFTQ_STEP_WALK")) h->step_walk_mode = std::strcmp(w, "sort") == 0 ? 0 : 1;
  // h->node_rec marks the state complete: it is assigned only once the records have been allocated AND their initialisation has
  // been launched (a failed launch used to leave the pointer set, and the next call stepped uninitialised records)
  void* rec = nullptr;
  HIPCHK(h, hipMalloc(&rec, h->ld * sizeof(NodeRec)));
  hipLaunchKernelGGL(node_init_kernel, dim3((unsigned)((h->ld + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, (NodeRec*)rec, h->ld);
  if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
    (void)hipFree(rec);
    HIPCHK(h, e);
  }
  h->node_rec = (decltype(h->node_rec))rec;
  h->have_terms = true;  // Step maintains the current-term gate itself (closed = 0 until a group leads)
  return RAFTQ_OK;
}

NodeArrays node_arrays(raftq_t* h, uint8_t recs = raftqk::kRecsCaller) {
  NodeArrays a;
  a.rec = (NodeRec*)h->node_rec;
  a.role = h->role;
  a.elapsed = h->elapsed;
  a.committed = h->committed[h->cur];
  a.first_idx = h->first_idx;
  a.match = h->match;
  a.votes = h->votes;
  a.ld = h->ld;
  a.n_peers = h->N;
  a.self = h->self_peer;
  a.msg_flags = h->step_msg_flags;
  a.recs = recs;
  a.n_groups = h->G;
  return a;
}

// The records' copies of role / committed / first_idx / match (raftq_step_kernels.hpp NodeRec) are re-read from the dense arrays
// when something other than Step has changed those since (raftq_capi.hip marks it): enqueued in front of whatever reads a record.
int ensure_mirror(raftq_t* h) {
  if (h->node_mirror_fresh) return RAFTQ_OK;
  hipLaunchKernelGGL(node_mirror_kernel, dim3((unsigned)((h->G + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, node_arrays(h), h->G);
  HIPCHK(h, hipGetLastError());
  h->node_mirror_fresh = true;
  h->node_mirror_refreshes++;
  return RAFTQ_OK;
}

}  // namespace
int raftq_detail::node_arrays_of(raftq_t* h, raftqk::NodeArrays* out) {
  if (int rc = ensure_node_state(h)) return rc;
  if (int rc = ensure_mirror(h)) return rc;
  *out = node_arrays(h);
  h->last_flags &= ~RAFTQ_SWEEP_NO_ADOPT;  // as in raftq_step_submit: the live state moves
  return RAFTQ_OK;
}
namespace {

// device scratch of one batch, carved from a single allocation
struct Scratch {
  MsgRec* msgs;
  void* msgs40;  // packed inbound records as they arrived (raftq_step_submit_packed), widened into msgs on the device
  void* outs;  // StepOutRec[n] or StepOutC[n], then the 16-byte tail at tail_off(n, rec)
  uint64_t *keys_in, *keys_out;
  uint32_t *order_in, *order_out;
  uint32_t* next;  // sort-free walk: list links
  unsigned long long* n_heads;
  void* sort_scratch;
  size_t sort_bytes;
  // raftq_step_submit_wire only: the staged frames and what the decoder needs
  uint64_t* w_off;       // [n + 1] frame offsets, the stream bytes right behind them
  uint8_t* w_stream;
  uint64_t *w_cnt, *w_base;  // [n + 1] entries per message, their exclusive scan
  unsigned long long* w_bad;
  WireEnt* w_ents;
  uint64_t w_ents_cap;
  void* w_scan;
  size_t w_scan_bytes;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }
// result area: n records of `rec` bytes, then (16-byte aligned) the {touched count, bad, skipped} tail
size_t tail_off(uint64_t n, uint32_t rec) { return ((size_t)n * rec + 15) / 16 * 16; }

int ensure_slot(raftq_t* h, raftq::StepSlot& sl, uint64_t n, int end_bit, Scratch* s, bool wire = false,
                uint64_t wire_nbytes = 0) {
  if (!h->step_s_in) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->step_s_in, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->step_s_out, hipStreamNonBlocking));
    // Stream topology.  Default 2: the H2D DMA on its own stream (overlaps the previous batch's kernels),
    // the result copy on the handle's stream.  Measured on MI355X (profiles/r01/step_pipeline_trace.txt):
    // a kernel that writes to host memory over PCIe keeps every other queue's NEXT kernel from starting
    // until it retires (a kernel that merely spins does not), so a third stream for the D2H buys nothing
    // (3: 186 us per 64K batch, 2: 176 us, 1 = everything on one stream: 426 us, 4 = copies share a stream: 459 us).
    // Round 2 of that A/B (profiles/r01/step_result_copy_ab.txt): the runtime's own D2H memcpy instead of our kernel
    // (382-394 us), the copy split into 2-32 short kernels (177-185 us), the walk writing straight into mapped host
    // memory (186 us) -- none beat 2; those variants are no longer in the code.  RAFTQ_STEP_STREAMS = 1..4 overrides.
    if (const char* m = std::getenv("RAFTQ_STEP_STREAMS")) h->step_stream_mode = std::atoi(m);
    if (const char* m = std::getenv("RAFTQ_STEP_DEFER_COPY")) h->step_defer_copy = std::atoi(m) != 0;
    if (const char* m = std::getenv("RAFTQ_STEP_LINK_SHARE")) h->step_link_share = std::min(100, std::max(0, std::atoi(m)));
  }
  if (!sl.ev_in) {
    HIPCHK(h, hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&sl.ev_comp, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
  }
  (void)end_bit;
  const size_t sort_bytes = raftqk::radix_plan(n).bytes;  // digit counts of the hand-written radix sort (raftq_sort_kernels.hpp)
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
  // the 16-byte {touched count, bad flag} tail sits right behind the result records: one copy moves both
  const size_t o_msgs = carve(n * sizeof(MsgRec)), o_outs = carve(n * sizeof(StepOutRec) + 16), o_ki = carve(n * 8),
               o_ko = carve(n * 8), o_oi = carve(n * 4), o_oo = carve(n * 4), o_sort = carve(sort_bytes), o_next = carve(n * 4),
               o_m40 = carve(n * sizeof(raftqk::Msg40Rec));
  const size_t o_nh = o_outs + tail_off(n, result_rec_bytes(h));
  // decoded entry headers: an entry costs its message two bytes at least, so nbytes / 2 + 1 always suffice
  const uint64_t w_ents_cap = wire ? wire_nbytes / 2 + 1 : 0;
  size_t w_scan_bytes = 0, o_wfr = 0, o_wcnt = 0, o_wbase = 0, o_wbad = 0, o_wents = 0, o_wscan = 0;
  if (wire) {
    w_scan_bytes = scan_sum_scratch_bytes(n + 1);  // tile totals of the hand-written scan (raftq_wire_kernels.hpp)
    o_wfr = carve((n + 1) * 8 + wire_nbytes + 16);
    o_wcnt = carve((n + 1) * 8);
    o_wbase = carve((n + 1) * 8);
    o_wbad = carve(8);
    o_wents = carve(w_ents_cap * sizeof(WireEnt));
    o_wscan = carve(w_scan_bytes);
  }
  if (off > sl.dev_bytes) {  // the slot is idle (its previous batch was collected): safe to regrow
    if (sl.dev) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      HIPCHK(h, hipFree(sl.dev));
      sl.dev = nullptr;
      sl.dev_bytes = 0;
    }
    const size_t want = std::max(off, (size_t)1 << 20) * 2;
    HIPCHK(h, hipMalloc(&sl.dev, want));
    sl.dev_bytes = want;
  }
  const size_t out_bytes = (size_t)n * sizeof(StepOutRec) + 16;
  if (out_bytes > sl.out_bytes) {
    if (sl.out_h) {
      if (h->step_last_out == sl.out_h) h->step_last_out = nullptr, h->step_last_n = 0;
      HIPCHK(h, hipHostFree(sl.out_h));
      sl.out_h = sl.out_d = nullptr;
      sl.out_bytes = 0;
    }
    const size_t want = std::max(out_bytes * 2, (size_t)1 << 20);
    HIPCHK(h, hipHostMalloc(&sl.out_h, want, hipHostMallocMapped));
    HIPCHK(h, hipHostGetDevicePointer(&sl.out_d, sl.out_h, 0));
    sl.out_bytes = want;
  }
  uint8_t* base = (uint8_t*)sl.dev;
  s->msgs = (MsgRec*)(base + o_msgs);
  s->msgs40 = base + o_m40;
  s->outs = base + o_outs;
  s->keys_in = (uint64_t*)(base + o_ki);
  s->keys_out = (uint64_t*)(base + o_ko);
  s->order_in = (uint32_t*)(base + o_oi);
  s->order_out = (uint32_t*)(base + o_oo);
  s->next = (uint32_t*)(base + o_next);
  s->n_heads = (unsigned long long*)(base + o_nh);
  s->sort_scratch = base + o_sort;
  s->sort_bytes = sort_bytes;
  s->w_off = (uint64_t*)(base + o_wfr);
  s->w_stream = base + o_wfr + (n + 1) * 8;
  s->w_cnt = (uint64_t*)(base + o_wcnt);
  s->w_base = (uint64_t*)(base + o_wbase);
  s->w_bad = (unsigned long long*)(base + o_wbad);
  s->w_ents = (WireEnt*)(base + o_wents);
  s->w_ents_cap = w_ents_cap;
  s->w_scan = base + o_wscan;
  s->w_scan_bytes = w_scan_bytes;
  return RAFTQ_OK;
}

int ensure_slot_staging(raftq_t* h, raftq::StepSlot& sl, uint64_t n, size_t raw_bytes = 0) {
  const size_t bytes = raw_bytes ? raw_bytes : (size_t)std::max<uint64_t>(n, 1) * sizeof(raftq_msg_t);
  if (bytes <= sl.in_bytes) return RAFTQ_OK;
  if (sl.in_h) {
    HIPCHK(h, hipHostFree(sl.in_h));
    sl.in_h = nullptr;
    sl.in_bytes = 0;
  }
  const size_t want = std::max(bytes * 2, (size_t)1 << 20);
  HIPCHK(h, hipHostMalloc(&sl.in_h, want, hipHostMallocDefault));
  sl.in_bytes = want;
  return RAFTQ_OK;
}

// the in-place staging of raftq_step_stage: device memory the host can write (large BAR), else the pinned buffer
int ensure_slot_bar(raftq_t* h, raftq::StepSlot& sl, uint64_t n, void** out, size_t raw_bytes = 0) {
  const size_t bytes = raw_bytes ? raw_bytes : (size_t)std::max<uint64_t>(n, 1) * sizeof(raftq_msg_t);
  if (h->bar_staging) {
    if (bytes > sl.in_bar_bytes) {
      if (sl.in_bar) {
        HIPCHK(h, hipFree(sl.in_bar));
        sl.in_bar = nullptr;
        sl.in_bar_bytes = 0;
      }
      const size_t want = std::max(bytes * 2, (size_t)1 << 20);
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, want, hipDeviceMallocFinegrained) == hipSuccess &&
          (h->bar_probed || raftq_detail::host_can_write(p, want))) {
        h->bar_probed = true;
        sl.in_bar = p;
        sl.in_bar_bytes = want;
      } else {
        if (p) (void)hipFree(p);
        (void)hipGetLastError();
        h->bar_staging = false;
      }
    }
    if (sl.in_bar) {
      *out = sl.in_bar;
      return RAFTQ_OK;
    }
  }
  if (int rc = ensure_slot_staging(h, sl, n, raw_bytes)) return rc;
  *out = sl.in_h;
  return RAFTQ_OK;
}

// raftq_load_node / raftq_read_node: the ABI speaks one flat array per field, the device keeps one record per group.  A
// slab of groups at a time goes through a device temporary and a kernel that writes / reads the field of every record.
int node_fields(raftq_t* h, bool put, uint64_t* term, uint32_t* vote, uint32_t* lead, uint64_t* last_index, uint64_t* last_term) {
  void* host[5] = {term, vote, lead, last_index, last_term};
  const size_t width[5] = {8, 4, 4, 8, 8};
  if (!term && !vote && !lead && !last_index && !last_term) return RAFTQ_OK;
  const uint64_t slab = std::min<uint64_t>(h->G, 1ull << 24);
  void* tmp = nullptr;
  HIPCHK(h, hipMalloc(&tmp, slab * 8));
  int rc = RAFTQ_OK;
  for (int f = 0; f < 5 && rc == RAFTQ_OK; ++f) {
    if (!host[f]) continue;
    for (uint64_t g0 = 0; g0 < h->G && rc == RAFTQ_OK; g0 += slab) {
      const uint64_t cnt = std::min(slab, h->G - g0);
      uint8_t* hp = (uint8_t*)host[f] + g0 * width[f];
      hipError_t e = hipSuccess;
      if (put) e = hipMemcpyAsync(tmp, hp, cnt * width[f], hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) {
        if (put) hipLaunchKernelGGL(node_field_kernel<true>, dim3((unsigned)((cnt + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, (NodeRec*)h->node_rec, g0, cnt, f, tmp);
        else hipLaunchKernelGGL(node_field_kernel<false>, dim3((unsigned)((cnt + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, (NodeRec*)h->node_rec, g0, cnt, f, tmp);
        e = hipGetLastError();
      }
      if (e == hipSuccess && !put) e = hipMemcpyAsync(hp, tmp, cnt * width[f], hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);  // the temporary is reused by the next slab / field
      if (e != hipSuccess) rc = fail(h, RAFTQ_EHIP, std::string("raftq node fields: ") + hipGetErrorString(e));
    }
  }
  (void)hipFree(tmp);
  return rc;
}

}  // namespace

void raftq_detail::free_node_state(raftq_t* h) {
  (void)hipFree(h->node_rec);
  h->node_rec = nullptr;
  (void)hipFree(h->step_stall);
  for (auto& sl : h->step_slot) {
    if (sl.busy && sl.copy_pending && h->stream) (void)hipStreamSynchronize(h->stream);  // its kernels; the copy never ran
    else if (sl.ev_out && sl.busy) (void)hipEventSynchronize(sl.ev_out);
    (void)hipFree(sl.dev);
    if (sl.in_h) (void)hipHostFree(sl.in_h);
    if (sl.in_bar) (void)hipFree(sl.in_bar);
    if (sl.out_h) (void)hipHostFree(sl.out_h);
    if (sl.w_pin) (void)hipHostFree(sl.w_pin);
    if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
    if (sl.ev_comp) (void)hipEventDestroy(sl.ev_comp);
    if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
  }
  for (auto& q : h->ld_nowait) {
    if (q.ev) (void)hipEventDestroy(q.ev);
    if (q.host) (void)hipHostFree(q.host);
  }
  if (h->step_s_in) (void)hipStreamDestroy(h->step_s_in);
  if (h->step_s_out) (void)hipStreamDestroy(h->step_s_out);
}

extern "C" {

int raftq_set_self(raftq_t* h, uint32_t self_peer) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (self_peer >= h->N) return fail(h, RAFTQ_EINVAL, "raftq_set_self: self_peer out of range");
  h->self_peer = self_peer;
  return RAFTQ_OK;
}

int raftq_load_node(raftq_t* h, const uint64_t* term, const uint32_t* vote, const uint32_t* lead,
                    const uint64_t* last_index, const uint64_t* last_term) {
  if (int rc = raftq_detail::use_device_idle(h, "raftq_load_node")) return rc;
  if (int rc = ensure_node_state(h)) return rc;
  for (uint64_t g = 0; g < h->G; ++g)
    if ((vote && vote[g] > h->N) || (lead && lead[g] > h->N))
      return fail(h, RAFTQ_EINVAL, "raftq_load_node: vote / lead must be 0 (None) or a peer slot + 1");
  if (int rc = node_fields(h, true, const_cast<uint64_t*>(term), const_cast<uint32_t*>(vote), const_cast<uint32_t*>(lead),
                           const_cast<uint64_t*>(last_index), const_cast<uint64_t*>(last_term)))
    return rc;
  return RAFTQ_OK;
}

int raftq_read_node(raftq_t* h, uint64_t* term, uint32_t* vote, uint32_t* lead, uint64_t* last_index,
                    uint64_t* last_term, uint64_t* first_idx_cur_term) {
  if (int rc = raftq_detail::use_device_idle(h, "raftq_read_node")) return rc;
  if (int rc = ensure_node_state(h)) return rc;
  if (int rc = node_fields(h, false, term, vote, lead, last_index, last_term)) return rc;
  if (first_idx_cur_term)
    HIPCHK(h, hipMemcpyAsync(first_idx_cur_term, h->first_idx, h->G * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

int raftq_step_stage(raftq_t* h, uint64_t n, raftq_msg_t** msgs) {
  if (int rc = use_device(h)) return rc;
  if (!msgs) return fail(h, RAFTQ_EINVAL, "raftq_step_stage: null argument");
  const int slot_no = (int)(h->step_submitted % raftq::kStepSlots);
  raftq::StepSlot& sl = h->step_slot[slot_no];  // the slot the next submit will use
  if (sl.busy) return fail(h, RAFTQ_ESTATE, "raftq_step_stage: three batches already in flight; collect one first");
  // a wire batch decoded in place in this slot has its stream in the buffer the producer is about to overwrite
  if (h->step_last_slot == slot_no && sl.w_off_in_place) h->step_last_slot = -1;
  void* p = nullptr;
  if (int rc = ensure_slot_bar(h, sl, n, &p)) return rc;
  *msgs = (raftq_msg_t*)p;
  return RAFTQ_OK;
}

int raftq_step_results(raftq_t* h, const raftq_step_out_t** out, uint64_t* n) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!out || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_results: null argument");
  if (h->step_last_out && h->step_last_rec != sizeof(StepOutRec))
    return fail(h, RAFTQ_ESTATE, "raftq_step_results: the last batch has compact records (raftq_step_results_c)");
  *out = (const raftq_step_out_t*)h->step_last_out;
  *n = h->step_last_n;
  return RAFTQ_OK;
}

int raftq_step_results_c(raftq_t* h, const raftq_step_out_c_t** out, uint64_t* n) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!out || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_results_c: null argument");
  if (h->step_last_out && h->step_last_rec != sizeof(StepOutC))
    return fail(h, RAFTQ_ESTATE, "raftq_step_results_c: the last batch has full records (raftq_step_results)");
  *out = (const raftq_step_out_c_t*)h->step_last_out;
  *n = h->step_last_n;
  return RAFTQ_OK;
}

int raftq_step_results_s(raftq_t* h, const raftq_step_out_s_t** out, uint64_t* n) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!out || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_results_s: null argument");
  if (h->step_last_out && h->step_last_rec != sizeof(raftqk::StepOutS))
    return fail(h, RAFTQ_ESTATE, "raftq_step_results_s: the last batch has records of another format (raftq_step_set_compact)");
  *out = (const raftq_step_out_s_t*)h->step_last_out;
  *n = h->step_last_n;
  return RAFTQ_OK;
}

int raftq_step_set_msg_flags(raftq_t* h, int on) {
  if (int rc = raftq_detail::use_device_idle(h, "raftq_step_set_msg_flags")) return rc;
  h->step_msg_flags = on != 0;
  return RAFTQ_OK;
}

int raftq_step_set_compact(raftq_t* h, int on) {
  if (int rc = raftq_detail::use_device_idle(h, "raftq_step_set_compact")) return rc;
  if (on < 0 || on > 2) return fail(h, RAFTQ_EINVAL, "raftq_step_set_compact: 0 (64-byte records), 1 (40-byte) or 2 (32-byte)");
  h->step_compact = (uint8_t)on;
  h->step_last_out = nullptr;
  h->step_last_n = 0;
  return RAFTQ_OK;
}

// One batch into the pipeline.  wire == nullptr: n raftq_msg_t records at msgs.  Otherwise the
// records are decoded on the device from n rafthttp stream frames (raftq_wire.h).
struct WireSrc {
  const void* stream;
  uint64_t nbytes;
  const uint64_t* frame_off;
  // raftq_step_frames: the arrays stay where they are (page-locked), the streaming decoder reads them there, hands the
  // decoded records and entry headers to the caller and keeps a copy of the records in the slot's scratch for Step
  bool frames = false;
  void* msgs_h = nullptr;
  void* ents_h = nullptr;
  uint64_t ents_cap = 0;
  int tail_appends = 0;
};

// key -> stable radix sort -> walk, on the handle's stream (the path that takes runs of any length)
static int enqueue_sorted_walk(raftq_t* h, const Scratch& s, uint64_t n, int end_bit, uint8_t recs, void* outs) {
  if (int rc = ensure_mirror(h)) return rc;
  unsigned int* bad = (unsigned int*)(s.n_heads + 1);
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
  hipLaunchKernelGGL(step_keys_kernel, grid, dim3(kBlock), 0, h->stream, (const MsgRec*)s.msgs, s.keys_in, s.order_in, n,
                     h->G, h->N, bad, h->step_msg_flags, recs, outs, h->step_compact);
  HIPCHK(h, hipGetLastError());
  int in_b = 0;  // which pair of buffers the sorted (group, batch position) pairs end up in
  HIPCHK(h, raftqk::radix_sort_pairs(h->stream, s.sort_scratch, s.keys_in, s.order_in, s.keys_out, s.order_out, n, end_bit, &in_b));
  hipLaunchKernelGGL(step_kernel, grid, dim3(kBlock), 0, h->stream, node_arrays(h, recs), (const MsgRec*)s.msgs,
                     (const uint64_t*)(in_b ? s.keys_out : s.keys_in), (const uint32_t*)(in_b ? s.order_out : s.order_in), outs,
                     h->step_compact, n, s.n_heads, (const unsigned int*)bad);
  HIPCHK(h, hipGetLastError());
  return RAFTQ_OK;
}

// link -> walk (and empty) the per-group lists: two launches, no sort (raftq_step_kernels.hpp 2b / 3b)
static unsigned d2h_blocks(uint64_t quads) { return (unsigned)std::min<uint64_t>(64, (quads + kBlock - 1) / kBlock); }

// `carry`: a slot whose result copy is still pending rides in this batch's walk kernel (nullptr: nothing to carry)
static int enqueue_list_walk(raftq_t* h, const Scratch& s, uint64_t n, uint8_t recs, raftq::StepSlot* carry) {
  if (int rc = ensure_mirror(h)) return rc;
  unsigned int* bad = (unsigned int*)(s.n_heads + 1);
  unsigned int* skipped = bad + 1;  // the last word of the 16-byte tail behind the result records
  const unsigned blocks = (unsigned)((n + kBlock - 1) / kBlock);
  // the carried copy is split over this batch's two kernels roughly as their own durations are (link ~ 1/5 of the chain)
  raftqk::CopyRide in_link{nullptr, nullptr, 0, 0, 0, 0}, in_walk = in_link;
  if (carry) {
    const uint64_t quads = carry->out_quads, cut = quads * (uint64_t)h->step_link_share / 100;
    in_link = {(const u64x2*)carry->outs_d, (u64x2*)carry->out_d, 0, cut, quads - 1, cut ? d2h_blocks(cut) : 0u};
    in_walk = {(const u64x2*)carry->outs_d, (u64x2*)carry->out_d, cut, quads, quads - 1, d2h_blocks(quads - cut)};
  }
  hipLaunchKernelGGL(step_link_kernel, dim3(blocks + in_link.blocks), dim3(kBlock), 0, h->stream, (const MsgRec*)s.msgs, n, h->G,
                     h->N, h->step_msg_flags, recs, (NodeRec*)h->node_rec, s.next, bad, h->step_stall, s.outs, h->step_compact, in_link);
  hipLaunchKernelGGL(step_lists_kernel, dim3(blocks + in_walk.blocks), dim3(kBlock), 0, h->stream, node_arrays(h, recs),
                     (const MsgRec*)s.msgs, s.outs, h->step_compact, n, h->G, (const uint32_t*)s.next, s.n_heads,
                     skipped, (const unsigned int*)bad, (const unsigned int*)h->step_stall, in_walk);
  HIPCHK(h, hipGetLastError());
  if (carry) {
    HIPCHK(h, hipEventRecord(carry->ev_out, h->stream));
    carry->copy_pending = false;
  }
  return RAFTQ_OK;
}

// the slot's result records (and tail) -> its pinned host buffer, as a kernel of its own
static int enqueue_result_copy(raftq_t* h, raftq::StepSlot& sl, hipStream_t st) {
  hipLaunchKernelGGL(step_d2h_kernel, dim3(d2h_blocks(sl.out_quads)), dim3(kBlock), 0, st, (const u64x2*)sl.outs_d, (u64x2*)sl.out_d,
                     sl.out_quads, true);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(sl.ev_out, st));
  sl.copy_pending = false;
  return RAFTQ_OK;
}

// rec_bytes: sizeof(raftq_msg_t), or sizeof(raftq_msg40_t) for packed records (widened on the device)
static int submit_impl(raftq_t* h, const void* msgs, uint64_t n, const WireSrc* wire, const char* who,
                       size_t rec_bytes = sizeof(raftq_msg_t)) {
  const bool packed = rec_bytes != sizeof(raftq_msg_t);
  if (int rc = use_device(h)) return rc;
  if (n == 0 || (!wire && !msgs)) return fail(h, RAFTQ_EINVAL, std::string(who) + ": empty batch");
  if (n > 0x7ffffffeull) return fail(h, RAFTQ_EINVAL, std::string(who) + ": batch too large (2^31 - 2 messages at most)");
  if (wire && ((!wire->stream && wire->nbytes) || !wire->frame_off))
    return fail(h, RAFTQ_EINVAL, std::string(who) + ": null argument");
  const int slot_no = (int)(h->step_submitted % raftq::kStepSlots);
  raftq::StepSlot& sl = h->step_slot[slot_no];
  if (sl.busy) return fail(h, RAFTQ_ESTATE, std::string(who) + ": three batches already in flight; collect one first");
  if (int rc = ensure_node_state(h)) return rc;
  int end_bit = 1;
  while (end_bit < 64 && (h->G >> end_bit) != 0) ++end_bit;
  size_t in_bytes;
  bool staged_in_device = false;
  const void* device_src = nullptr;
  // raftq_step_stage_wire's arrays, filled in place: in device memory the decoder reads them where they lie; in pinned
  // host memory two DMAs take them (the boundaries and the stream are not adjacent there)
  bool wire_staged = false;
  const bool frames = wire && wire->frames;
  const uint8_t recs = frames ? raftqk::kRecsFrames : wire ? raftqk::kRecsWire : raftqk::kRecsCaller;
  if (frames) {
    if (!h->step_msg_flags)
      return fail(h, RAFTQ_ESTATE, std::string(who) + ": the handle has not opted in to RAFTQ_MSGF_* (raftq_step_set_msg_flags)");
    in_bytes = 0;  // nothing is staged: the decoder reads the caller's arrays where they lie
  } else if (wire) {
    in_bytes = (size_t)(n + 1) * 8 + (size_t)wire->nbytes;
    const uint8_t* fo = (const uint8_t*)wire->frame_off;
    const uint8_t* sb = (const uint8_t*)wire->stream;
    auto inside = [&](const void* base, size_t cap) {
      const uint8_t* b = (const uint8_t*)base;
      return b && fo == b && sb == b + sl.w_stage_stream_off && sl.w_stage_stream_off >= (size_t)(n + 1) * 8 &&
             sl.w_stage_stream_off + (size_t)wire->nbytes <= cap;
    };
    if (inside(sl.in_bar, sl.in_bar_bytes)) {
      wire_staged = staged_in_device = true;
      device_src = sl.in_bar;
#if defined(__x86_64__)
      __builtin_ia32_sfence();
#endif
    } else if (inside(sl.in_h, sl.in_bytes)) {
      wire_staged = true;
    } else {
      // staging = [frame offsets][stream bytes]: one DMA moves both
      if (int rc = ensure_slot_staging(h, sl, n, in_bytes)) return rc;
      std::memcpy(sl.in_h, wire->frame_off, (size_t)(n + 1) * 8);
      if (wire->nbytes) std::memcpy((uint8_t*)sl.in_h + (size_t)(n + 1) * 8, wire->stream, (size_t)wire->nbytes);
    }
  } else {
    in_bytes = (size_t)n * rec_bytes;
    // unless the caller filled this slot's staging area in place (raftq_step_stage), copy into the pinned one
    // (either slot's device staging counts: a caller that keeps filling the buffer raftq_step_stage gave it for an
    // earlier batch must not be served by a host memcpy that READS device memory over the BAR -- 43 ms for 4 MB)
    for (const raftq::StepSlot& t : h->step_slot)
      if (t.in_bar && (const uint8_t*)msgs >= (const uint8_t*)t.in_bar &&
          (const uint8_t*)msgs + in_bytes <= (const uint8_t*)t.in_bar + t.in_bar_bytes)
        device_src = msgs;
    if (device_src) {
      staged_in_device = true;
#if defined(__x86_64__)
      __builtin_ia32_sfence();  // the producer's write-combined stores into device memory, before the doorbell
#endif
    } else if ((const void*)msgs != sl.in_h) {
      if (int rc = ensure_slot_staging(h, sl, n)) return rc;
      std::memcpy(sl.in_h, msgs, in_bytes);
    } else if (in_bytes > sl.in_bytes) {
      return fail(h, RAFTQ_EINVAL, std::string(who) + ": more messages than were staged");
    }
  }
  if (h->step_last_slot == slot_no) h->step_last_slot = -1;  // that batch's decoded records are about to be overwritten
  sl.wire = false;
  Scratch s;
  if (int rc = ensure_slot(h, sl, n, end_bit, &s, wire != nullptr && !frames, wire && !frames ? wire->nbytes : 0)) return rc;
  // Step moves the live commit index: a what-if (NO_ADOPT) sweep's shadow values are no longer what
  // raftq_read_committed should hand out
  h->last_flags &= ~RAFTQ_SWEEP_NO_ADOPT;
  // copy-in stream -> compute stream (the handle's: state changes stay ordered with every other
  // call on the handle) -> copy-out stream, chained by events
  const int mode = h->step_stream_mode;
  hipStream_t s_in = mode == 1 ? h->stream : h->step_s_in;
  hipStream_t s_out = (mode == 1 || mode == 2) ? h->stream : mode == 4 ? h->step_s_in : h->step_s_out;
  // in: one DMA over PCIe from the pinned staging.  A batch the producer wrote straight into device memory needs neither
  // a DMA nor a copy: the kernels read it where it lies (the slot's staging stays untouched until the slot's next batch,
  // which is after this one's collect -- the replay path reads it there as well).  Round 2 first kept a device-to-device
  // copy into the slot's scratch; with the result copy inside the walk kernel that 3 us blit sat on the critical path,
  // held back like every other kernel while a kernel writes to host memory (rocprof: 2.4 .. 75 us).
  sl.msgs_in_place = nullptr;
  sl.w_off_in_place = sl.w_stream_in_place = nullptr;
  const void* packed_src = s.msgs40;
  const bool own_staging = staged_in_device && (const uint8_t*)device_src >= (const uint8_t*)sl.in_bar &&
                           (const uint8_t*)device_src + in_bytes <= (const uint8_t*)sl.in_bar + sl.in_bar_bytes;
  if (frames) {
    // (enqueued below, on the handle's stream, in front of Step's kernels)
  } else if (wire && wire_staged && staged_in_device) {
    s.w_off = (uint64_t*)const_cast<void*>(device_src);
    s.w_stream = (uint8_t*)const_cast<void*>(device_src) + sl.w_stage_stream_off;
    sl.w_off_in_place = s.w_off;
    sl.w_stream_in_place = s.w_stream;
  } else if (wire && wire_staged) {
    HIPCHK(h, hipMemcpyAsync(s.w_off, sl.in_h, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s_in));
    if (wire->nbytes)
      HIPCHK(h, hipMemcpyAsync(s.w_stream, (const uint8_t*)sl.in_h + sl.w_stage_stream_off, (size_t)wire->nbytes,
                               hipMemcpyHostToDevice, s_in));
    if (s_in != h->stream) {
      HIPCHK(h, hipEventRecord(sl.ev_in, s_in));
      HIPCHK(h, hipStreamWaitEvent(h->stream, sl.ev_in, 0));
    }
  } else if (own_staging) {
    if (packed) packed_src = device_src;
    else {
      s.msgs = (MsgRec*)const_cast<void*>(device_src);
      sl.msgs_in_place = device_src;
    }
  } else if (staged_in_device) {
    // another slot's staging (a caller that kept an earlier batch's pointer): that buffer may be handed out again while
    // this batch is in flight, so the records are copied into this slot's scratch, device to device
    void* in_dst = packed ? s.msgs40 : (void*)s.msgs;
    HIPCHK(h, hipMemcpyAsync(in_dst, device_src, in_bytes, hipMemcpyDeviceToDevice, s_in));
    if (s_in != h->stream) {
      HIPCHK(h, hipEventRecord(sl.ev_in, s_in));
      HIPCHK(h, hipStreamWaitEvent(h->stream, sl.ev_in, 0));
    }
  } else {
    void* in_dst = wire ? (void*)s.w_off : packed ? s.msgs40 : (void*)s.msgs;
    HIPCHK(h, hipMemcpyAsync(in_dst, sl.in_h, in_bytes, hipMemcpyHostToDevice, s_in));
    if (s_in != h->stream) {
      HIPCHK(h, hipEventRecord(sl.ev_in, s_in));
      HIPCHK(h, hipStreamWaitEvent(h->stream, sl.ev_in, 0));
    }
  }
  // touched count + bad + skipped: the default result copy leaves them zeroed behind it (step_d2h_kernel zero_tail)
  const uint32_t rec = result_rec_bytes(h);
  // (a frames batch: the decoder that heads the chain zeroes them)
  if (!frames && (!sl.tail_zeroed || sl.tail_n != n || sl.tail_rec != rec || sl.dev != sl.tail_dev)) hipLaunchKernelGGL(step_reset_kernel, dim3(1), dim3(64), 0, h->stream, s.n_heads);
  sl.tail_zeroed = false;
  if (packed) {
    hipLaunchKernelGGL(raftqk::step_unpack40_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream,
                       (const raftqk::Msg40Rec*)packed_src, s.msgs, n);
    HIPCHK(h, hipGetLastError());
  }
  if (frames) {
    // one kernel: readers pull boundaries + stream, workers parse, check, flag (RAFTQ_MSGF_*) and push records + entry headers
    // to the caller's arrays and the records once more into this slot's scratch
    if (int rc = raftq_detail::wire_frames_enqueue(h, wire->stream, wire->nbytes, wire->frame_off, n, wire->msgs_h, wire->ents_h,
                                                   wire->ents_cap, s.msgs, wire->tail_appends, s.n_heads))
      return rc;
  } else if (wire) {
    // frames -> the batch's message records, in HBM: the 64-byte records never cross PCIe.  (s.w_bad only counts
    // malformed frames for raftq_wire_decode's callers; here a malformed frame fails the batch through its flag byte.)
    const dim3 grid1((unsigned)((n + 1 + kBlock - 1) / kBlock));
    hipLaunchKernelGGL(wire_dec_kernel, grid1, dim3(kBlock), 0, h->stream, (const uint8_t*)s.w_stream, wire->nbytes,
                       (const uint64_t*)s.w_off, n, (WireMsg*)s.msgs, s.w_cnt, s.w_bad);
    HIPCHK(h, hipGetLastError());
    // Step reads only the headers.  The entry headers (count -> scan -> second parse) are produced when
    // raftq_step_wire_msgs / _entries first asks for them: fetch_wire below.
  }
  // traffic that keeps producing long runs (a few very hot groups) would stall and replay every batch: after two
  // stalls in a row the next 16 batches go straight to the sorted walk, then the list walk is tried again
  if (h->step_sorted_left) --h->step_sorted_left;
  const bool lists = h->step_walk_mode == 1 && h->step_sorted_left == 0;
  // Result copies.  With everything on the handle's stream (the default topology) a batch's copy is DEFERRED: it rides in
  // the walk kernel of the batch submitted behind it, or is launched by its own collect if none is.  The batch submitted
  // just before this one may be waiting for exactly that.
  // Only for batches that arrived in device memory (raftq_step_stage behind a large BAR): a batch that needs an inbound
  // DMA is better off with its own copy kernel -- the DMA of the batch after next cannot start while a kernel is writing
  // to host memory either, and the fused kernel writes for longer (measured: caller-owned arrays 157 us per batch with
  // their own copy kernels, 189 us deferred; staged batches 101 -> 77 us; profiles/r02/step_deferred_copy_ab.txt).
  const bool defer = s_out == h->stream && h->step_defer_copy && staged_in_device;
  raftq::StepSlot* carry = nullptr;
  if (h->step_submitted > h->step_collected) {
    raftq::StepSlot& prev = h->step_slot[(h->step_submitted - 1) % raftq::kStepSlots];
    if (prev.busy && prev.copy_pending) carry = &prev;
  }
  if (lists) {
    if (int rc = enqueue_list_walk(h, s, n, recs, carry)) return rc;
  } else {
    if (carry)
      if (int rc = enqueue_result_copy(h, *carry, h->stream)) return rc;
    if (int rc = enqueue_sorted_walk(h, s, n, end_bit, recs, s.outs)) return rc;
  }
  sl.outs_d = s.outs;
  sl.out_quads = tail_off(n, rec) / 16 + 1;  // records + the 16-byte tail
  if (defer) {
    sl.copy_pending = true;
  } else {
    if (s_out != h->stream) {
      HIPCHK(h, hipEventRecord(sl.ev_comp, h->stream));
      HIPCHK(h, hipStreamWaitEvent(s_out, sl.ev_comp, 0));
    }
    if (int rc = enqueue_result_copy(h, sl, s_out)) return rc;
  }
  sl.tail_zeroed = true;  // (once the copy has run) holds for the next batch of the same size and format in the same scratch
  sl.tail_n = n;
  sl.tail_rec = rec;
  sl.tail_dev = sl.dev;
  sl.n = n;
  sl.busy = true;
  sl.lists = lists;
  sl.replayed = false;
  sl.end_bit = end_bit;
  sl.rec = rec;
  sl.w_nbytes = wire ? wire->nbytes : 0;
  sl.wire = wire != nullptr && !frames;
  sl.recs = recs;
  sl.w_msgs_d = s.msgs;
  sl.w_ents_d = s.w_ents;
  sl.w_ent_total_d = s.w_base + n;
  sl.w_ents_cap = s.w_ents_cap;
  sl.w_msgs_fetched = sl.w_ents_fetched = false;
  sl.w_n_ents = 0;
  h->step_submitted++;
  return RAFTQ_OK;
}

int raftq_step_submit(raftq_t* h, const raftq_msg_t* msgs, uint64_t n) {
  return submit_impl(h, msgs, n, nullptr, "raftq_step_submit");
}

int raftq_step_stage_packed(raftq_t* h, uint64_t n, raftq_msg40_t** msgs) {
  static_assert(sizeof(raftq_msg40_t) == sizeof(raftqk::Msg40Rec), "ABI struct mismatch");
  return raftq_step_stage(h, n, (raftq_msg_t**)msgs);  // a staging area for n 64-byte records holds n packed ones
}

int raftq_step_submit_packed(raftq_t* h, const raftq_msg40_t* msgs, uint64_t n) {
  if (h && h->G > 0x100000000ull) return fail(h, RAFTQ_EINVAL, "raftq_step_submit_packed: group ids of this handle do not fit 32 bits");
  return submit_impl(h, msgs, n, nullptr, "raftq_step_submit_packed", sizeof(raftq_msg40_t));
}

int raftq_step_stage_wire(raftq_t* h, uint64_t n_cap, uint64_t nbytes_cap, uint64_t** frame_off, void** stream) {
  if (int rc = use_device(h)) return rc;
  if (!frame_off || !stream) return fail(h, RAFTQ_EINVAL, "raftq_step_stage_wire: null argument");
  const int slot_no = (int)(h->step_submitted % raftq::kStepSlots);
  raftq::StepSlot& sl = h->step_slot[slot_no];  // the slot the next submit will use
  if (sl.busy) return fail(h, RAFTQ_ESTATE, "raftq_step_stage_wire: three batches already in flight; collect one first");
  // the slot's previous batch may still be the one raftq_step_wire_msgs / _entries would parse entries for -- from the
  // very bytes the producer is about to overwrite
  if (h->step_last_slot == slot_no) h->step_last_slot = -1;
  const size_t off_bytes = align256((size_t)(n_cap + 1) * 8);
  void* p = nullptr;
  if (int rc = ensure_slot_bar(h, sl, n_cap, &p, off_bytes + (size_t)nbytes_cap + 16)) return rc;
  sl.w_stage_stream_off = off_bytes;
  *frame_off = (uint64_t*)p;
  *stream = (uint8_t*)p + off_bytes;
  return RAFTQ_OK;
}

int raftq_step_submit_wire(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n) {
  const WireSrc w = {stream, nbytes, frame_off};
  return submit_impl(h, nullptr, n, &w, "raftq_step_submit_wire");
}

// the decoded records of the last collected batch, fetched from its slot on demand
static int fetch_wire(raftq_t* h, bool want_ents, const char* who, raftq::StepSlot** out) {
  if (int rc = use_device(h)) return rc;
  if (h->step_last_slot < 0 || !h->step_slot[h->step_last_slot].wire)
    return fail(h, RAFTQ_ESTATE, std::string(who) + ": the last collected batch was not submitted from the wire "
                                                      "(or its slot has been reused)");
  raftq::StepSlot& sl = h->step_slot[h->step_last_slot];
  const size_t msg_bytes = (size_t)sl.n * sizeof(WireMsg);
  hipStream_t st = h->step_s_out;  // not behind whatever batch is in flight on the handle's stream
  if (!sl.w_msgs_fetched) {
    // first request for this batch: the decoder's second pass (entry counts -> exclusive scan -> entry headers in
    // message order, ent_first of every message), kept out of the Step chain because Step does not need it
    Scratch s;
    if (int rc = ensure_slot(h, sl, sl.n, sl.end_bit, &s, true, sl.w_nbytes)) return rc;
    if (sl.w_off_in_place) {  // the batch was decoded where the producer wrote it; so are its entries
      s.w_off = (uint64_t*)const_cast<void*>(sl.w_off_in_place);
      s.w_stream = (uint8_t*)const_cast<void*>(sl.w_stream_in_place);
    }
    HIPCHK(h, exclusive_sum_u64((const uint64_t*)s.w_cnt, s.w_base, sl.n + 1, (uint64_t*)s.w_scan, st));
    hipLaunchKernelGGL(wire_dec_ents_kernel, dim3((unsigned)((sl.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       (const uint8_t*)s.w_stream, sl.w_nbytes, (const uint64_t*)s.w_off, sl.n, (WireMsg*)s.msgs,
                       (const uint64_t*)s.w_base, s.w_ents, s.w_ents_cap);
    HIPCHK(h, hipGetLastError());
    // learn the entry count, size the pinned block for both arrays (so a pointer handed out for the messages
    // stays valid when the entries are asked for later)
    uint64_t total = 0;
    HIPCHK(h, hipMemcpyAsync(&total, sl.w_ent_total_d, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (total > sl.w_ents_cap) total = sl.w_ents_cap;  // cannot happen: the cap is the worst case
    const size_t need = align256(msg_bytes) + (size_t)total * sizeof(WireEnt);
    if (need > sl.w_pin_bytes) {
      if (sl.w_pin) HIPCHK(h, hipHostFree(sl.w_pin));
      sl.w_pin = nullptr;
      sl.w_pin_bytes = 0;
      const size_t want = std::max(need * 2, (size_t)1 << 20);
      HIPCHK(h, hipHostMalloc(&sl.w_pin, want, hipHostMallocDefault));
      sl.w_pin_bytes = want;
    }
    HIPCHK(h, hipMemcpyAsync(sl.w_pin, sl.w_msgs_d, msg_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    sl.w_n_ents = total;
    sl.w_msgs_fetched = true;
  }
  if (want_ents && !sl.w_ents_fetched) {
    if (sl.w_n_ents) {
      HIPCHK(h, hipMemcpyAsync((uint8_t*)sl.w_pin + align256(msg_bytes), sl.w_ents_d,
                               (size_t)sl.w_n_ents * sizeof(WireEnt), hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
    }
    sl.w_ents_fetched = true;
  }
  *out = &sl;
  return RAFTQ_OK;
}

int raftq_step_wire_msgs(raftq_t* h, const raftq_wire_msg_t** msgs, uint64_t* n) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!msgs || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_wire_msgs: null argument");
  raftq::StepSlot* sl = nullptr;
  if (int rc = fetch_wire(h, false, "raftq_step_wire_msgs", &sl)) return rc;
  *msgs = (const raftq_wire_msg_t*)sl->w_pin;
  *n = sl->n;
  return RAFTQ_OK;
}

int raftq_step_wire_entries(raftq_t* h, const raftq_wire_ent_t** ents, uint64_t* n_ents) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!ents || !n_ents) return fail(h, RAFTQ_EINVAL, "raftq_step_wire_entries: null argument");
  raftq::StepSlot* sl = nullptr;
  if (int rc = fetch_wire(h, true, "raftq_step_wire_entries", &sl)) return rc;
  *ents = (const raftq_wire_ent_t*)((const uint8_t*)sl->w_pin + align256((size_t)sl->n * sizeof(WireMsg)));
  *n_ents = sl->w_n_ents;
  return RAFTQ_OK;
}

// batch `first` (already taken out of flight by its collect) reported `skipped`; the batches submitted behind it are
// still in flight and were skipped as well: all of them go through the sorted walk again, in submission order
static int replay_stalled(raftq_t* h, uint64_t first) {
  for (uint64_t b = first; b < h->step_submitted; ++b) {
    raftq::StepSlot& sl = h->step_slot[b % raftq::kStepSlots];
    if (b != first) {
      if (!sl.busy) break;
      // its copy, if one was enqueued (it rode in the batch behind it), must be over before the replay's own lands in
      // the same host buffer; a pending one is simply replaced by the replay's
      if (!sl.copy_pending) HIPCHK(h, hipEventSynchronize(sl.ev_out));
    }
    Scratch s;
    if (int rc = ensure_slot(h, sl, sl.n, sl.end_bit, &s, sl.wire, sl.w_nbytes)) return rc;
    if (sl.msgs_in_place) s.msgs = (MsgRec*)const_cast<void*>(sl.msgs_in_place);
    hipLaunchKernelGGL(step_reset_kernel, dim3(1), dim3(64), 0, h->stream, s.n_heads);
    if (int rc = enqueue_sorted_walk(h, s, sl.n, sl.end_bit, sl.recs, s.outs)) return rc;
    const uint64_t out_quads = tail_off(sl.n, sl.rec) / 16 + 1;
    hipLaunchKernelGGL(step_d2h_kernel, dim3(d2h_blocks(out_quads)), dim3(kBlock), 0, h->stream, (const u64x2*)s.outs,
                       (u64x2*)sl.out_d, out_quads);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(sl.ev_out, h->stream));
    sl.copy_pending = false;
    sl.tail_zeroed = false;
    sl.replayed = true;
    h->step_replays++;
  }
  HIPCHK(h, hipMemsetAsync(h->step_stall, 0, 4, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

int raftq_step_collect(raftq_t* h, raftq_step_out_t* out, raftq_step_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) counts->n_msgs = counts->n_groups_touched = 0;
  if (h->step_collected == h->step_submitted) return fail(h, RAFTQ_ESTATE, "raftq_step_collect: nothing in flight");
  raftq::StepSlot& sl = h->step_slot[h->step_collected % raftq::kStepSlots];
  if (sl.copy_pending)  // nothing was submitted behind this batch: its results leave now
    if (int rc = enqueue_result_copy(h, sl, h->stream)) return rc;
  HIPCHK(h, hipEventSynchronize(sl.ev_out));
  sl.busy = false;
  h->step_last_slot = (int)(h->step_collected % raftq::kStepSlots);
  h->step_collected++;
  const uint64_t n = sl.n;
  const uint8_t* tail = (const uint8_t*)sl.out_h + tail_off(n, sl.rec);
  unsigned long long heads;
  unsigned int bad_h, skipped_h;
  std::memcpy(&skipped_h, tail + 12, 4);
  if (skipped_h) {
    // this batch has a run longer than the list walk takes: nothing of it, nor of the batch submitted behind it,
    // was applied.  Replay them in submission order through the sorted walk, then let the list walk resume.
    if (int rc = replay_stalled(h, h->step_collected - 1)) return rc;
    if (++h->step_stalls_in_a_row >= 2) h->step_sorted_left = 16;
  } else if (sl.lists && !sl.replayed) {
    h->step_stalls_in_a_row = 0;
  }
  std::memcpy(&heads, tail, 8);
  std::memcpy(&bad_h, tail + 8, 4);
  if (bad_h) {
    h->step_last_out = nullptr;
    h->step_last_n = 0;
    return fail(h, RAFTQ_EINVAL,
                "raftq_step: a message is malformed (group or from out of range, or a type Step does not take); "
                "nothing of that batch was applied");
  }
  h->step_last_out = sl.out_h;
  h->step_last_n = n;
  h->step_last_rec = sl.rec;
  if (out) {
    if (sl.rec != sizeof(StepOutRec))
      return fail(h, RAFTQ_EINVAL, "raftq_step_collect: compact result records are read in place (raftq_step_results_c), pass out = NULL");
    std::memcpy(out, sl.out_h, (size_t)n * sizeof(StepOutRec));
  }
  if (counts) {
    counts->n_msgs = n;
    counts->n_groups_touched = heads;
  }
  return RAFTQ_OK;
}

int raftq_step_batch(raftq_t* h, const raftq_msg_t* msgs, uint64_t n, raftq_step_out_t* out,
                     raftq_step_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) counts->n_msgs = counts->n_groups_touched = 0;
  if (h->step_collected != h->step_submitted)
    return fail(h, RAFTQ_ESTATE, "raftq_step_batch: submitted batches are still in flight; collect them first");
  h->step_last_out = nullptr;
  h->step_last_n = 0;
  if (n == 0) return RAFTQ_OK;
  if (!msgs) return fail(h, RAFTQ_EINVAL, "raftq_step_batch: null messages");
  if (out && h->step_compact)
    return fail(h, RAFTQ_EINVAL, "raftq_step_batch: compact result records are read in place (raftq_step_results_c), pass out = NULL");
  if (int rc = raftq_step_submit(h, msgs, n)) return rc;
  return raftq_step_collect(h, out, counts);
}

int raftq_step_frames(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, int tail_appends,
                      raftq_wire_msg_t* msgs, raftq_wire_ent_t* ents, uint64_t ents_cap, raftq_wire_counts_t* counts) {
  if (int rc = use_device(h)) return rc;
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (h->step_collected != h->step_submitted)
    return fail(h, RAFTQ_ESTATE, "raftq_step_frames: submitted batches are still in flight; collect them first");
  h->step_last_out = nullptr;
  h->step_last_n = 0;
  if (n == 0) return RAFTQ_OK;
  if ((!stream && nbytes) || !frame_off || !msgs) return fail(h, RAFTQ_EINVAL, "raftq_step_frames: null argument");
  WireSrc w{stream, nbytes, frame_off, true, msgs, ents, ents ? ents_cap : 0, tail_appends};
  if (int rc = submit_impl(h, nullptr, n, &w, "raftq_step_frames")) return rc;
  // one wait: the result copy is the last kernel of the chain the decoder heads
  const int rc_step = raftq_step_collect(h, nullptr, nullptr);
  const int rc_dec = raftq_detail::wire_frames_finish(h, frame_off, n, ents != nullptr, ents_cap, counts);
  if (rc_dec != RAFTQ_OK) {  // a look-back gave up: the records Step read are not to be trusted, nor is what it made of them
    h->step_last_out = nullptr;
    h->step_last_n = 0;
    return rc_dec;
  }
  return rc_step;
}

// wait = false (raftq_apply_log_deltas_nowait): the same kernels from a staging area of their own, enqueued and left
static int log_deltas_impl(raftq_t* h, const raftq_log_delta_t* d, uint64_t n, uint64_t* committed_out, bool wait) {
  if (int rc = raftq_detail::use_device_idle(h, wait ? "raftq_apply_log_deltas" : "raftq_apply_log_deltas_nowait")) return rc;
  if (n == 0) return RAFTQ_OK;
  if (!d) return fail(h, RAFTQ_EINVAL, "raftq_apply_log_deltas: null argument");
  for (uint64_t i = 0; i < n; ++i)
    if (d[i].group >= h->G) return fail(h, RAFTQ_EINVAL, "a log delta is out of range; nothing applied");
  if (int rc = ensure_node_state(h)) return rc;
  if (int rc = ensure_mirror(h)) return rc;
  const size_t off_out = align256((size_t)n * sizeof(raftq_log_delta_t));
  if (n > 0xfffffffeull) return fail(h, RAFTQ_EINVAL, "raftq_apply_log_deltas: batch too large (n must fit 32 bits)");
  void *stage_h = nullptr, *stage_d = nullptr;
  hipEvent_t stage_ev = nullptr;
  if (wait) {
    if (int rc = ensure_staging(h, off_out + (size_t)n * 8)) return rc;
    stage_h = h->stage_h;
    stage_d = h->stage_d;
  } else {
    // two pinned areas taken in turn: the host may not write into one before the kernels that read it have run -- which they
    // have, two calls later, in anything but a pathological queue (the event says so)
    raftq_t::LdNowait& q = h->ld_nowait[h->ld_nowait_next++ & 1];
    if (q.ev) HIPCHK(h, hipEventSynchronize(q.ev));
    else HIPCHK(h, hipEventCreateWithFlags(&q.ev, hipEventDisableTiming));
    const size_t need = (size_t)n * sizeof(raftq_log_delta_t);
    if (need > q.bytes) {
      if (q.host) HIPCHK(h, hipHostFree(q.host));
      q.host = q.dev = nullptr;
      q.bytes = 0;
      const size_t want = std::max(need * 2, (size_t)1 << 16);
      HIPCHK(h, hipHostMalloc(&q.host, want, hipHostMallocMapped));
      HIPCHK(h, hipHostGetDevicePointer(&q.dev, q.host, 0));
      q.bytes = want;
    }
    stage_h = q.host;
    stage_d = q.dev;
    stage_ev = q.ev;
  }
  // The common batch names every group once: one round, caller order, no bookkeeping.  Whether it does is found
  // with a per-group mark (epoch << 32 | records seen this call) -- an array lookup per record, not a hash map:
  // the map this replaces was most of the call's host time at 10^4 records (raftq_node's "deltas" phase).
  std::vector<uint32_t>& round = h->ld_round;    // per record: its round
  std::vector<uint32_t>& pos_of = h->ld_pos;     // pos_of[staged position] = caller's index
  std::vector<uint64_t>& round_start = h->ld_start;  // staged position where round r begins (+ the end)
  uint32_t n_rounds = 1;
  try {
    if (h->ld_mark.size() != h->G) h->ld_mark.assign(h->G, 0);
    if (++h->ld_epoch == 0) {
      std::fill(h->ld_mark.begin(), h->ld_mark.end(), 0ull);
      h->ld_epoch = 1;
    }
    const uint64_t ep = (uint64_t)h->ld_epoch << 32;
    round.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
      uint64_t& mk = h->ld_mark[d[i].group];
      const uint32_t r = (mk >> 32) == h->ld_epoch ? (uint32_t)mk : 0;
      mk = ep | (r + 1);
      round[i] = r;
      n_rounds = std::max(n_rounds, r + 1);
    }
    if (n_rounds > 1) {
      pos_of.resize(n);
      round_start.assign((size_t)n_rounds + 1, 0);
      for (uint64_t i = 0; i < n; ++i) round_start[round[i] + 1]++;
      for (uint32_t r = 0; r < n_rounds; ++r) round_start[r + 1] += round_start[r];
    }
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_apply_log_deltas: host allocation failed");
  }
  h->last_flags &= ~RAFTQ_SWEEP_NO_ADOPT;  // as in raftq_step_submit: the live commit index moves
  raftq_log_delta_t* dst = (raftq_log_delta_t*)stage_h;
  uint64_t* out_h = (uint64_t*)((uint8_t*)stage_h + off_out);
  uint64_t* out_d = (uint64_t*)((uint8_t*)stage_d + off_out);
  auto launch = [&](uint64_t start, uint64_t m) {
    hipLaunchKernelGGL(log_deltas_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream,
                       node_arrays(h), (const LogDeltaRec*)stage_d + start, m,
                       committed_out ? out_d + start : (uint64_t*)nullptr);
  };
  if (n_rounds == 1) {
    std::memcpy(dst, d, (size_t)n * sizeof(raftq_log_delta_t));
    launch(0, n);
    HIPCHK(h, hipGetLastError());
    if (!wait) {
      HIPCHK(h, hipEventRecord(stage_ev, h->stream));
      return RAFTQ_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (committed_out) std::memcpy(committed_out, out_h, (size_t)n * 8);
    return RAFTQ_OK;
  }
  {
    // records of one group apply in order: the k-th record of a group goes into launch k.  One counting sort on
    // the round number buckets the batch in O(n) whatever the skew; stable, so caller order within a round
    std::vector<uint64_t> fill;
    try {  // a bad_alloc must not cross the extern "C" boundary
      fill.assign(round_start.begin(), round_start.end() - 1);
    } catch (...) {
      return fail(h, RAFTQ_ENOMEM, "raftq_apply_log_deltas: host allocation failed");
    }
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t pos = fill[round[i]]++;
      pos_of[pos] = (uint32_t)i;
      dst[pos] = d[i];
    }
  }
  for (uint32_t r = 0; r < n_rounds; ++r) {
    const uint64_t start = round_start[r], m = round_start[r + 1] - start;
    if (m == 0) continue;
    launch(start, m);
    HIPCHK(h, hipGetLastError());
  }
  if (!wait) {
    HIPCHK(h, hipEventRecord(stage_ev, h->stream));
    return RAFTQ_OK;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (committed_out)
    for (uint64_t k = 0; k < n; ++k) committed_out[pos_of[k]] = out_h[k];
  return RAFTQ_OK;
}

int raftq_apply_log_deltas(raftq_t* h, const raftq_log_delta_t* d, uint64_t n, uint64_t* committed_out) {
  return log_deltas_impl(h, d, n, committed_out, true);
}

int raftq_apply_log_deltas_nowait(raftq_t* h, const raftq_log_delta_t* d, uint64_t n) {
  return log_deltas_impl(h, d, n, nullptr, false);
}

}  // extern "C"
