// raftq_internal.hpp -- the handle behind raftq_t, shared by the translation units
// that implement include/raftq.h (raftq_capi.hip) and include/raftq_step.h (raftq_step.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "raftq.h"
#include "raftq_wire.h"
#include "raftq_kernels.hpp"

struct raftq {
  int device = 0;
  uint64_t G = 0, gpad = 0, ld = 0;  // groups, groups padded to the tile granule, row stride
  uint32_t N = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  struct raftq_set* in_set = nullptr;  // member of a sweep set: the set owns the stream
  uint64_t* match = nullptr;
  uint64_t* committed[2] = {nullptr, nullptr};
  int cur = 0;
  uint64_t* first_idx = nullptr;
  uint8_t* votes = nullptr;
  uint8_t* outcome = nullptr;
  uint64_t* changed_bits = nullptr;
  uint4* partials = nullptr;
  uint4* h_partials = nullptr;  // pinned
  uint64_t n_partials = 0;      // of the most recent sweep
  uint64_t max_partials = 0;
  uint64_t* offsets = nullptr;  // [max_partials + 1]; last = total
  uint64_t* h_total = nullptr;  // pinned
  // pinned, device-mapped host buffers: deltas go in, advances come out, both
  // accessed by the kernels directly over PCIe (no staging memcpy launches)
  void* stage_h = nullptr;      // delta staging (host pointer)
  void* stage_d = nullptr;      // same memory, device pointer
  size_t stage_bytes = 0;
  // The batching turn's ack buffer (raftq_stage*, raftq_cycle*, raftq_apply_[vote_]deltas).  On a device whose memory
  // the host can address (large BAR) it lives IN HBM: the producer's stores are posted PCIe writes that land in device
  // memory as the acks arrive (45 GB/s measured from one core), and the ingest kernel reads HBM instead of pulling the
  // batch over PCIe in 64-byte requests (35 GB/s, 30 of a turn's 52 kernel-us).  Fine-grained, so the GPU never serves
  // it from a stale L2 line.  The host only ever WRITES it (reads over the BAR are uncached and slow), which is why
  // calls that read results back from staging keep the pinned buffer above.  Elsewhere it is pinned host memory.
  void* ingest_h = nullptr;     // what the host writes
  void* ingest_d = nullptr;     // what the kernels read (same pointer when the buffer is device memory)
  size_t ingest_bytes = 0;
  bool ingest_in_device = false;
  bool bar_staging = false;     // decided at create: large BAR present and not disabled (RAFTQ_STAGE=host)
  bool bar_probed = false;      // such memory has been found mapped writable into this process (/proc/self/maps)
  // raftq_apply_log_deltas' host bookkeeping, kept across calls: per-group (epoch << 32 | records seen this call)
  std::vector<uint64_t> ld_mark, ld_start;
  // raftq_apply_log_deltas_nowait: two pinned staging areas taken in turn, an event each (the kernels that read it have run)
  struct LdNowait {
    void* host = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
  };
  LdNowait ld_nowait[2];
  uint32_t ld_nowait_next = 0;
  std::vector<uint32_t> ld_round, ld_pos;
  uint32_t ld_epoch = 0;
  raftqk::Advance* adv_h = nullptr;     // compacted advance list (host pointer)
  raftqk::Advance* adv_d = nullptr;
  uint64_t adv_cap = 0;
  uint64_t adv_listed = 0;     // entries of adv_h valid after the last collect / cycle
  bool adv_packed = false;     // ... in the 16-byte layout (raftq_cycle_packed)
  // RAFTQ_CYCLE_SEGMENTED: the last list lies in adv_h as one segment of seg_stride records per sweep tile, seg_h[t] of them valid
  bool adv_segmented = false;
  uint32_t* seg_h = nullptr;   // pinned [seg_cap] per-tile counts
  uint32_t* seg_hd = nullptr;  // ... as the device addresses them
  unsigned int* seg_d = nullptr;  // device copy (the flag kernel adds them up)
  uint32_t seg_cap = 0, seg_tiles = 0, seg_stride = 0, seg_one = 0;  // seg_one: the count of a contiguous list presented as one segment
  uint64_t flag_mask = ~0ull;  // which bits of the completion word are the epoch the turn's wait compares (segmented: the top half)
  unsigned int* compact_arrived = nullptr;  // (spare device word)
  uint64_t compact_epoch = 0;  // completion-flag values handed to hipStreamWriteValue64 (h_total[3])
  uint64_t compact_epoch_armed = 0;  // epoch the current turn's wait may poll for (0 = blocking wait)
  bool stream_write_ok = true; // hipStreamWriteValue64 works on this stack
  // wait_turn: consecutive waits whose flag did not land in time and waits sat out since -- per KIND of wait ([0] the batching
  // turn, [1] every other call that ends on the completion word: codecs, tick lists): a codec call under a profiler must not
  // switch the turn's wake-up off
  uint32_t flag_misses[2] = {0, 0}, flag_rested[2] = {0, 0};
  unsigned int* arrive_count = nullptr;       // RAFTQ_CYCLE_FLAG=arrive: the arrival counter (device, zero between kernels)
  uint64_t flag_fallbacks = 0;                // turns that ended in the blocking wait although a flag was armed
  uint32_t* claim = nullptr;    // u32 [N][ld] vote-slot claims, lazily allocated
  // sparse ingest: device copy of the batch (validated on the way in) and the "bad batch" epoch words
  void* delta_dev = nullptr;
  size_t delta_dev_bytes = 0;
  unsigned long long* delta_bad = nullptr;  // device: [0] match deltas, [1] vote deltas -- epoch of the last bad batch
  unsigned long long delta_epoch = 0;       // batches enqueued so far
  unsigned long long delta_check[2] = {0, 0};  // epochs whose host-visible flag (h_total[1], [2]) is still to be checked
  // batched Tick state, lazily allocated
  uint8_t* role = nullptr;      // [ld]
  uint32_t* elapsed = nullptr;  // [ld]
  uint8_t* action = nullptr;    // [ld]
  uint64_t* hup_bits = nullptr; // [gpad/64]
  uint64_t* beat_bits = nullptr; // [gpad/64]
  uint4* tick_partials = nullptr;  // [gpad/256]
  // raftq_tick_collect_lists: the two lists left in place (page-locked; 4-byte group ids; MsgBeat optionally as a bitmap)
  uint32_t* tl_h = nullptr;     // pinned: [tl_hup_cap] MsgHup ids | [tl_beat_cap] MsgBeat ids | beat bitmap (gpad / 64 words, 16-byte aligned)
  uint32_t* tl_d = nullptr;     // ... as the device addresses it
  uint64_t tl_bytes = 0, tl_hup_cap = 0, tl_beat_cap = 0, tl_beat_at = 0, tl_map_off = 0;  // tl_beat_at: index of the first MsgBeat id; tl_map_off: byte offset of the bitmap
  uint64_t tl_n_hup = 0, tl_n_beat = 0;
  unsigned tl_flags = 0;
  bool tl_valid = false;
  uint64_t* tick_offsets2 = nullptr;  // [gpad/256 + 1]: the MsgBeat offsets of raftq_tick_collect on handles of more than 16K waves
  uint32_t election_tick = 10, heartbeat_tick = 1;  // reference raft.go:154-155
  uint64_t tick_seed = 0x1000, tick_no = 0;
  bool ticked = false;
  uint64_t* d_total = nullptr;  // device alias of h_total
  bool have_terms = false;
  unsigned last_flags = 0;
  int last_gpl = 0;
  const uint64_t* last_old = nullptr;
  const uint64_t* last_new = nullptr;
  // batched Step node state (raftq_step.hip), lazily allocated
  uint32_t self_peer = 0;
  // the records' copies of the dense arrays (role, committed, first_idx, match) are what those arrays hold: false after anything but
  // Step / tail reports / proposals wrote them; the next Step-family call re-reads them first (raftq_step.hip ensure_mirror)
  bool node_mirror_fresh = false;
  uint64_t node_mirror_refreshes = 0;
  void* node_rec = nullptr;        // raftqk::NodeRec [ld]: term, vote, lead, last_index, last_term + the list words of the batch in flight
  unsigned int* step_stall = nullptr;  // device word: a batch needs the sorted path; later batches wait for the replay
  uint8_t step_compact = 0;        // result records: 0 = 64 bytes, 1 = 40, 2 = 32 (raftq_step_set_compact)
  bool step_msg_flags = false;     // raftq_msg_t._pad[1] / _resv carry RAFTQ_MSGF_* (raftq_step_set_msg_flags); padding otherwise
  int step_walk_mode = 1;          // 1 = lists (default), 0 = always the sorted walk (RAFTQ_STEP_WALK=sort)
  uint32_t step_stalls_in_a_row = 0, step_sorted_left = 0;  // back-off from the list walk under hot-group traffic
  uint64_t step_replays = 0;       // batches that went through the sorted path after a stall
  // raftq_step_batch / _submit / _collect: two batches may be in flight, each in its own slot
  // (pinned staging in, device scratch, pinned results out), pipelined over two streams (DMA in | kernels +
  // result copy): the H2D of batch k+1 overlaps the kernels and the result copy of batch k
  // Batches of the pipelined Step in flight at most.  Three, because a batch's result copy rides in the walk kernel of
  // the batch behind it (see copy_pending): while the host waits for batch k it must be able to have batch k+1 on the
  // device AND be handing over batch k+2 -- with two slots that form lost to round 1's (profiles/r02/step_deferred_copy_ab.txt).
  static constexpr int kStepSlots = 3;
  struct StepSlot {
    void* in_h = nullptr;          // pinned staging: the copying forms (caller-owned arrays, received frames) go through it
    size_t in_bytes = 0;
    void* in_bar = nullptr;        // what raftq_step_stage hands out behind a large BAR: fine-grained DEVICE memory the
    size_t in_bar_bytes = 0;       // producer writes in place (posted PCIe writes); the batch then never needs a DMA
    void* dev = nullptr;           // device scratch: msgs, keys, order, outs, sort temp, flags
    size_t dev_bytes = 0;
    void* out_h = nullptr;         // pinned, device-mapped result records + {touched-group count, bad flag}
    void* out_d = nullptr;         // device alias of out_h
    size_t out_bytes = 0;
    uint64_t n = 0;
    hipEvent_t ev_in = nullptr, ev_comp = nullptr, ev_out = nullptr;
    bool busy = false;
    bool lists = false;            // submitted through the sort-free walk
    bool replayed = false;         // already re-run through the sorted path (after a stall)
    // the batch's result copy has not been enqueued yet: it rides in the NEXT batch's walk kernel (a copy kernel of its
    // own keeps that batch's kernels from starting until it retires), or is launched by the batch's collect
    bool copy_pending = false;
    const void* w_off_in_place = nullptr;     // raftq_step_stage_wire: the decoder read the frame boundaries / the stream where
    const void* w_stream_in_place = nullptr;  // the producer wrote them (in_bar)
    size_t w_stage_stream_off = 0;            // where raftq_step_stage_wire put the stream within the slot's staging
    const void* msgs_in_place = nullptr;  // the walk read this batch's records where the producer wrote them (in_bar), not in `dev`
    const void* outs_d = nullptr;  // device result records of this batch (in `dev`)
    uint64_t out_quads = 0;        // ... in 16-byte units, tail included
    int end_bit = 0;
    uint32_t rec = 64;             // bytes per result record of this batch (64, or 40 compact)
    bool tail_zeroed = false;      // the device copy of the 16-byte result tail is known to be zero (for tail_n, tail_dev)
    uint64_t tail_n = 0;
    uint32_t tail_rec = 0;
    const void* tail_dev = nullptr;
    uint64_t w_nbytes = 0;
    // raftq_step_submit_wire: where this batch's decoded records sit in `dev`, and the pinned
    // copies raftq_step_wire_msgs / _entries hand out (fetched on demand)
    bool wire = false;
    uint8_t recs = 0;              // raftqk::kRecsCaller / kRecsWire / kRecsFrames: whose records the batch holds (replays need it)
    void* w_msgs_d = nullptr;      // raftq_wire_msg_t [n]
    void* w_ents_d = nullptr;      // raftq_wire_ent_t [w_ents_cap]
    const uint64_t* w_ent_total_d = nullptr;
    uint64_t w_ents_cap = 0;
    void* w_pin = nullptr;         // pinned: msgs, then entries
    size_t w_pin_bytes = 0;
    bool w_msgs_fetched = false, w_ents_fetched = false;
    uint64_t w_n_ents = 0;
  } step_slot[kStepSlots];
  int step_last_slot = -1;         // slot of the last collected batch
  uint64_t step_submitted = 0, step_collected = 0;  // slot of batch k = k % kStepSlots
  hipStream_t step_s_in = nullptr, step_s_out = nullptr;
  int step_stream_mode = 2;
  int step_link_share = 25;        // per cent of a carried result copy that rides in the link kernel (the rest: the walk kernel)
  bool step_defer_copy = true;     // RAFTQ_STEP_DEFER_COPY=0: every batch launches its own result copy (round 1's form; for A/B)
  const void* step_last_out = nullptr;  // results of the last collected batch
  uint64_t step_last_n = 0;
  uint32_t step_last_rec = 64;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // raftq_sweep_many_async over handles that share one stream: every other launch goes to this auxiliary stream
  hipStream_t many_aux = nullptr;
  hipEvent_t many_fork = nullptr, many_join = nullptr;
  // wire / WAL codecs (raftq_wire.hip): growable device scratch (inputs + temporaries), device
  // output buffer, a small pinned block for totals / flags
  void* wire_dev = nullptr;
  size_t wire_dev_bytes = 0;
  void* wire_out = nullptr;
  size_t wire_out_bytes = 0;
  uint64_t* wire_pin = nullptr;    // pinned, 256 bytes
  uint64_t* wire_pin_d = nullptr;  // the same block as the device addresses it (the codecs' last kernel writes totals / flags there)
  unsigned long long* wire_flags = nullptr;  // device, 64 bytes, zero between calls: the codecs' malformed counters / bad flags
  // the streaming codec kernels (raftq_wire_kernels.hpp "the streaming form"): ticket word + per-tile look-back status
  unsigned long long* wire_lb = nullptr;     // device: kLbHead words {ticket, gave-up flag | landed waves | - | chunk ticket}, then kLbArrays status arrays of wire_lb_tiles words
  uint64_t wire_lb_tiles = 0;
  uint32_t wire_last_tiles = 0;              // tiles of the streaming decode enqueued last (measurement builds dump its stamps)
  uint32_t wire_ticket_base = 0, wire_epoch = 0;
  uint32_t wire_chunk_base = 0, wire_chunk_pending = 0;  // the readers' chunk tickets (the head's fourth word), accounted like the tiles'
  bool wire_chunk_unknown = false;           // the last launch had no reader workgroups: how far its chunk ticket got is not known
  unsigned int prop_stamp = 0;               // raftq_propose_frames: the call's stamp (its validation's verdict word holds it when a record was refused)
  hipStream_t wire_copy_stream = nullptr;    // RAFTQ_WIRE_SDMA=1 (A/B only): the runtime's copies bring the decoder's input in
  hipEvent_t wire_copy_ev = nullptr;
  // raftq_wal_encode_begin .. _end: enqueued, its totals in wire_pin[8 ..]; `done`: a later wait has covered it and what _end
  // will report is kept here
  bool wal_pending = false, wal_pending_done = false;
  bool wal_pending_waited = false;  // a wait on the handle's stream has come back since the begin: its kernel has run
  uint64_t wal_pending_n = 0, wal_pending_cap = 0;
  uint32_t wal_pending_prev = 0;
  int wal_pending_rc = 0;
  raftq_wal_counts_t wal_pending_counts{};
  std::string wal_pending_err;
  std::string err;
  // RAFTQ_PROFILE=1: host-side phase times of raftq_cycle, printed at destroy
  double prof[6] = {0, 0, 0, 0, 0, 0};
  uint64_t prof_n = 0;
};


// A set of handles of one shape on one GPU, swept by a single dispatch (raftq_set_*).
struct raftq_set {
  int device = 0;
  uint32_t N = 0;
  uint64_t gpad = 0;
  std::vector<raftq_t*> members;
  hipStream_t stream = nullptr;       // the set's stream; every member is re-homed onto it
  // device tables of the members' SweepArgs: [0] every member reads commit buffer 0, [1] buffer 1,
  // [2] rebuilt per launch when the members' buffers differ (someone swept a member on its own)
  raftqk::SweepArgs* tab[3] = {nullptr, nullptr, nullptr};
  std::vector<raftqk::SweepArgs> tab_host;  // pageable on purpose: hipMemcpyAsync stages it before returning
  uint64_t* counts_d = nullptr;       // [members][4] {changed, won, lost, 0}, reduced on the device
  uint64_t* np_d = nullptr;           // [members] number of per-wave partials of each member's last sweep
  std::vector<uint64_t> np_host;
  uint64_t* counts_h = nullptr;       // pinned
  bool swept = false;
  bool broken = false;                // a member was destroyed under the set
  unsigned last_flags = 0;
  int mode = 0;                       // 0 = K-deep grid, 1 = persistent walk (RAFTQ_SET_MODE / raftq_set_mode)
  uint32_t persist_wgs = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // raftq_set_tick: device table of the members' TickArgs (as of tick_host), set ticks dispatched since it was built
  raftqk::TickArgs* tick_tab = nullptr;
  std::vector<raftqk::TickArgs> tick_host;
  uint64_t tick_since = 0;
  std::string err;
};


struct raftq_wire_counts;  // raftq_wire.h
namespace raftqk {
struct NodeArrays;  // raftq_step_kernels.hpp
}
namespace raftq_detail {
// raftq_step.hip, for raftq_propose_frames (raftq_wire.hip): Step's device state exists (allocated on first use) and *out views it
int node_arrays_of(raftq_t* h, raftqk::NodeArrays* out);
int fail(raftq_t* h, int code, const std::string& msg);
int use_device(raftq_t* h);
int use_device_idle(raftq_t* h, const char* who);  // + no Step batch in flight (RAFTQ_ESTATE otherwise)
int ensure_staging(raftq_t* h, size_t bytes);   // pinned, device-mapped staging (term deltas, campaign lists, log deltas)
bool host_can_write(void* p, size_t bytes);     // [p, p + bytes) is mapped writable into this process (/proc/self/maps)
int ensure_ingest(raftq_t* h, size_t bytes);    // the ack buffer of the batching turn: device memory behind a large BAR, else pinned
unsigned host_coherence_flag();                 // hipHostMallocCoherent unless RAFTQ_HOST_COHERENT=0
int ensure_tick_state(raftq_t* h);              // role / elapsed / action (+ hup bitmap)
void free_node_state(raftq_t* h);               // raftq_step.hip's allocations (called by raftq_destroy)
void free_wire_state(raftq_t* h);               // raftq_wire.hip's allocations (called by raftq_destroy)
// raftq_wire.hip, for raftq_step_frames (raftq_step.hip): the streaming decode with a node's checks, enqueued on the handle's
// stream and not waited for (the records also go to msgs_d, in HBM, for the Step kernels enqueued behind it); and what the
// decode has to say once the call's one wait is over.  RAFTQ_EINVAL when an array is not page-locked and 16-byte aligned.
int wire_frames_enqueue(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, void* msgs, void* ents,
                        uint64_t ents_cap, void* msgs_d, int tail_appends, void* zero2 /* two device words left zero, or nullptr */);
int wire_frames_finish(raftq_t* h, const uint64_t* frame_off, uint64_t n, bool have_ents, uint64_t ents_cap, ::raftq_wire_counts* counts);
// The wait that ends a call whose results the kernels wrote into page-locked memory themselves (the streaming codecs,
// raftq_step_frames): a one-thread kernel raises the handle's completion word behind everything enqueued so far and the host
// polls it (raftq_cycle's wait, raftq_capi.hip wait_turn) -- a stream synchronisation costs 15-20 us more than the word does.
// The blocking wait where the word cannot be had or RAFTQ_CALL_WAIT=block.
hipError_t wait_call(raftq_t* h);
}  // namespace raftq_detail

#define HIPCHK(h, expr)                                                                        \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      return raftq_detail::fail((h), _e == hipErrorOutOfMemory ? RAFTQ_ENOMEM : RAFTQ_EHIP,    \
                                std::string(#expr) + ": " + hipGetErrorString(_e));            \
    }                                                                                          \
  } while (0)
