// raftq_step_kernels.hpp -- device code of the batched raft Step (include/raftq_step.h).
//
// etcd's raft.Step is a per-group sequential state machine; what is parallel is the G
// groups.  A batch is therefore grouped by raft group with the arrival order kept, and one
// lane per group gathers its group's scalar state into registers, applies the group's
// messages in order and scatters the state back.  Two ways to group (same bytes out):
//   list walk (default):  step_link_kernel threads every message onto its group's list with
//       three atomics, the lane of the group's first message orders and walks it (2b / 3b)
//   sorted walk (fallback for long runs): (1) keys, (2) stable radix sort of (group, position)
//       (raftq_sort_kernels.hpp), (3) step_kernel: one lane per run of equal keys
// Different lanes touch different groups, so there are no atomics on state and the result is
// identical to calling Step message by message (tests/test_step_gpu.py).
//
// Restates (2015-era etcd raft, reached from raft.go:268-270 / :223-224): Step, stepLeader,
// stepCandidate, stepFollower, reset, becomeFollower / becomeCandidate / becomeLeader,
// campaign, poll, maybeCommit, handleHeartbeat, raftLog.isUpToDate / commitTo,
// Progress.maybeUpdate.  Sparse, scattered access: bound by HBM latency / PCIe, not bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raftq_kernels.hpp"

namespace raftqk {

struct MsgRec {  // == raftq_msg_t
  uint64_t group, term, log_term, index, commit, reject_hint;
  uint32_t from;
  uint8_t type, reject, pad[2];
  uint64_t resv;
};
struct StepOutRec {  // == raftq_step_out_t
  uint64_t group, term, index, log_term, commit, last_index;
  uint32_t to, vote, lead;
  uint8_t type, reject, flags, role;
};
struct LogDeltaRec {  // == raftq_log_delta_t
  uint64_t group, last_index, last_term, commit_to;
};
struct StepOutC {  // == raftq_step_out_c_t: the result record without what the caller's own batch already says
  uint64_t term, index, commit, aux;
  uint8_t vote, lead, type, reject, flags, role, pad[2];
};
struct StepOutS {  // == raftq_step_out_s_t: ... and without aux (include/raftq_step.h says what carried it)
  uint64_t term, index, commit;
  uint8_t vote, lead, type, reject, flags, role, pad[2];
};
static_assert(sizeof(MsgRec) == 64 && sizeof(StepOutRec) == 64 && sizeof(LogDeltaRec) == 32 && sizeof(StepOutC) == 40 && sizeof(StepOutS) == 32,
              "record layout");
constexpr uint8_t kFmtFull = 0, kFmtC40 = 1, kFmtS32 = 2;  // raftq_step_set_compact

// Everything Step's common paths need of a group, in ONE 128-byte line (round 6): the raft scalars, the words of the group's
// message list of the batch in flight (step_link_kernel / step_lists_kernel), and a copy of what the dense kernels own -- role,
// committed, the current-term gate and the N match words.  Round 3 kept each scalar in an array of its own (18 scattered lines
// per touched group); round 4 made the scalars one 64-byte record but still read role, committed, match[from] and -- for every
// acknowledgement -- all N match rows and the gate from their dense arrays: 33 MB read per 64K-message batch by the DRAM request
// counters, x2.3 of what the walk needs (profiles/r05/pmc_legs_summary.txt step_lists_kernel: 1.05 M 32-byte requests).  Now a
// touched group is one line in and one line out.
//
// What the dense kernels stream stays peer-major / group-major SoA and stays AUTHORITATIVE: match, committed (the sweep),
// first_idx (the gated sweep), role / elapsed (Tick), the vote words (the tally).  Step writes a word it changes to BOTH places.
// A call that changes the dense arrays without Step (raftq_load_*, a batching turn's ingest, an adopted sweep, a campaign list)
// marks the copies stale (raftq_t::node_mirror_fresh); the next Step-family call re-reads them (node_mirror_kernel) first.
// raftq_node -- Step, tail reports, proposals, Tick (which writes neither) -- never does.
struct __attribute__((aligned(128))) NodeRec {
  uint64_t term, last_index, last_term;
  uint32_t lst_head;  // batch position of the last-linked message of the group, kNil = none
  uint32_t lst_cnt;   // messages of the group in this batch
  uint32_t lst_min;   // smallest batch position of the group
  uint8_t vote, lead; // 0 = None, else slot + 1
  uint8_t role, pad;  // copy of role[g]
  uint64_t committed, first_idx;  // copies of committed[g], first_idx[g]
  uint64_t match[kMaxPeers];      // copies of match[p][g]
};
static_assert(sizeof(NodeRec) == 128, "one cache line");

struct NodeArrays {
  NodeRec* rec;     // [ld]
  uint8_t* role;
  uint32_t* elapsed;
  uint64_t* committed;
  uint64_t* first_idx;
  uint64_t* match;  // [N][ld]
  uint8_t* votes;   // [ld] packed vote words: 2 bits per peer, 16-bit words (N <= 8) or 32-bit (N = 9)
  uint64_t ld;
  uint32_t n_peers, self;
  bool msg_flags;   // the handle opted in to RAFTQ_MSGF_* (raftq_step_set_msg_flags): otherwise a record's pad bytes are padding
  uint8_t recs;     // whose records the batch holds: kRecsCaller, kRecsWire (raftq_step_submit_wire: the decoder's, strict) or
                    // kRecsFrames (raftq_step_frames: the decoder's, with RAFTQ_MSGF_* set by it)
  uint64_t n_groups;
};
constexpr uint8_t kRecsCaller = 0, kRecsWire = 1, kRecsFrames = 2;

constexpr uint8_t kMsgHup = 0, kMsgBeat = 1, kMsgApp = 3, kMsgAppResp = 4, kMsgVote = 5, kMsgVoteResp = 6,
                  kMsgHeartbeat = 8, kMsgHeartbeatResp = 9;
constexpr uint8_t kOutNone = 0, kOutVoteResp = 1, kOutHeartbeatResp = 2, kOutCampaign = 3, kOutBecameLeader = 4,
                  kOutProgress = 5, kOutBcastHeartbeat = 6, kOutAppend = 7, kOutAppended = 8, kOutDeferred = 9, kOutSkipped = 10,
                  kOutHeld = 11;
constexpr uint8_t kMsgfEntries = 0x80, kMsgfBarrier = 0x40, kMsgfHold = 0x20, kMsgfSkip = 0x10;  // raftq_msg_t._pad[1]: RAFTQ_MSGF_*
constexpr uint8_t kFlagHardState = 1, kFlagCommitted = 2, kFlagUpdated = 4, kFlagSteppedDown = 8;
constexpr uint8_t kFollower = 0, kCandidate = 1, kLeader = 2;

// result record i, in the handle's format.  Compact drops group and addressee (= msgs[i].group / .from) and folds
// log_term / last_index into one slot: log_term is only set by the two result types whose index IS last_index.
__device__ __forceinline__ void put_result(void* out, uint64_t i, const StepOutRec& o, uint8_t fmt) {
  if (fmt == kFmtFull) {
    static_cast<StepOutRec*>(out)[i] = o;
    return;
  }
  if (fmt == kFmtS32) {
    StepOutS c;
    c.term = o.term;
    c.index = o.index;
    c.commit = o.type == kOutCampaign ? o.log_term : o.commit;
    c.vote = (uint8_t)o.vote;
    c.lead = (uint8_t)o.lead;
    c.type = o.type;
    c.reject = o.reject;
    c.flags = o.flags;
    c.role = o.role;
    c.pad[0] = c.pad[1] = 0;
    static_cast<StepOutS*>(out)[i] = c;
    return;
  }
  StepOutC c;
  c.term = o.term;
  c.index = o.index;
  c.commit = o.commit;
  c.aux = (o.type == kOutCampaign || o.type == kOutBecameLeader) ? o.log_term : o.last_index;
  c.vote = (uint8_t)o.vote;
  c.lead = (uint8_t)o.lead;
  c.type = o.type;
  c.reject = o.reject;
  c.flags = o.flags;
  c.role = o.role;
  c.pad[0] = c.pad[1] = 0;
  static_cast<StepOutC*>(out)[i] = c;
}

// zero the batch's {touched-group count, bad flag} (a plain kernel: no runtime blit in the chain)
static __global__ void step_reset_kernel(unsigned long long* flags) {
  if (threadIdx.x < 2) flags[threadIdx.x] = 0;
}

// ---- (0) packed inbound records (raftq_msg40_t) -> the 64-byte records everything below reads.  In HBM: the
// widening costs a 3 us launch and no PCIe byte.
struct Msg40Rec {  // == raftq_msg40_t
  uint32_t group;
  uint8_t from, type, reject, pad;
  uint64_t term, index, aux, commit;
};
static_assert(sizeof(Msg40Rec) == 40, "ABI struct mismatch");

static __global__ __launch_bounds__(kBlock) void step_unpack40_kernel(const Msg40Rec* __restrict__ in, MsgRec* __restrict__ out,
                                                                      uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const Msg40Rec r = in[i];
  MsgRec m;
  m.group = r.group;
  m.term = r.term;
  m.index = r.index;
  m.commit = r.commit;
  const bool hint = r.type == kMsgAppResp;  // the one kind that carries a RejectHint (and no LogTerm)
  m.log_term = hint ? 0 : r.aux;
  m.reject_hint = hint ? r.aux : 0;
  m.from = r.from;
  m.type = r.type;
  m.reject = r.reject;
  m.pad[0] = m.pad[1] = 0;
  m.resv = 0;
  out[i] = m;
}

// What the two grouping kernels make of record i.  kSkip: RAFTQ_MSGF_SKIP, nobody's (answered on the spot, never grouped);
// kBad: fails the batch; kTake: goes to its group (a RAFTQ_MSGF_HOLD record too: only its group is looked at).
enum RecClass : int { kTake = 0, kSkip = 1, kBad = 2 };
__device__ __forceinline__ RecClass classify(const MsgRec& m, uint64_t n_groups, uint32_t n_peers, bool msg_flags, uint8_t recs) {
  const uint8_t fl = msg_flags ? m.pad[1] : (uint8_t)0;
  if (fl & kMsgfSkip) return kSkip;
  if (m.group >= n_groups) return kBad;
  if (fl & kMsgfHold) return kTake;
  const uint8_t t = m.type;
  const bool local = t == kMsgHup || t == kMsgBeat;
  const bool known = local || t == kMsgApp || t == kMsgAppResp || t == kMsgVote || t == kMsgVoteResp ||
                     t == kMsgHeartbeat || t == kMsgHeartbeatResp;
  if (!known || (!local && m.from >= n_peers)) return kBad;
  // records decoded on the device from stream frames (raftq_step_submit_wire) also carry the addressee's slot and the
  // decoder's flags in the two pad bytes: a frame that did not parse, or one addressed to no peer of this cluster, fails
  // the batch like any malformed message.  (raftq_step_frames' decoder has made those RAFTQ_MSGF_SKIP itself.)
  if (recs == kRecsWire && ((m.pad[1] & 1u) != 0 || m.pad[0] >= n_peers)) return kBad;
  return kTake;
}
__device__ __forceinline__ void put_skipped(void* out, uint64_t i, uint8_t compact) {
  StepOutRec o;
  __builtin_memset(&o, 0, sizeof o);
  o.type = kOutSkipped;
  put_result(out, i, o, compact);
}

// ---- (1) validate + sort keys.  The batch was DMA-copied from the pinned staging area into
// HBM; one lane per record reads its first 16 B (group, term) and the 16 B holding from/type.
// A malformed record raises *bad; step_kernel then applies nothing (the ABI's all-or-nothing rule).
static __global__ __launch_bounds__(kBlock) void step_keys_kernel(const MsgRec* __restrict__ msgs,
                                                                  uint64_t* __restrict__ keys,
                                                                  uint32_t* __restrict__ order, uint64_t n,
                                                                  uint64_t n_groups, uint32_t n_peers,
                                                                  unsigned int* bad, bool msg_flags, uint8_t recs,
                                                                  void* __restrict__ out, uint8_t compact) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool is_bad = false;
  if (i < n) {
    const RecClass c = classify(msgs[i], n_groups, n_peers, msg_flags, recs);
    is_bad = c == kBad;
    // a skipped record sorts behind every group (the key range has room for n_groups itself) and is answered here
    keys[i] = c == kTake ? msgs[i].group : c == kSkip ? n_groups : 0;  // (bad: keep the sort's key range valid)
    order[i] = (uint32_t)i;
    if (c == kSkip) put_skipped(out, i, compact);
  }
  if (__ballot(is_bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// ---- (3) one group's state in registers ------------------------------------------------------
constexpr uint32_t kMaxRun = 32;
constexpr uint32_t kNil = 0xffffffffu;

struct Node {
  const NodeArrays& a;
  uint64_t g;
  uint64_t term, last_index, last_term, committed, first_idx;
  uint64_t mt[kMaxPeers];  // Progress.Match of every peer, from the record
  uint32_t mt_dirty = 0;   // bit p: mt[p] differs from the dense match[p][g]
  uint32_t vote, lead;
  uint32_t vw = 0;  // the group's vote word (raft.votes as 2 bits per peer)
  uint32_t lst_head, lst_cnt, lst_min;  // the record's list words as loaded (the walk empties them when it stores)
  uint8_t role;
  bool held = false;  // this batch only: a MsgApp with RAFTQ_MSGF_BARRIER was left to the caller -- the group's later messages wait
  // what has to be written to the dense arrays besides the record (which always is, list words emptied)
  uint64_t committed0, first_idx0;
  uint8_t role0;
  bool vw_known = false, vw_dirty = false, elapsed_reset = false;

  __device__ Node(const NodeArrays& arr, uint64_t group) : a(arr), g(group) {
    const NodeRec r = a.rec[g];  // one line: eight 16-byte loads
    term = r.term; last_index = r.last_index; last_term = r.last_term;
    vote = r.vote; lead = r.lead;
    lst_head = r.lst_head; lst_cnt = r.lst_cnt; lst_min = r.lst_min;
    committed = committed0 = r.committed;
    first_idx = first_idx0 = r.first_idx;
    role = role0 = r.role;
#pragma unroll
    for (int p = 0; p < kMaxPeers; ++p) mt[p] = r.match[p];
  }
  // list_reset: the walk hands the group's list back empty (log_deltas_kernel leaves the words as they are)
  __device__ void store(bool list_reset = true) const {
    NodeRec r;
    r.term = term; r.last_index = last_index; r.last_term = last_term;
    r.vote = (uint8_t)vote; r.lead = (uint8_t)lead;
    r.role = role; r.pad = 0;
    r.lst_head = list_reset ? kNil : lst_head;
    r.lst_cnt = list_reset ? 0u : lst_cnt;
    r.lst_min = list_reset ? kNil : lst_min;
    r.committed = committed;
    r.first_idx = first_idx;
#pragma unroll
    for (int p = 0; p < kMaxPeers; ++p) r.match[p] = mt[p];
    a.rec[g] = r;
    // ... and what changed, where the dense kernels read it
    if (committed != committed0) a.committed[g] = committed;
    if (role != role0) a.role[g] = role;
    if (first_idx != first_idx0) a.first_idx[g] = first_idx;
#pragma unroll
    for (int p = 0; p < kMaxPeers; ++p)
      if ((mt_dirty >> p) & 1u) a.match[(uint64_t)p * a.ld + g] = mt[p];
    if (elapsed_reset) a.elapsed[g] = 0;
    if (vw_dirty) {
      if (a.n_peers <= 8) reinterpret_cast<uint16_t*>(a.votes)[g] = (uint16_t)vw;
      else reinterpret_cast<uint32_t*>(a.votes)[g] = vw;
    }
  }
  __device__ void set_first_idx(uint64_t v) { first_idx = v; }
  __device__ uint64_t get_first_idx() const { return first_idx; }
  __device__ void set_votes(uint32_t w) {
    vw = w;
    vw_known = vw_dirty = true;
  }
  __device__ uint32_t get_votes() {
    if (!vw_known) {
      vw = a.n_peers <= 8 ? (uint32_t)reinterpret_cast<const uint16_t*>(a.votes)[g] : reinterpret_cast<const uint32_t*>(a.votes)[g];
      vw_known = true;
    }
    return vw;
  }
  // Progress.Match of peer p (a run-time index into registers: selects, not scratch)
  __device__ uint64_t match(uint32_t p) const {
    uint64_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)kMaxPeers; ++k) v = k == p ? mt[k] : v;
    return v;
  }
  __device__ void set_match(uint32_t p, uint64_t v) {
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)kMaxPeers; ++k) mt[k] = k == p ? v : mt[k];
    mt_dirty |= 1u << p;
  }
  __device__ uint32_t quorum() const { return a.n_peers / 2 + 1; }

  // raft.reset(term)
  __device__ void reset(uint64_t t) {
    if (term != t) { term = t; vote = 0; }
    lead = 0;
    elapsed_reset = true;
    set_votes(0);
#pragma unroll
    for (uint32_t p = 0; p < (uint32_t)kMaxPeers; ++p)
      if (p < a.n_peers) {
        const uint64_t v = p == a.self ? last_index : 0;
        if (mt[p] != v) mt_dirty |= 1u << p;
        mt[p] = v;
      }
  }
  __device__ void become_follower(uint64_t t, uint32_t new_lead) {
    reset(t);
    lead = new_lead;
    role = kFollower;
    set_first_idx(0);
  }
  __device__ void become_candidate() {
    reset(term + 1);
    vote = a.self + 1;
    role = kCandidate;
    set_first_idx(0);
  }
  // raft.maybeCommit + raftLog.maybeCommit: the largest index held by >= q peers (counting form),
  // then the compact current-term gate
  __device__ bool maybe_commit() {
    uint64_t m[kMaxPeers];
    const uint32_t n = a.n_peers, q = quorum();
#pragma unroll
    for (uint32_t p = 0; p < kMaxPeers; ++p) m[p] = p < n ? mt[p] : 0;
    uint64_t mci = 0;
#pragma unroll
    for (uint32_t c = 0; c < kMaxPeers; ++c) {
      uint32_t ge = 0;
#pragma unroll
      for (uint32_t p = 0; p < kMaxPeers; ++p) ge += (p < n && m[p] >= m[c]) ? 1u : 0u;
      if (c < n && ge >= q && m[c] > mci) mci = m[c];
    }
    if (mci > committed) {  // (the gate's word is only read when there is something to gate)
      const uint64_t fi = get_first_idx();
      if (fi != 0 && mci >= fi) {
        committed = mci;
        return true;
      }
    }
    return false;
  }
  // raft.becomeLeader incl. appendEntry(empty entry of the new term)
  __device__ void become_leader() {
    reset(term);
    lead = a.self + 1;
    role = kLeader;
    last_index += 1;
    last_term = term;
    set_first_idx(last_index);
    set_match(a.self, last_index);
    (void)maybe_commit();
  }
  // raft.poll: the first response of a peer wins; returns {granted, recorded}
  __device__ void poll(uint32_t from, bool granted, uint32_t& n_granted, uint32_t& n_recorded) {
    uint32_t w = get_votes();
    const uint32_t cur = (w >> (2 * from)) & 3u;
    if (cur != 1 && cur != 2) {
      w = (w & ~(3u << (2 * from))) | ((granted ? 1u : 2u) << (2 * from));
      set_votes(w);
    }
    const uint32_t low = 0x55555555u & ((1u << (2 * a.n_peers)) - 1u);
    const uint32_t g1 = w & ~(w >> 1) & low, r1 = (w >> 1) & ~w & low;
    n_granted = __popc(g1);
    n_recorded = __popc(g1 | r1);
  }
  __device__ void commit_to(uint64_t tocommit) {
    if (tocommit > last_index) tocommit = last_index;
    if (committed < tocommit) committed = tocommit;
  }

  // handleAppendEntries once the header is accepted.  A message that says what it carries (RAFTQ_MSGF_ENTRIES) and
  // appends at the log's tail needs nothing from the entries but their count and the last one's term:
  // raftLog.maybeAppend = matchTerm(tail) -> no conflict possible -> append -> commitTo(min(m.Commit, lastnewi)).
  __device__ void handle_append(const MsgRec& m, StepOutRec& o) {
    o.type = kOutAppend;
    if (a.msg_flags && (m.pad[1] & kMsgfEntries) && m.index == last_index && m.log_term == last_term) {
      // (the decoder's record keeps ent_first | n_ents << 32 where the caller's keeps the count)
      const uint64_t k = a.recs == kRecsFrames ? m.resv >> 32 : m.resv & 0xffffffffull;
      if (k) {
        last_index = m.index + k;
        last_term = m.reject_hint;
      }
      commit_to(m.commit);  // clamps to last_index = lastnewi
      o.type = kOutAppended;
      o.index = last_index;
    }
  }

  __device__ void step(const MsgRec& m, StepOutRec& o) {
    const uint64_t term0 = term, commit0 = committed;
    const uint32_t vote0 = vote;
    const uint8_t role0 = role;
    o.index = 0; o.log_term = 0; o.type = kOutNone; o.reject = 0; o.flags = 0;
    const bool hold = a.msg_flags && (m.pad[1] & kMsgfHold) != 0;  // RAFTQ_OUT_HELD: the caller's, where it stands; the rest waits
    if (held || hold) {  // RAFTQ_OUT_DEFERRED: nothing of this message is applied
      o.type = hold ? kOutHeld : kOutDeferred;
      held = true;
      o.group = g; o.term = term; o.commit = committed; o.last_index = last_index;
      o.to = m.from; o.vote = vote; o.lead = lead; o.role = role;
      return;
    }
    const uint32_t q = quorum();
    bool handled = false;
    if (m.type == kMsgHup) {
      handled = true;
      if (role != kLeader) {  // campaign()
        become_candidate();
        uint32_t gr, rec;
        poll(a.self, true, gr, rec);
        if (gr == q) { become_leader(); o.type = kOutBecameLeader; }
        else o.type = kOutCampaign;
        o.index = last_index;
        o.log_term = last_term;
      }
    } else if (m.term != 0) {
      if (m.term > term) become_follower(m.term, m.type == kMsgVote ? 0u : m.from + 1);
      else if (m.term < term) handled = true;  // stale: ignored
    }
    if (!handled) {
      if (role == kLeader) {
        if (m.type == kMsgBeat) {
          o.type = kOutBcastHeartbeat;
        } else if (m.type == kMsgVote) {
          o.type = kOutVoteResp; o.reject = 1;
        } else if (m.type == kMsgAppResp) {
          o.type = kOutProgress; o.reject = m.reject;
          if (!m.reject) {
            const uint64_t idx = m.index > last_index ? last_index : m.index;
            if (match(m.from) < idx) {
              set_match(m.from, idx);
              o.flags |= kFlagUpdated;
              (void)maybe_commit();
            }
          }
          o.index = match(m.from);
        } else if (m.type == kMsgHeartbeatResp) {
          o.type = kOutProgress;
          o.index = match(m.from);
        }
      } else if (role == kCandidate) {
        if (m.type == kMsgApp) {
          become_follower(term, m.from + 1);
          handle_append(m, o);
        } else if (m.type == kMsgHeartbeat) {
          become_follower(term, m.from + 1);
          commit_to(m.commit);
          o.type = kOutHeartbeatResp;
        } else if (m.type == kMsgVote) {
          o.type = kOutVoteResp; o.reject = 1;
        } else if (m.type == kMsgVoteResp) {
          uint32_t gr, rec;
          poll(m.from, !m.reject, gr, rec);
          if (gr == q) {
            become_leader();
            o.type = kOutBecameLeader; o.index = last_index; o.log_term = last_term;
          } else if (rec - gr == q) {
            become_follower(term, 0);
          }
        }
      } else {
        if (m.type == kMsgApp) {
          elapsed_reset = true; lead = m.from + 1;
          handle_append(m, o);
        } else if (m.type == kMsgHeartbeat) {
          elapsed_reset = true; lead = m.from + 1;
          commit_to(m.commit);
          o.type = kOutHeartbeatResp;
        } else if (m.type == kMsgVote) {
          o.type = kOutVoteResp;
          const bool up_to_date = m.log_term > last_term || (m.log_term == last_term && m.index >= last_index);
          if ((vote == 0 || vote == m.from + 1) && up_to_date) { elapsed_reset = true; vote = m.from + 1; }
          else o.reject = 1;
        }
      }
    }
    if (a.msg_flags && o.type == kOutAppend && m.type == kMsgApp && (m.pad[1] & kMsgfBarrier)) held = true;
    o.group = g; o.term = term; o.commit = committed; o.last_index = last_index;
    o.to = m.from; o.vote = vote; o.lead = lead; o.role = role;
    if (term != term0 || vote != vote0 || committed != commit0) o.flags |= kFlagHardState;
    if (committed != commit0) o.flags |= kFlagCommitted;
    if (role0 != kFollower && role == kFollower) o.flags |= kFlagSteppedDown;
  }
};

static __global__ __launch_bounds__(kBlock) void step_kernel(NodeArrays a, const MsgRec* __restrict__ msgs,
                                                      const uint64_t* __restrict__ keys_sorted,
                                                      const uint32_t* __restrict__ order, void* __restrict__ out,
                                                      uint8_t compact, uint64_t n, unsigned long long* n_heads,
                                                      const unsigned int* bad) {
  if (*bad) return;  // a malformed record somewhere in the batch: nothing is applied
  const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  bool head = false;
  uint64_t g = 0;
  if (k < n) {
    g = keys_sorted[k];
    head = (k == 0 || keys_sorted[k - 1] != g) && g < a.n_groups;  // (key n_groups: the RAFTQ_MSGF_SKIP records, answered by step_keys_kernel)
  }
  const uint64_t hb = __ballot(head);
  if ((threadIdx.x & 63) == 0 && hb) atomicAdd(n_heads, (unsigned long long)__popcll(hb));
  if (!head) return;
  Node node(a, g);
  for (uint64_t j = k; j < n && keys_sorted[j] == g; ++j) {
    const uint32_t i = order[j];
    const MsgRec m = msgs[i];
    StepOutRec o;
    node.step(m, o);
    put_result(out, i, o, compact);
  }
  node.store();
}

// ---- (2b, 3b) the same walk WITHOUT the sort.  Inbound traffic rarely brings more than a few messages
// of one group in one batch, so grouping by a full stable sort (4-12 launches, 40-60 us of a 95 us kernel
// chain at 64K messages) is the wrong tool.  step_link_kernel threads every message onto its group's list
// with three atomics on group-indexed arrays (exchange the list head, count, minimum batch position);
// the lane of a group's FIRST message then owns the group: it gathers the list (arbitrary order), sorts
// the <= kMaxRun positions in a private array, applies the messages in arrival order and empties the list -- the same
// sequence the sorted walk applies, so the result records are identical byte for byte
// (tests/test_step_gpu.py runs both paths).  A batch with a longer run sets the handle's `stall` word:
// nothing of it -- nor of any later batch already in flight -- is applied, and the host replays those
// batches, in order, through the sorted path (raftq_step.hip: replay_stalled).
// A slice [q0, q1) of the PREVIOUS batch's result records on its way to the host, carried by the first `blocks`
// workgroups of a kernel of THIS batch (step_d2h_kernel's loop).  As a kernel of its own that copy holds back this
// batch's kernels until it retires -- measured in round 1, profiles/r01/step_pipeline_trace.txt -- so a pipelined batch
// cost walk + copy; inside this batch's kernels the two overlap (the copy is PCIe-bound, the kernels HBM-latency-bound).
// The slice that ends the records also clears the device copy of the tail for the slot's next batch (zero_tail).
struct CopyRide {
  const u64x2* src;
  u64x2* dst;
  uint64_t q0, q1, q_last;  // q_last: index of the tail quad
  uint32_t blocks;
};

__device__ __forceinline__ void copy_ride(const CopyRide& c) {
  const uint64_t stride = (uint64_t)c.blocks * kBlock;
  for (uint64_t q = c.q0 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < c.q1; q += stride) {
    const u64x2 v = c.src[q];
    __builtin_nontemporal_store(v, c.dst + q);
    if (q == c.q_last) const_cast<u64x2*>(c.src)[q] = u64x2{0, 0};
  }
}

static __global__ __launch_bounds__(kBlock) void step_link_kernel(const MsgRec* __restrict__ msgs, uint64_t n,
                                                                  uint64_t n_groups, uint32_t n_peers, bool msg_flags, uint8_t recs,
                                                                  NodeRec* rec, uint32_t* __restrict__ next,
                                                                  unsigned int* bad, unsigned int* stall,
                                                                  void* __restrict__ out, uint8_t compact, CopyRide ride) {
  if (blockIdx.x < ride.blocks) {
    copy_ride(ride);
    return;
  }
  const uint64_t i = (uint64_t)(blockIdx.x - ride.blocks) * kBlock + threadIdx.x;
  bool is_bad = false, too_long = false;
  if (i < n) {
    const RecClass c = classify(msgs[i], n_groups, n_peers, msg_flags, recs);
    is_bad = c == kBad;
    if (c == kTake) {
      NodeRec* r = rec + msgs[i].group;  // three atomics on one line
      next[i] = atomicExch(&r->lst_head, (uint32_t)i);
      too_long = atomicAdd(&r->lst_cnt, 1u) + 1 > kMaxRun;
      atomicMin(&r->lst_min, (uint32_t)i);
    } else if (c == kSkip) {
      // nobody's: answered on the spot.  (A batch that turns out bad or stalled is not applied and its results are not
      // read -- a replay through the sorted walk writes this record again.)
      put_skipped(out, i, compact);
    }
  }
  if (__ballot(is_bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
  if (__ballot(too_long) != 0 && (threadIdx.x & 63) == 0) atomicOr(stall, 1u);
}

static __global__ __launch_bounds__(kBlock) void step_lists_kernel(NodeArrays a, const MsgRec* __restrict__ msgs,
                                                                   void* __restrict__ out, uint8_t compact, uint64_t n,
                                                                   uint64_t n_groups,
                                                                   const uint32_t* __restrict__ next,
                                                                   unsigned long long* n_heads, unsigned int* tail_skipped,
                                                                   const unsigned int* bad, const unsigned int* stall, CopyRide ride) {
  if (blockIdx.x < ride.blocks) {  // the rest of the previous batch's results (see CopyRide)
    copy_ride(ride);
    return;
  }
  const uint64_t i = (uint64_t)(blockIdx.x - ride.blocks) * kBlock + threadIdx.x;
  const bool stalled = *stall != 0;  // this batch, or one before it that has not been replayed yet, needs the sorted path
  if (stalled || *bad) {             // (*bad: a malformed record somewhere in the batch) -- nothing is applied;
    if (stalled && i == 0) *tail_skipped = 1u;
    if (i < n && classify(msgs[i], n_groups, a.n_peers, a.msg_flags, a.recs) == kTake) {  // every message empties its group's list words (idempotent)
      NodeRec* r = a.rec + msgs[i].group;
      r->lst_head = kNil;
      r->lst_cnt = 0;
      r->lst_min = kNil;
    }
    return;
  }
  uint64_t g = 0;
  bool owner = false;
  if (i < n) {
    g = msgs[i].group;
    // (a RAFTQ_MSGF_SKIP record belongs to no group -- its group field may hold anything -- and was answered by the link kernel)
    owner = !(a.msg_flags && (msgs[i].pad[1] & kMsgfSkip)) && a.rec[g].lst_min == (uint32_t)i;
  }
  const uint64_t ob = __ballot(owner);
  if ((threadIdx.x & 63) == 0 && ob) atomicAdd(n_heads, (unsigned long long)__popcll(ob));
  if (!owner) return;
  Node node(a, g);  // (the record's line is in the L1 already: the owner test read it)
  const uint32_t c = node.lst_cnt;
  if (c == 1) {
    StepOutRec o;
    node.step(msgs[i], o);
    put_result(out, i, o, compact);
  } else {
    uint32_t pos[kMaxRun];
    uint32_t p = node.lst_head;
    for (uint32_t k = 0; k < c; ++k) {  // gather, inserting in ascending order of batch position
      uint32_t j = k;
      while (j > 0 && pos[j - 1] > p) {
        pos[j] = pos[j - 1];
        --j;
      }
      pos[j] = p;
      p = next[p];
    }
    for (uint32_t k = 0; k < c; ++k) {
      const MsgRec m = msgs[pos[k]];
      StepOutRec o;
      node.step(m, o);
      put_result(out, pos[k], o, compact);
    }
  }
  // the record goes back with the group's list empty for the next batch.  Other lanes of this group only compare lst_min
  // with their own position to learn that they are not the owner: kNil tells them the same.
  node.store();
}

// ---- the record array's bulk paths (raftq_load_node / raftq_read_node: set-up and test traffic, not the hot path) ----------
static __global__ __launch_bounds__(kBlock) void node_init_kernel(NodeRec* rec, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  NodeRec r;
  __builtin_memset(&r, 0, sizeof r);
  r.lst_head = r.lst_min = kNil;
  rec[i] = r;
}
// the record's copies of what the dense kernels own, re-read (see NodeRec): one lane per group
static __global__ __launch_bounds__(kBlock) void node_mirror_kernel(NodeArrays a, uint64_t n) {
  const uint64_t g = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= n) return;
  NodeRec* r = a.rec + g;
  r->role = a.role[g];
  r->committed = a.committed[g];
  r->first_idx = a.first_idx[g];
#pragma unroll
  for (uint32_t p = 0; p < (uint32_t)kMaxPeers; ++p) r->match[p] = p < a.n_peers ? a.match[(uint64_t)p * a.ld + g] : 0;
}
// field: 0 term, 1 vote, 2 lead, 3 last_index, 4 last_term.  `flat` is the ABI's array for groups [g0, g0 + n): u64 or u32.
template <bool PUT>
static __global__ __launch_bounds__(kBlock) void node_field_kernel(NodeRec* rec, uint64_t g0, uint64_t n, int field, void* flat) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  NodeRec* r = rec + g0 + i;
  uint64_t* f64 = static_cast<uint64_t*>(flat);
  uint32_t* f32 = static_cast<uint32_t*>(flat);
  if (PUT) {
    if (field == 0) r->term = f64[i];
    else if (field == 1) r->vote = (uint8_t)f32[i];
    else if (field == 2) r->lead = (uint8_t)f32[i];
    else if (field == 3) r->last_index = f64[i];
    else r->last_term = f64[i];
  } else {
    if (field == 0) f64[i] = r->term;
    else if (field == 1) f32[i] = r->vote;
    else if (field == 2) f32[i] = r->lead;
    else if (field == 3) f64[i] = r->last_index;
    else f64[i] = r->last_term;
  }
}

// ---- (4) results -> pinned, device-mapped host memory.  A small grid (grid-stride, 16 B per lane): the
// transfer is PCIe-bound and 64 workgroups keep the link full (77 us for 4 MB, the same as the
// runtime's own blit copy).
// zero_tail: the last quad is the batch's {touched count, bad, skipped} tail; once it is on its way to the host the
// device copy is cleared for the slot's next batch (saves that batch a reset launch).
static __global__ __launch_bounds__(kBlock) void step_d2h_kernel(const u64x2* __restrict__ src, u64x2* __restrict__ dst,
                                                                 uint64_t n_quads, bool zero_tail = false) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_quads; i += stride) {
    const u64x2 v = src[i];
    __builtin_nontemporal_store(v, dst + i);
    if (zero_tail && i == n_quads - 1) const_cast<u64x2*>(src)[i] = u64x2{0, 0};
  }
}

// the log owner's tail reports; records are unique per group within a launch (the host splits
// repeated groups into successive launches)
static __global__ __launch_bounds__(kBlock) void log_deltas_kernel(NodeArrays a, const LogDeltaRec* __restrict__ d,
                                                                   uint64_t n, uint64_t* __restrict__ committed_out) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const LogDeltaRec r = d[i];
  Node node(a, r.group);
  node.last_index = r.last_index;
  node.last_term = r.last_term;
  if (node.role == kLeader) {
    if (node.match(a.self) < node.last_index) node.set_match(a.self, node.last_index);
    (void)node.maybe_commit();
  } else if (r.commit_to != 0) {
    node.commit_to(r.commit_to);
  }
  node.store(false);
  if (committed_out) committed_out[i] = node.committed;
}

}  // namespace raftqk
