/*
 * raftq_node.h -- one raft NODE for G groups: the reference's raftNode
 * (raft.go:36-78) multiplied by G, with the per-group pipe surface kept.
 *
 * The reference runs one raft group per process: newRaftNode wires a raft.Node
 * to a WAL, a rafthttp transport and the three pipe channels
 * (raftpipe.go:3-17), and serveChannels (raft.go:204-246) turns the crank:
 *     ticker.C            -> rc.node.Tick()                       raft.go:223-224
 *     proposeC            -> rc.node.Propose(ctx, []byte(prop))   raft.go:211-215
 *     transport           -> rc.Process -> rc.node.Step(ctx, m)   raft.go:268-270
 *     rc.node.Ready()     -> wal.Save, raftStorage.Append, transport.Send,
 *                            publishEntries, rc.node.Advance      raft.go:227-235
 * A raftq_node does the same for G groups of which this process is peer slot
 * `self_peer`: every consensus decision (Step for every payload-free message
 * kind, Tick, the commit index) is made by the batched GPU engine
 * (raftq_step.h) on device-resident state; the host side below owns what the
 * reference's raftNode owns -- the log entries (raft.MemoryStorage), the
 * replication cursor (Progress.Next), the outbound message queues and the
 * commit channels -- and drives one Ready-loop iteration per
 * raftq_node_advance() call for all groups at once.
 *
 * Per group, towards the application, the contract of raftPipe / newRaftNode:
 *     ProposeC <- s        raftq_node_propose(n, group, s, len)
 *     <-CommitC            raftq_node_recv(n, group, ...): every replayed
 *                          entry, then the nil sentinel, then live entries
 *     Close() / ErrorC     raftq_node_close(n) returns the error (0 = nil)
 * Live entries are published when COMMITTED; the reference publishes
 * rd.Entries, i.e. appended entries (raft.go:231) -- upstream
 * raftexample publishes CommittedEntries.  As in the reference, HardState is
 * not restored by replay (raft.go:124): a restarted node begins
 * at term 0 with its log intact, unless raftq_node_set_hard_state is called.
 *
 * Transport is the caller's (the reference uses rafthttp, raft.go:170-186):
 * raftq_node_poll(n, to, ...) hands out the bytes addressed to peer `to`,
 * raftq_node_deliver() takes bytes a peer polled for this node.  The byte
 * format is this library's own framing of raftq_msg_t (below), NOT raftpb.
 *
 * Not built (same as raftq_step.h): snapshots / log compaction, conf changes,
 * the inflight window.  The log lives in host memory, like the reference's
 * raft.MemoryStorage (raft.go:70).
 *
 * Thread-safety: propose / deliver / tick / recv / poll / status from any
 * thread; advance from one thread at a time (the serveChannels goroutine).
 */
#ifndef RAFTQ_NODE_H
#define RAFTQ_NODE_H

#include <stdint.h>

#include "raftq_step.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct raftq_node raftq_node_t;

/* what raftq_node_recv delivered (same values as RAFTQ_PIPE_*) */
#define RAFTQ_NODE_ENTRY 0    /* a committed entry's payload (a *string on commitC) */
#define RAFTQ_NODE_SENTINEL 1 /* the nil that marks "commit channel is current" (raft.go:132) */
#define RAFTQ_NODE_CLOSED 2   /* commit channel closed */
#define RAFTQ_NODE_TIMEOUT 3  /* nothing within timeout_ms */

/* Wire frame (little endian), what raftq_node_poll emits and raftq_node_deliver parses:
 *   raftq_msg_t header (64 B): type = raftpb.MessageType (RAFTQ_MSG_* plus MsgProp = 2),
 *       from = sender's peer slot, _resv = number of entries that follow
 *   per entry: u64 term | u32 len | u32 0 | payload | zero padding to a multiple of 8
 * MsgApp: index / log_term = the entry preceding the first one carried, commit = leader's commit. */
#define RAFTQ_MSG_PROP 2

typedef struct raftq_node_status {
  uint64_t term, commit, last_index, applied;
  uint32_t lead; /* 0 = none, else peer slot + 1 */
  uint32_t vote; /* 0 = none, else peer slot + 1 */
  uint8_t role;  /* RAFTQ_ROLE_* */
  uint8_t _pad[7];
} raftq_node_status_t;

typedef struct raftq_node_stats {
  uint64_t turns;            /* raftq_node_advance calls that did work */
  uint64_t msgs_stepped;     /* messages that went through raftq_step_batch */
  uint64_t msgs_sent;        /* frames queued for peers */
  uint64_t entries_published;
  uint64_t hard_states;      /* HardState changes that a WAL would have had to persist (raft.go:228) */
  uint64_t proposals_dropped; /* proposals that met a group with no leader (etcd drops them) */
} raftq_node_stats_t;

int raftq_node_create(int device, uint64_t n_groups, uint32_t n_peers, uint32_t self_peer, raftq_node_t** out);
/* before start: the WAL's entries of one group, in order (raft.go:122-134 replayWAL) */
int raftq_node_replay(raftq_node_t* n, uint64_t group, const uint64_t* terms, const void* const* data,
                      const uint32_t* lens, uint64_t count);
/* before start, optional: restore HardState (the reference does not) */
int raftq_node_set_hard_state(raftq_node_t* n, uint64_t group, uint64_t term, uint32_t vote, uint64_t commit);
/* raft.Config{ElectionTick, HeartbeatTick} (raft.go:154-155: 10, 1) + seed of the randomised timeout */
int raftq_node_start(raftq_node_t* n, uint32_t election_tick, uint32_t heartbeat_tick, uint64_t seed);

int raftq_node_propose(raftq_node_t* n, uint64_t group, const void* data, uint32_t len);
int raftq_node_tick(raftq_node_t* n);
int raftq_node_deliver(raftq_node_t* n, const void* frames, uint64_t len);
/* one Ready-loop iteration for all groups; *n_published = entries put on commit channels */
int raftq_node_advance(raftq_node_t* n, uint64_t* n_published);
/* whole frames addressed to `to_peer`, at most cap bytes; *len = bytes written (0 = nothing queued) */
int raftq_node_poll(raftq_node_t* n, uint32_t to_peer, void* buf, uint64_t cap, uint64_t* len);

int raftq_node_recv(raftq_node_t* n, uint64_t group, int timeout_ms, void* buf, uint32_t cap, uint32_t* len,
                    int* kind);
int raftq_node_status(raftq_node_t* n, uint64_t group, raftq_node_status_t* st);
int raftq_node_stats(raftq_node_t* n, raftq_node_stats_t* st);
int raftq_node_entry(raftq_node_t* n, uint64_t group, uint64_t index, void* buf, uint32_t cap, uint32_t* len,
                     uint64_t* term);
/* the engine underneath (for bulk read-back in tests / tools); owned by the node */
raftq_t* raftq_node_engine(raftq_node_t* n);

int raftq_node_close(raftq_node_t* n);
int raftq_node_error(const raftq_node_t* n);
const char* raftq_node_last_error(const raftq_node_t* n);
void raftq_node_destroy(raftq_node_t* n);

#ifdef __cplusplus
}
#endif
#endif /* RAFTQ_NODE_H */
