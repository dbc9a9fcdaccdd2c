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
 * raftq_node_deliver() takes bytes a peer polled for this node.  The bytes
 * are rafthttp message-stream frames -- u64 big-endian length | raftpb.Message,
 * what rc.transport.Send(rd.Messages) puts on a stream (raft.go:230; format
 * and the one `group` extension field in raftq_wire.h) -- marshalled and
 * unmarshalled for all groups at once by the GPU codecs, one
 * raftq_wire_encode / raftq_wire_decode per raftq_node_advance().
 *
 * WAL (opt-in, raftq_node_wal_enable): every advance() also produces what
 * rc.wal.Save(rd.HardState, rd.Entries) (raft.go:228) would have appended --
 * walpb.Record frames with the segment's CRC-32C chain, every touched group's
 * entries and HardState in one raftq_wal_encode -- for the caller to write and
 * fsync BEFORE it transmits that turn's raftq_node_poll bytes (the reference's
 * save-then-send order, raft.go:228-230).  raftq_node_replay_wal() is
 * replayWAL (raft.go:122-134) from such bytes.
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

#include "raftq_wire.h" /* (raftq_step.h with it) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct raftq_node raftq_node_t;

/* what raftq_node_recv delivered (same values as RAFTQ_PIPE_*) */
#define RAFTQ_NODE_ENTRY 0    /* a committed entry's payload (a *string on commitC) */
#define RAFTQ_NODE_SENTINEL 1 /* the nil that marks "commit channel is current" (raft.go:132) */
#define RAFTQ_NODE_CLOSED 2   /* commit channel closed */
#define RAFTQ_NODE_TIMEOUT 3  /* nothing within timeout_ms */

/* On the wire: raftpb.Message frames (raftq_wire.h).  type = raftpb.MessageType (RAFTQ_MSG_* plus
 * MsgProp = 2, a follower forwarding a proposal to its leader); to / from = raft IDs (peer slot + 1).
 * MsgApp: index / logTerm = the entry preceding the first one carried, commit = leader's commit;
 * entries carry their own Index and Term.  Frames that do not parse, are not addressed to this node,
 * or are of a kind a peer never sends (MsgHup, MsgBeat, MsgSnap, unknown) are dropped and counted. */
/* (RAFTQ_MSG_PROP = 2: raftq_wire.h) */

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
  uint64_t frames_dropped;    /* inbound frames that did not parse / were not for this node */
  uint64_t wal_records;       /* walpb.Records produced (raftq_node_wal_enable) */
  uint64_t msgs_built_on_device; /* of msgs_sent: MsgApps of proposals that raftq_propose_frames built in HBM (round 6) */
} raftq_node_stats_t;

int raftq_node_create(int device, uint64_t n_groups, uint32_t n_peers, uint32_t self_peer, raftq_node_t** out);
/* before start: the WAL's entries of one group, in order (raft.go:122-134 replayWAL) */
int raftq_node_replay(raftq_node_t* n, uint64_t group, const uint64_t* terms, const void* const* data,
                      const uint32_t* lens, uint64_t count);
/* before start, optional: restore HardState (the reference does not) */
int raftq_node_set_hard_state(raftq_node_t* n, uint64_t group, uint64_t term, uint32_t vote, uint64_t commit);
/* before start: replayWAL (raft.go:122-134) from WAL bytes as raftq_node_wal_poll produced them (whole
 * frames; a torn tail is ignored).  Every record's CRC is checked against the chain: a mismatch or a
 * record that does not parse is RAFTQ_EINVAL (the reference log.Fatalf's, raft.go:126).  Entries go to
 * their groups' logs; HardState records are restored only if restore_hard_state != 0 (the reference
 * discards them).  A node replayed this way keeps appending to the same CRC chain. */
int raftq_node_replay_wal(raftq_node_t* n, const void* wal, uint64_t len, int restore_hard_state, uint64_t* n_records);
/* before start: produce WAL bytes from now on (see above) */
int raftq_node_wal_enable(raftq_node_t* n);
/* raft.Config{ElectionTick, HeartbeatTick} (raft.go:154-155: 10, 1) + seed of the randomised timeout */
int raftq_node_start(raftq_node_t* n, uint32_t election_tick, uint32_t heartbeat_tick, uint64_t seed);

int raftq_node_propose(raftq_node_t* n, uint64_t group, const void* data, uint32_t len);
/* the same for k proposals at once: proposal i is blob[offsets[i], offsets[i + 1]) for group groups[i] (one lock
 * acquisition instead of k -- the batching goroutine of a G-group server drains its ProposeC's into one call) */
int raftq_node_propose_batch(raftq_node_t* n, const uint64_t* groups, const uint64_t* offsets /*[k+1]*/, const void* blob,
                             uint64_t k);
int raftq_node_tick(raftq_node_t* n);
/* raft.Node.Campaign(ctx) for k groups: each gets a local MsgHup at the next raftq_node_advance(), before the messages
 * received since and whatever the election timers raise.  The reference never calls it (raft.go uses Propose / Tick /
 * Ready / Advance / Step / Stop); it is here so that message-driven scenarios -- etcd's own raft_test.go network tests
 * (tests/test_node_scenarios_gpu.py) -- can elect a chosen node without waiting for a randomised timer. */
int raftq_node_campaign(raftq_node_t* n, const uint64_t* groups, uint64_t k);
int raftq_node_deliver(raftq_node_t* n, const void* frames, uint64_t len);
/* one Ready-loop iteration for all groups; *n_published = entries put on commit channels */
int raftq_node_advance(raftq_node_t* n, uint64_t* n_published);
/* whole frames addressed to `to_peer`, at most cap bytes; *len = bytes written (0 = nothing queued) */
int raftq_node_poll(raftq_node_t* n, uint32_t to_peer, void* buf, uint64_t cap, uint64_t* len);

/* In-process transport: everything `from` has queued for peer `to_peer` goes straight to `to` (raftq_node_poll +
 * raftq_node_deliver without the caller's buffer in between) -- how several nodes of one process exchange their frames
 * (the reference's tests run three nodes in one process over loopback TCP, raftsql_test.go:11-35; here the bytes never
 * leave the process).  to == NULL: the frames are dropped (a lost transfer, a partitioned or stopped peer).
 * *moved (may be NULL) = bytes taken off `from`'s queue.  The two nodes' locks are never held together. */
int raftq_node_forward(raftq_node_t* from, uint32_t to_peer, raftq_node_t* to, uint64_t* moved);

/* The nodes of ONE process turned in lock-step, each on a thread of its own (thread p pinned to cpus[p] when cpus != NULL
 * and cpus[p] >= 0): what the reference's tests do with three raftNodes in one process (raftsql_test.go:11-35).
 * nodes[p] must be peer slot p of an n-peer cluster (n <= 32) or NULL (a stopped node, never live); the crank does not
 * own them and must be destroyed first.
 * raftq_crank_step = for every node p with bit p of live_mask set: raftq_node_tick (when tick != 0) and
 * raftq_node_advance, all at once; then, all at once per ADDRESSEE q, raftq_node_forward(p -> q) for every live sender p
 * in slot order starting at first_sender (mod n) -- dropped instead when q is not live or lost[q * n + p] != 0 (lost may be
 * NULL).  So what a node receives in a step, and in which order, does not depend on thread timing; a caller that passes the
 * step number as first_sender gives no slot the first word every time (two candidates of one tick: whose MsgVote a third
 * node reads first decides the election -- with a fixed order the lowest slot would win every tie).  published[p] (may be NULL) = entries node p put on
 * its commit channels; node_rc[p] (may be NULL) = node p's first error; returns the first non-zero of those.  One
 * raftq_crank_step at a time per crank (the caller's thread waits in it); the nodes' other entry points (propose, recv,
 * status ...) stay callable from any thread meanwhile. */
typedef struct raftq_crank raftq_crank_t;
int raftq_crank_create(raftq_node_t* const* nodes, uint32_t n, const int* cpus /*[n]|NULL*/, raftq_crank_t** out);
int raftq_crank_step(raftq_crank_t* c, uint32_t live_mask, int tick, const uint8_t* lost /*[n*n]|NULL*/, uint32_t first_sender,
                     uint64_t* published /*[n]|NULL*/, int* node_rc /*[n]|NULL*/);
/* wall time of all steps so far, by half: every node's turn (the slowest decides), the transport */
void raftq_crank_seconds(const raftq_crank_t* c, double* turns, double* transport);
void raftq_crank_destroy(raftq_crank_t* c);

/* The SHARDS of one node.  One raftq_node handle turns its groups on ONE host thread, and at tens of thousands of groups a
 * turn is mostly host work (Progress, logs, queues: 3.5 of 4.0 ms at 32,768 groups).  Groups are independent -- the reference
 * runs one raftNode goroutine per group (raft.go:204-246) -- so a process may split a node's groups over K handles (the same
 * n_peers and self_peer; group g of the node = group g mod G/K of shard g / (G/K), or any other split the transport knows)
 * and turn them all at once: raftq_shards_turn = raftq_node_tick (when tick != 0) + raftq_node_advance of every shard, each
 * on a thread of its own (thread i pinned to cpus[i] when cpus != NULL and cpus[i] >= 0), one call.  Everything else stays
 * per shard handle: raftq_node_deliver / _propose / _poll / _wal_poll / _recv (a transport keeps a stream per shard and
 * peer, so no frame has to be looked into to find its shard).  published[i] / shard_rc[i] (may be NULL) as raftq_crank_step's.
 * The set does not own the shards; destroy it (raftq_crank_destroy) before them.  k <= 32. */
int raftq_shards_create(raftq_node_t* const* shards, uint32_t k, const int* cpus /*[k]|NULL*/, raftq_crank_t** out);
int raftq_shards_turn(raftq_crank_t* set, int tick, uint64_t* published /*[k]|NULL*/, int* shard_rc /*[k]|NULL*/);

/* whole WAL frames produced so far, at most cap bytes; *len = bytes written (0 = nothing pending) */
int raftq_node_wal_poll(raftq_node_t* n, void* buf, uint64_t cap, uint64_t* len);

int raftq_node_recv(raftq_node_t* n, uint64_t group, int timeout_ms, void* buf, uint32_t cap, uint32_t* len,
                    int* kind);
int raftq_node_status(raftq_node_t* n, uint64_t group, raftq_node_status_t* st);
/* raft.Node.Status() for groups [first_group, first_group + count) in one call (st[count]): what a G-group
 * host polls instead of G calls -- leader discovery over 32,768 groups x 3 nodes was 1.4 s of ctypes calls */
int raftq_node_status_batch(raftq_node_t* n, uint64_t first_group, uint64_t count, raftq_node_status_t* st);
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
