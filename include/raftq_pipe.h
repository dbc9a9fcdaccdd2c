/*
 * raftq_pipe.h -- the multi-group propose -> commit pipeline.
 *
 * Host-side driver above raftq.h that keeps, per raft group, the surface of
 * the reference's raftPipe (raftpipe.go:3-17):
 *     ProposeC  chan<- string    ->  raftq_pipe_propose(p, group, data, len)
 *     CommitC   <-chan *string   ->  raftq_pipe_recv(p, group, ...)
 *     ErrorC    <-chan error     ->  raftq_pipe_close() return / raftq_pipe_error
 *     Close()   error            ->  raftq_pipe_close(p)
 * with the contract of newRaftNode (raft.go:57-62): all logged entries are
 * replayed on the commit channel, then a nil sentinel, then new entries; to
 * shut down, close the proposal side and read the error.
 *
 * What is different from the reference, on purpose:
 *   - one pipe drives G groups; the commit check of ALL groups runs as one
 *     GPU sweep per batching turn (raftq_cycle) instead of once per message
 *     inside raft.Node.Step (raft.go:268-270);
 *   - live entries are published when COMMITTED (the quorum index passed them
 *     and the current-term gate holds), not when appended -- the reference
 *     publishes rd.Entries (raft.go:231); upstream raftexample
 *     publishes CommittedEntries;
 *   - this node is the leader of every group it drives (peer slot 0); peer
 *     transport, WAL and elections are outside this path.  Outbound appends
 *     are exposed through raftq_pipe_take_appends for a transport to ship;
 *     inbound acks arrive through raftq_pipe_process_app_resp, the
 *     MsgAppResp half of rc.Process (raft.go:268-270).
 *
 * Written in C++ because the image has no Go toolchain; go/raftq/batcher.go is
 * the same loop as (uncompiled) Go source.  Thread-safety: propose /
 * process_app_resp / recv may be called from any thread; flush / close from one.
 */
#ifndef RAFTQ_PIPE_H
#define RAFTQ_PIPE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct raftq_pipe raftq_pipe_t;

/* what raftq_pipe_recv delivered */
#define RAFTQ_PIPE_ENTRY 0    /* a committed entry's payload (a *string) */
#define RAFTQ_PIPE_SENTINEL 1 /* the nil that marks "commit channel is current" */
#define RAFTQ_PIPE_CLOSED 2   /* commit channel closed (after raftq_pipe_close) */
#define RAFTQ_PIPE_TIMEOUT 3  /* nothing within timeout_ms */

/* an outbound MsgApp the transport should send to the followers of `group` */
typedef struct raftq_append {
  uint64_t group;
  uint64_t index;
  uint64_t term;
  uint32_t len; /* payload bytes; fetch with raftq_pipe_entry */
  uint32_t _pad;
} raftq_append_t;

/* NewRaftPipe for G groups x N peers on one GPU (this node = peer slot 0 of
 * every group).  Nothing is published until raftq_pipe_start. */
int raftq_pipe_create(int device, uint64_t n_groups, uint32_t n_peers, raftq_pipe_t** out);

/* replayWAL (raft.go:122-134): preload `n` logged entries of one group, in
 * log order, before start.  terms[i] must be non-decreasing and >= 1. */
int raftq_pipe_replay(raftq_pipe_t* p, uint64_t group, const uint64_t* terms, const void* const* data,
                      const uint32_t* lens, uint64_t n);

/* startRaft: publish every replayed entry of every group followed by the nil
 * sentinel; become leader of every group at term (last logged term + 1) and
 * append the leader's empty entry, as etcd's becomeLeader does.  With
 * background != 0 a batching thread runs raftq_pipe_flush whenever work is
 * pending (at most max_wait_us after the first pending message, or as soon as
 * max_batch messages are pending). */
int raftq_pipe_start(raftq_pipe_t* p, uint32_t max_batch, uint32_t max_wait_us, int background);

/* ProposeC <- data.  Empty payloads are legal and, like the reference's
 * publishEntries (raft.go:82-96), are never delivered on the commit side. */
int raftq_pipe_propose(raftq_pipe_t* p, uint64_t group, const void* data, uint32_t len);

/* rc.Process(ctx, MsgAppResp{From: from, Index: index}) for `group`.
 * from in 1..N-1 (0 is this node).  Acks past the leader's last index are
 * rejected with RAFTQ_EINVAL (a follower cannot hold what was never sent). */
int raftq_pipe_process_app_resp(raftq_pipe_t* p, uint64_t group, uint32_t from, uint64_t index);

/* one batching turn now: scatter pending acks, sweep all groups (gated),
 * publish the newly committed entries.  n_advanced may be NULL. */
int raftq_pipe_flush(raftq_pipe_t* p, uint64_t* n_advanced);

/* <-CommitC of `group`.  kind = RAFTQ_PIPE_*; for ENTRY, up to cap bytes are
 * copied and *len is the full payload length. */
int raftq_pipe_recv(raftq_pipe_t* p, uint64_t group, int timeout_ms, void* buf, uint32_t cap, uint32_t* len,
                    int* kind);

/* outbound appends since the last call (for the transport) */
int raftq_pipe_take_appends(raftq_pipe_t* p, raftq_append_t* out, uint64_t cap, uint64_t* n);
int raftq_pipe_entry(raftq_pipe_t* p, uint64_t group, uint64_t index, void* buf, uint32_t cap, uint32_t* len,
                     uint64_t* term);

/* introspection */
int raftq_pipe_last_index(raftq_pipe_t* p, uint64_t group, uint64_t* index);
int raftq_pipe_committed(raftq_pipe_t* p, uint64_t group, uint64_t* index);
int raftq_pipe_term(raftq_pipe_t* p, uint64_t group, uint64_t* term);

/* Close(): stop accepting proposals, stop the batching thread, close every
 * commit channel, and return the error state (0 = the nil error). */
int raftq_pipe_close(raftq_pipe_t* p);
/* the ErrorC value (0 while healthy) and its text */
int raftq_pipe_error(const raftq_pipe_t* p);
const char* raftq_pipe_last_error(const raftq_pipe_t* p);
void raftq_pipe_destroy(raftq_pipe_t* p);

#ifdef __cplusplus
}
#endif
#endif /* RAFTQ_PIPE_H */
