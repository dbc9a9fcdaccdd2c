/*
 * raftq_oracle.h -- CPU oracle for the quorum hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library, and only as the checker / the timed CPU baseline.
 * The product (libraftq.so) never links or loads it.
 *
 * PARITY UNPINNED.  The arithmetic restated here lives in the reference's
 * un-vendored, un-pinned dependency github.com/coreos/etcd/raft (2015-era,
 * v2.2-v2.3 line by API shape).  Its source
 * is not in /root/reference, there is no Go toolchain, and the reference's
 * own tests (raftsql_test.go:92-171) hold no numeric vector for this path.
 * What pins this file instead: a second independent implementation
 * (rq_oracle_mci_count, the brute-force definition), a numpy third in
 * tests/, hand-written known-answer vectors and properties (tests/).
 */
#ifndef RAFTQ_ORACLE_H
#define RAFTQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#include "raftq_step.h" /* record layouts of the batched Step (types only: the oracle links nothing of the product) */
#ifdef __cplusplus
extern "C" {
#endif

#define RQ_MAX_PEERS 9

/* a5: etcd raft.q() -- quorum size over N voting peers */
int rq_oracle_quorum(int n);

/* a6: etcd raft.maybeCommit, first half -- gather the peers' Match into a
 * fresh slice, sort descending, take element q-1.  Sort-shaped on purpose. */
uint64_t rq_oracle_mci_sort(const uint64_t* match, int n);
/* independent restatement: the largest index stored on >= q peers */
uint64_t rq_oracle_mci_count(const uint64_t* match, int n);

/* raftLog.term(i) over a run-length log: runs r = 0..nruns-1 cover
 * [run_start[r], run_start[r+1]) with term run_term[r]; the last run ends at
 * last_index.  Index 0, indices below run_start[0] (compacted) and indices
 * above last_index yield 0 (etcd's zeroTermOnErrCompacted + out-of-range). */
uint64_t rq_oracle_log_term(const uint64_t* run_start, const uint64_t* run_term,
                            int nruns, uint64_t last_index, uint64_t i);

/* a7: etcd raftLog.maybeCommit(maxIndex, term) -- returns the new committed */
uint64_t rq_oracle_maybe_commit(uint64_t mci, uint64_t committed, int gated,
                                uint64_t term_of_mci, uint64_t cur_term);

/* a8: etcd raft.poll + the candidate's MsgVoteResp switch, on a snapshot:
 * 1 = won (granted >= q), 2 = lost (rejected >= q), 0 = pending. */
uint8_t rq_oracle_poll(const uint8_t* votes, int n);

/* ---- batched forms over G groups, SoA x[p*ld + g] ----------------------- */
/* ungated (gated=0) or compact-gated (gated=1, first_idx_cur_term[g]) */
uint64_t rq_oracle_commit_advance(const uint64_t* match, size_t ld, int n, size_t G,
                                  const uint64_t* committed, int gated,
                                  const uint64_t* first_idx_cur_term,
                                  uint64_t* committed_out);
/* full-log gated form: term looked up in each group's run-length log.
 * runs are CSR: group g owns runs [run_off[g], run_off[g+1]). */
uint64_t rq_oracle_commit_advance_log(const uint64_t* match, size_t ld, int n, size_t G,
                                      const uint64_t* committed, const uint64_t* cur_term,
                                      const uint64_t* run_off, const uint64_t* run_start,
                                      const uint64_t* run_term, const uint64_t* last_index,
                                      uint64_t* committed_out);
/* derive the compact encoding from the run-length log */
void rq_oracle_first_idx_cur_term(size_t G, const uint64_t* cur_term,
                                  const uint64_t* run_off, const uint64_t* run_start,
                                  const uint64_t* run_term, uint64_t* first_idx_out);
void rq_oracle_vote_tally(const uint8_t* votes, size_t ld, int n, size_t G,
                          uint8_t* outcome_out, uint64_t* n_won, uint64_t* n_lost);

/* sparse ingest restatements (Progress.maybeUpdate / poll's first-wins) */
void rq_oracle_apply_deltas(uint64_t* match, size_t ld, int n, size_t G,
                            const uint64_t* d_group, const uint32_t* d_peer,
                            const uint64_t* d_match, size_t nd);
void rq_oracle_apply_vote_deltas(uint8_t* votes, size_t ld, int n, size_t G,
                                 const uint64_t* d_group, const uint32_t* d_peer,
                                 const uint8_t* d_vote, size_t nd);

/* ---- batched Tick: etcd raft.tickElection / tickHeartbeat ----
 * role: 0 follower, 1 candidate, 2 leader.  action_out: 0 none, 1 MsgHup
 * (campaign), 2 MsgBeat (heartbeat).  The randomised election timeout uses a
 * counter-based stream instead of Go's math/rand (which cannot be reproduced
 * without the Go runtime): key = splitmix64 finaliser of (seed, tick_no), once
 * per tick; rnd = fmix32(group ^ key.lo) ^ key.hi (murmur3's 32-bit finaliser:
 * two 32-bit multiplies per group). */
uint64_t rq_oracle_tick_key(uint64_t seed, uint64_t tick_no);
uint32_t rq_oracle_tick_rand(uint64_t seed, uint64_t tick_no, uint64_t group);
void rq_oracle_tick(const uint8_t* role, uint32_t* elapsed /*in/out*/, size_t G, uint32_t election_tick,
                    uint32_t heartbeat_tick, uint64_t seed, uint64_t tick_no, uint8_t* action_out,
                    uint64_t* n_hup, uint64_t* n_beat);
/* becomeCandidate for the listed groups: role = candidate, every vote slot
 * cleared, the candidate's own slot granted, elapsed = 0. */
void rq_oracle_campaign(uint8_t* role, uint32_t* elapsed, uint8_t* votes, size_t ld, int n, size_t G,
                        const uint64_t* groups, size_t ng, uint32_t self_peer);

/* ---- batched Step (raftq_step_oracle.c) -------------------
 * One raft node's state for G groups, SoA, caller-owned arrays.  role/elapsed are
 * the arrays rq_oracle_tick works on; committed/first_idx/match/votes those of the
 * sweep oracles above.  vote / lead: 0 = None, else peer slot + 1. */
typedef struct rq_node_state {
  size_t G, ld; /* groups; row stride of match / votes */
  int n;        /* voting peers */
  uint32_t self; /* this node's peer slot */
  uint8_t* role;
  uint32_t* elapsed;
  uint64_t* term;
  uint32_t* vote;
  uint32_t* lead;
  uint64_t* last_index;
  uint64_t* last_term;
  uint64_t* committed;
  uint64_t* first_idx; /* compact current-term gate: first index of Term, 0 = none */
  uint64_t* match;     /* [n][ld] */
  uint8_t* votes;      /* [n][ld] */
} rq_node_state_t;
/* etcd raft.Step for every message, in order; out[i] answers msgs[i] */
void rq_oracle_step_batch(rq_node_state_t* s, const raftq_msg_t* msgs, size_t n, raftq_step_out_t* out);
void rq_oracle_apply_log_deltas(rq_node_state_t* s, const raftq_log_delta_t* d, size_t n,
                                uint64_t* committed_out /*[n]|NULL*/);

/* ---- wire / WAL codecs (raftq_wire_oracle.c, structs in include/raftq_wire.h) ----
 * raftpb.Message marshal / unmarshal in rafthttp frames, walpb.Record frames with the chained
 * CRC-32C.  Pinned against the protobuf runtime and RFC 3720 (see raftq_wire_oracle.c). */
struct raftq_wire_msg;
struct raftq_wire_ent;
struct raftq_wal_rec;
uint32_t rq_crc32c_update(uint32_t crc, const uint8_t* p, size_t n);       /* hash/crc32 Update, bitwise */
uint32_t rq_crc32c_update_table(uint32_t crc, const uint8_t* p, size_t n); /* same, table-driven */
void rq_wire_set_fast_crc(int on); /* WAL functions: 1 = table CRC (cpu_baseline timing), 0 = bitwise (default) */
uint32_t rq_crc32c_mulmod(uint32_t a, uint32_t b);
uint32_t rq_crc32c_xpow8(uint64_t n);
uint32_t rq_crc32c_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);
/* returns the bytes the encoding takes (also when that exceeds cap: nothing past cap is written) */
uint64_t rq_wire_encode(const struct raftq_wire_msg* msgs, uint64_t n, const struct raftq_wire_ent* ents,
                        const uint8_t* pool, uint8_t* out, uint64_t cap, uint64_t* frame_off /*[n+1]|NULL*/);
int rq_wire_decode(const uint8_t* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n,
                   struct raftq_wire_msg* msgs, struct raftq_wire_ent* ents /*|NULL*/, uint64_t ents_cap,
                   uint64_t* n_ents, uint64_t* n_bad);
int rq_wire_scan_frames(const uint8_t* buf, uint64_t nbytes, int big_endian, uint64_t* off, uint64_t cap,
                        uint64_t* n_frames, uint64_t* consumed);
uint64_t rq_wal_encode(const struct raftq_wal_rec* recs, uint64_t n, const uint8_t* pool, uint32_t prev_crc,
                       uint8_t* out, uint64_t cap, uint64_t* frame_off /*[n+1]|NULL*/, uint32_t* last_crc);
int rq_wal_decode(const uint8_t* bytes, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, uint32_t prev_crc,
                  struct raftq_wal_rec* recs, uint64_t* n_valid, uint32_t* last_crc);

/* ---- timed CPU baselines (bench.py cpu_baseline leg) -------------------- */
/* kind 0: reference-shaped loop (malloc N-slice, sort desc, index q-1, scan
 * votes); kind 1: tight selection network, no allocation.  Runs `sweeps`
 * passes over the G groups on `threads` pthreads (contiguous group ranges),
 * returns wall seconds; outputs land in committed_out / outcome_out. */
double rq_oracle_timed_sweeps(int kind, int threads, int sweeps,
                              const uint64_t* match, size_t ld, int n, size_t G,
                              const uint64_t* committed, int gated,
                              const uint64_t* first_idx_cur_term,
                              const uint8_t* votes, size_t ldv,
                              uint64_t* committed_out, uint8_t* outcome_out);

#ifdef __cplusplus
}
#endif
#endif
