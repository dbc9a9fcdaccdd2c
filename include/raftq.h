/*
 * raftq.h -- C-ABI of the MI355X batched multi-raft quorum engine.
 *
 * This is the drop-in boundary for ONE hot path of chzchzchz/raftsql: the
 * per-group quorum arithmetic that the reference reaches through
 *     rc.node.Step     (raft.go:268-270, inbound MsgAppResp / MsgVoteResp)
 *     rc.node.Propose  (raft.go:211-215, leader-local append)
 *     rc.node.Tick     (raft.go:223-224)
 * and whose result the reference consumes from rc.node.Ready() (raft.go:227,
 * HardState.Commit / CommittedEntries).  The arithmetic itself lives in the
 * un-vendored dependency github.com/coreos/etcd/raft (SURVEY.md section 0,
 * F1/F2): raft.maybeCommit, raftLog.maybeCommit, raft.q and raft.poll.  The
 * reference has no FFI for it, so the entry points below are what a cgo shim
 * in a G-group raftsql would bind (see INTEGRATION.md for that shim).
 *
 * Conventions
 *   - every function returns RAFTQ_OK (0) or a negative RAFTQ_E* code; none
 *     throws, aborts or calls exit().  raftq_last_error() gives the text.
 *   - a handle owns device-resident SoA state for G groups x N peers on one
 *     GPU.  One handle is NOT thread-safe (the caller serialises, mirroring
 *     the one-goroutine rule of raft.Node); different handles are independent.
 *   - host buffers are caller-owned and are only read/written during the
 *     call; no pointer is retained after return (cgo pointer rule).
 *   - host-side matrices are peer-major and dense:  x[p * G + g].
 *   - vote encoding (uint8): 0 = no response yet, 1 = granted, 2 = rejected.
 *     Any other byte value counts as "no response".
 *   - outcome encoding (uint8): 0 = pending, 1 = won, 2 = lost.
 *   - log indices and terms are uint64, as raftpb's.  Index 0 is the dummy
 *     entry: it carries term 0 and never commits.
 */
#ifndef RAFTQ_H
#define RAFTQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAFTQ_ABI_VERSION 1
#define RAFTQ_MAX_PEERS 9 /* north_star: selection network for N <= 9 */

/* return codes */
#define RAFTQ_OK 0
#define RAFTQ_EINVAL (-1)  /* bad argument (NULL, N out of range, G == 0 ...) */
#define RAFTQ_ENOMEM (-2)  /* device or pinned-host allocation failed */
#define RAFTQ_EHIP (-3)    /* a HIP runtime call failed; see raftq_last_error */
#define RAFTQ_ESTATE (-4)  /* call sequence error (e.g. gated sweep, no terms) */
#define RAFTQ_ENODEV (-5)  /* no usable GPU / device index out of range */

/* sweep flags (raftq_step_async) */
#define RAFTQ_SWEEP_COMMIT 0x01u /* commit-advance: etcd raft.maybeCommit */
#define RAFTQ_SWEEP_GATED 0x02u  /* + raftLog.maybeCommit's current-term gate */
#define RAFTQ_SWEEP_VOTES 0x04u  /* RequestVote tally: etcd raft.poll */
#define RAFTQ_SWEEP_NO_ADOPT 0x08u /* evaluate into the shadow commit buffer
                                      but do not make it current ("what-if") */
#define RAFTQ_SWEEP_LDS 0x10u /* A/B variant: LDS-staged odd-even transposition
                                 sort instead of the in-register network */
#define RAFTQ_SWEEP_CHANGED 0x20u /* also emit the changed-groups bitmap that
                                     raftq_collect_changed() compacts */
/* cache policy of the sweep's loads/stores.  Default (neither bit): chosen by
 * the handle's footprint -- state that fits the 256 MiB Infinity Cache is
 * swept with normal (cached) accesses so the next sweep hits on-die, larger
 * state is streamed non-temporally.  A caller that knows better (e.g. many
 * handles swept round-robin, so none stays cached) forces one. */
#define RAFTQ_SWEEP_STREAM 0x40u /* force non-temporal streaming accesses */
#define RAFTQ_SWEEP_CACHED 0x80u /* force normal cached accesses */

/* raftq_cycle / raftq_cycle_packed only: the caller vouches for the ranges of its records (a driver that built them
 * itself), so the library validates and scatters the match deltas in ONE pass straight from the batch.  A record of
 * EITHER kind that is out of range after all is dropped on its own -- what a trusted turn applies never depends on
 * another record of the batch: every other match and vote record is applied, the sweep is adopted, every output is
 * filled in, and the call returns RAFTQ_EINVAL with "that record was dropped, every other record ... was applied" in
 * raftq_last_error().  Without the flag a turn is all-or-nothing. */
#define RAFTQ_CYCLE_TRUSTED 0x100u

typedef struct raftq raftq_t;

/* per-sweep tallies (reduced from per-wave partials when waited for) */
typedef struct raftq_counts {
  uint64_t n_changed; /* groups whose commit index advanced   */
  uint64_t n_won;     /* groups whose candidate reached quorum */
  uint64_t n_lost;    /* groups whose candidate was rejected by a quorum */
} raftq_counts_t;

/* one MsgAppResp-shaped update: peer `peer` of group `group` now matches
 * `match` (raft.go:268-270 -> Step -> Progress.maybeUpdate: only increases) */
typedef struct raftq_delta {
  uint64_t group;
  uint64_t match;
  uint32_t peer;
  uint32_t _pad;
} raftq_delta_t;

/* one MsgVoteResp-shaped update (first response from a peer wins, as poll) */
typedef struct raftq_vote_delta {
  uint64_t group;
  uint32_t peer;
  uint8_t vote; /* 1 granted, 2 rejected */
  uint8_t _pad[3];
} raftq_vote_delta_t;

/* one group whose leader entered a new term (or appended the first entry of
 * its term): updates the gate of raftLog.maybeCommit for that group */
typedef struct raftq_term_delta {
  uint64_t group;
  uint64_t cur_term;
  uint64_t first_idx_cur_term; /* 0 = no entry of cur_term in the log yet */
} raftq_term_delta_t;

/* one advanced group: what would surface in Ready.HardState.Commit */
typedef struct raftq_advance {
  uint64_t group;
  uint64_t old_commit;
  uint64_t new_commit;
} raftq_advance_t;

/* the same two records in 16 bytes, for handles of at most 2^32 groups (raftq_cycle_packed): the batching turn
 * is bound by PCIe both ways, and these are a third fewer bytes each way */
typedef struct raftq_delta16 {
  uint64_t match;
  uint32_t group;
  uint32_t peer;
} raftq_delta16_t;
typedef struct raftq_advance16 {
  uint64_t new_commit;
  uint32_t group;
  uint32_t advanced_by; /* new_commit - old_commit, saturated at 2^32 - 1 (then old_commit is not recoverable) */
} raftq_advance16_t;

/* ---- library / device ------------------------------------------------- */
int raftq_abi_version(void);
int raftq_device_count(int* n);
/* quorum size q = floor(N/2)+1  (etcd raft.q) -- host-side helper */
uint32_t raftq_quorum(uint32_t n_peers);

/* ---- handle lifetime --------------------------------------------------- */
/* allocates the padded SoA state for G groups x N peers in HBM on `device`;
 * everything starts zeroed (match 0, committed 0, no votes, no terms).
 * 1 <= n_groups <= RAFTQ_MAX_GROUPS (RAFTQ_EINVAL above): the bound is what the
 * parity suite has compared against the oracle on one handle (2^29 + 70001 groups,
 * tests/test_envelope_gpu.py), rounded up to the next power of two -- a larger
 * population is several handles (one sweep set), which is also how it shards. */
#define RAFTQ_MAX_GROUPS (1ull << 30)
int raftq_create(int device, uint64_t n_groups, uint32_t n_peers, raftq_t** out);
void raftq_destroy(raftq_t* h);
uint64_t raftq_groups(const raftq_t* h);
uint32_t raftq_peers(const raftq_t* h);
const char* raftq_last_error(const raftq_t* h); /* h may be NULL: global */

/* all handles default to their own non-blocking stream.  A caller that owns
 * streams (torch, a Go batching goroutine with one stream per device) can
 * install one; `stream` is a hipStream_t passed as void*. */
int raftq_set_stream(raftq_t* h, void* stream);
void* raftq_get_stream(const raftq_t* h);

/* ---- bulk load of resident state (host -> HBM) ------------------------- */
int raftq_load_match(raftq_t* h, const uint64_t* match /*[N][G]*/,
                     const uint64_t* committed /*[G]*/);
/* first_idx_cur_term[g] = first log index whose entry has term cur_term[g],
 * or 0 when the leader's log holds no entry of its current term yet. */
int raftq_load_terms(raftq_t* h, const uint64_t* cur_term /*[G]*/,
                     const uint64_t* first_idx_cur_term /*[G]*/);
int raftq_load_votes(raftq_t* h, const uint8_t* votes /*[N][G]*/);

/* ---- sparse ingest ----------------------------------- */
int raftq_apply_deltas(raftq_t* h, const raftq_delta_t* d, uint64_t n);
int raftq_apply_vote_deltas(raftq_t* h, const raftq_vote_delta_t* d, uint64_t n);
/* later records of the same group win (applied in order on the host side) */
int raftq_apply_term_deltas(raftq_t* h, const raftq_term_delta_t* d, uint64_t n);

/* ---- the sweep ---------------------------------------------------------- */
/* enqueue one pass over all G groups on the handle's stream. */
int raftq_step_async(raftq_t* h, unsigned flags);
/* block until everything enqueued on the handle has finished; if `counts` is
 * not NULL, fill it with the tallies of the most recent sweep. */
int raftq_wait(raftq_t* h, raftq_counts_t* counts);

/* synchronous conveniences = step_async + wait + optional read-back */
int raftq_commit_advance(raftq_t* h, int gated, uint64_t* committed_out /*[G]|NULL*/,
                         uint64_t* n_changed /*|NULL*/);
int raftq_vote_tally(raftq_t* h, uint8_t* outcome_out /*[G]|NULL*/,
                     raftq_counts_t* counts /*|NULL*/);

/* ---- read-back ---------------------------------------------------------- */
int raftq_read_committed(raftq_t* h, uint64_t* committed_out /*[G]*/);
int raftq_read_outcome(raftq_t* h, uint8_t* outcome_out /*[G]*/);
int raftq_read_match(raftq_t* h, uint64_t* match_out /*[N][G]*/);
int raftq_read_votes(raftq_t* h, uint8_t* votes_out /*[N][G]*/);
/* compacted list of the groups the last RAFTQ_SWEEP_CHANGED sweep advanced,
 * in ascending group order; returns the count in *n (<= cap entries stored). */
int raftq_collect_changed(raftq_t* h, raftq_advance_t* out, uint64_t cap, uint64_t* n);

/* ---- batched Tick --------------------------------------
 * rc.node.Tick() (raft.go:223-224) for every group at once: etcd's
 * tickHeartbeat for leaders, tickElection + isElectionTimeout for the rest.
 * role: 0 follower, 1 candidate, 2 leader; action: 0 none, 1 MsgHup, 2 MsgBeat.
 * Timers default to the reference's ElectionTick 10 / HeartbeatTick 1
 * (raft.go:154-155).  The randomised election timeout draws from a
 * counter-based stream, not Go's math/rand (which cannot be reproduced
 * without the Go runtime).  The stream, exactly (round 6):
 *   key = fin64((seed ^ tick_no * 0xD1B54A32D192ED03) + 0x9E3779B97F4A7C15)   once per tick,
 *         fin64(z): z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9; z = (z ^ z >> 27) * 0x94D049BB133111EB; z ^ z >> 31
 *   rnd = fin32(lo32(group) ^ hi32(group) ^ lo32(key)) ^ hi32(key)             per group,
 *         fin32(x): x ^= x >> 16; x *= 0x85EBCA6B; x ^= x >> 13; x *= 0xC2B2AE35; x ^ x >> 16
 * and a timer fires when elapsed - election_tick > rnd % election_tick
 * (etcd's isElectionTimeout with its rand.Int() replaced by rnd). */
#define RAFTQ_ROLE_FOLLOWER 0
#define RAFTQ_ROLE_CANDIDATE 1
#define RAFTQ_ROLE_LEADER 2
typedef struct raftq_tick_counts {
  uint64_t n_hup;  /* groups whose election timer fired: they campaign */
  uint64_t n_beat; /* leader groups due a heartbeat */
} raftq_tick_counts_t;
int raftq_set_timers(raftq_t* h, uint32_t election_tick, uint32_t heartbeat_tick, uint64_t seed);
int raftq_load_roles(raftq_t* h, const uint8_t* role /*[G]*/, const uint32_t* elapsed /*[G]|NULL = 0*/);
/* one Tick for every group; synchronous when counts != NULL, else enqueued */
int raftq_tick(raftq_t* h, raftq_tick_counts_t* counts);
/* any of the outputs may be NULL */
int raftq_read_tick(raftq_t* h, uint8_t* action /*[G]*/, uint32_t* elapsed /*[G]*/, uint8_t* role /*[G]*/);
/* ascending list of the groups the last raftq_tick sent MsgHup to */
int raftq_collect_hups(raftq_t* h, uint64_t* groups, uint64_t cap, uint64_t* n);
/* ascending list of the leader groups the last raftq_tick sent MsgBeat to: they owe their followers a heartbeat
 * round (etcd tickHeartbeat -> Step(MsgBeat) -> bcastHeartbeat, reached from rc.node.Tick(), raft.go:223-224) */
int raftq_collect_beats(raftq_t* h, uint64_t* groups, uint64_t cap, uint64_t* n);
/* raftq_tick + both lists in one call: two launches and ONE wait (tick, then one kernel that ranks the MsgHup and the
 * MsgBeat bitmaps into two ascending lists), where raftq_tick + raftq_collect_hups + raftq_collect_beats are five
 * launches and two waits.  *n_hup / *n_beat receive the full counts even when they exceed the caps. */
int raftq_tick_collect(raftq_t* h, uint64_t* hups, uint64_t hup_cap, uint64_t* n_hup, uint64_t* beats, uint64_t beat_cap,
                       uint64_t* n_beat);

/* The same Tick + lists as they are meant to be read by a batching host: 4-byte group ids (a handle holds at most 2^30
 * groups) LEFT IN PLACE in page-locked memory -- nothing is copied into caller arrays -- and, with RAFTQ_TICK_BEAT_BITMAP,
 * the MsgBeat groups as a bitmap in group order (bit g % 64 of word g / 64) instead of a list: with HeartbeatTick 1
 * (raft.go:155) every leader owes a heartbeat on every tick, so the beat list IS the leader set, tick after tick.
 * n_hup / n_beat are the totals; at most hup_cap / beat_cap ids are listed (beat_cap is ignored with the bitmap).  Three
 * launches and one wait on the turn's completion word.  Stands in for rc.node.Tick() of every group (raft.go:223-224). */
#define RAFTQ_TICK_BEAT_BITMAP 1u
int raftq_tick_collect_lists(raftq_t* h, unsigned flags, uint64_t hup_cap, uint64_t beat_cap, uint64_t* n_hup, uint64_t* n_beat);
/* what the last raftq_tick_collect_lists left: ascending ids (valid until the next call of either on this handle); any
 * pointer argument may be NULL.  With the bitmap: *beats = NULL, *n_beats = 0, *bitmap_words = ceil(G / 64). */
int raftq_last_tick_lists(raftq_t* h, const uint32_t** hups, uint64_t* n_hups, const uint32_t** beats, uint64_t* n_beats,
                          const uint64_t** beat_bitmap, uint64_t* bitmap_words);
/* becomeCandidate for `n` distinct groups: role = candidate, elapsed = 0, votes
 * cleared, the candidate's own slot (`self_peer`) granted.  Term bookkeeping is
 * the caller's (raftq_apply_term_deltas). */
int raftq_campaign(raftq_t* h, const uint64_t* groups, uint64_t n, uint32_t self_peer);

/* ---- one batching iteration ----------------------------
 * What the single batching goroutine does per turn, fused into one call with
 * ONE host/device sync: scatter the MsgAppResp / MsgVoteResp deltas, sweep all
 * groups with `flags`, and return the compacted list of advanced groups (the
 * batched Ready.HardState.Commit).  Any of the arrays may be NULL with length 0;
 * `n_advanced` receives the full count even when it exceeds `cap`. */
int raftq_cycle(raftq_t* h, const raftq_delta_t* deltas, uint64_t n_deltas,
                const raftq_vote_delta_t* vote_deltas, uint64_t n_vote_deltas, unsigned flags,
                raftq_advance_t* advances_out, uint64_t cap, uint64_t* n_advanced,
                raftq_counts_t* counts);

/* zero-copy variants: raftq_stage returns the handle's ack buffer with room for the
 * given counts -- fine-grained DEVICE memory when the host can address it (large BAR:
 * the handlers' stores land in HBM as the acks arrive and the turn never pulls them
 * over PCIe), pinned device-visible host memory otherwise (or with RAFTQ_STAGE=host).
 * WRITE-ONLY for the host: reads of device memory over the BAR are uncached and slow.
 * Fill it and pass the SAME pointers to raftq_cycle and no staging copy is made.  They stay valid until the next
 * raftq_stage / raftq_apply_* / raftq_destroy on the handle.  With
 * advances_out == NULL and cap > 0, raftq_cycle leaves the advance list in
 * pinned memory; raftq_last_advances returns it (valid until the next
 * raftq_cycle / raftq_collect_changed). */
int raftq_stage(raftq_t* h, uint64_t n_deltas, uint64_t n_vote_deltas, raftq_delta_t** deltas,
                raftq_vote_delta_t** vote_deltas);
int raftq_last_advances(raftq_t* h, const raftq_advance_t** list, uint64_t* n_listed);

/* The batching turn -- one iteration of the Ready loop (raft.go:227-235) for every group -- with the 16-byte
 * records (handles of at most 2^32 groups).  Same semantics as raftq_cycle /
 * raftq_stage / raftq_last_advances; match deltas arrive as raftq_delta16_t, the advance list leaves as
 * raftq_advance16_t.  Without RAFTQ_CYCLE_TRUSTED a turn is all-or-nothing in either layout: one out-of-range record
 * of either kind and no record of the call is applied, the sweep it ran is not adopted, RAFTQ_EINVAL (with the flag:
 * see RAFTQ_CYCLE_TRUSTED above).  Arrays passed from the handle's ack buffer must have been staged with THIS call's
 * counts and layout (raftq_stage* with the same n_deltas / n_vote_deltas): RAFTQ_EINVAL otherwise. */
int raftq_cycle_packed(raftq_t* h, const raftq_delta16_t* deltas, uint64_t n_deltas,
                       const raftq_vote_delta_t* vote_deltas, uint64_t n_vote_deltas, unsigned flags,
                       raftq_advance16_t* advances_out, uint64_t cap, uint64_t* n_advanced,
                       raftq_counts_t* counts);
int raftq_stage_packed(raftq_t* h, uint64_t n_deltas, uint64_t n_vote_deltas, raftq_delta16_t** deltas,
                       raftq_vote_delta_t** vote_deltas);
int raftq_last_advances_packed(raftq_t* h, const raftq_advance16_t** list, uint64_t* n_listed);

/* RAFTQ_CYCLE_SEGMENTED on raftq_cycle_packed (advances_out == NULL, counts == NULL): the advance list may be left in SEGMENTS --
 * the turn's sweep writes the records of every tile of 1,024 groups itself, ascending, into that tile's own stretch of the
 * pinned list, and the compaction pass (a kernel, a boundary, 13 us of a 47 us turn) does not run: two kernels and the
 * completion word per turn instead of four.  Read them with raftq_last_advance_segments: segment s holds counts[s] records
 * at recs + s * stride, and walking the segments in order IS the ascending list raftq_last_advances_packed would have
 * returned (n_advanced is the same total).  A turn that cannot take that form (a vote tally or counts asked for, the list
 * copied out, the A/B sweep, a handle of more than 4M groups -- the pinned list has a slot per group) produces the contiguous
 * list as always, `cap` records of it at most, and raftq_last_advance_segments presents it as ONE segment: a consumer that
 * passes the flag reads segments, whatever happened (and compares counts[0] with n_advanced when there is one).  Valid until the next raftq_cycle* /
 * raftq_collect_changed on the handle. */
#define RAFTQ_CYCLE_SEGMENTED 0x200u
int raftq_last_advance_segments(raftq_t* h, const raftq_advance16_t** recs, const uint32_t** counts, uint32_t* n_segments,
                                uint64_t* stride);

/* ---- sweep sets: many handles, one dispatch --------------------------------
 * A host that runs more groups than it wants in one handle (several tenants, several
 * shards of one keyspace, 1M-group batches of a larger population) sweeps them together:
 * the G-fold Ready loop of raft.go:220-245 once more over K handles.  A set takes K
 * handles of one shape (same device, peer count and padded group count) and evaluates
 * them in ONE kernel launch (grid = tiles x K, the members' array pointers come from a
 * device-resident table), so the fixed cost of a launch boundary -- about a tenth of a
 * 1M x 5 sweep on MI355X -- is paid once per set instead of once per member.  Results
 * are exactly those of raftq_step_async on every member, and every per-handle call
 * (raftq_wait counts, raftq_read_*, raftq_collect_changed, raftq_apply_*) keeps working
 * on a member between set sweeps.
 *   - raftq_set_create re-homes every member onto the set's own stream (so member calls
 *     and set sweeps stay ordered); raftq_set_stream on a member is refused meanwhile;
 *     raftq_set_destroy gives every member a fresh stream back.  Destroy the set before
 *     its members (a member destroyed under a set makes the set refuse further calls).
 *   - flags are raftq_step_async's (RAFTQ_SWEEP_LDS excepted), applied to every member;
 *     a gated sweep needs terms loaded on every member.
 *   - like a handle, a set is not thread-safe, and its members must not be used from
 *     another thread while the set is. */
typedef struct raftq_set raftq_set_t;
#define RAFTQ_SET_GRID 0       /* one K-deep grid: blockIdx.y = member (default) */
#define RAFTQ_SET_PERSISTENT 1 /* resident workgroups walk all K x tiles, next tile's loads in flight */
int raftq_set_create(raftq_t* const* handles, uint32_t n, raftq_set_t** out);
void raftq_set_destroy(raftq_set_t* s);
uint32_t raftq_set_size(const raftq_set_t* s);
const char* raftq_set_last_error(const raftq_set_t* s); /* s may be NULL: global */
void* raftq_set_get_stream(const raftq_set_t* s);
/* launch shape of the set's sweeps; persist_workgroups 0 keeps the current / default count */
int raftq_set_mode(raftq_set_t* s, int mode, uint32_t persist_workgroups);
/* enqueue one pass over every member on the set's stream */
int raftq_set_sweep_async(raftq_set_t* s, unsigned flags);
/* one Tick of every group of every member as ONE dispatch, enqueued on the set's stream; each member is left exactly as
 * raftq_tick(member, NULL) would leave it (raftq_collect_hups / _beats / raftq_read_tick per member afterwards) */
int raftq_set_tick(raftq_set_t* s);
/* block until the set's stream is idle; tallies of each member's most recent sweep into
 * per_member[raftq_set_size] and / or their sum into *total (either may be NULL) */
int raftq_set_wait(raftq_set_t* s, raftq_counts_t* per_member, raftq_counts_t* total);
int raftq_set_timer_begin(raftq_set_t* s);
int raftq_set_timer_end(raftq_set_t* s, float* elapsed_ms);
/* the same K sweeps as K launches (raftq_step_async on every handle in turn, each on its
 * own stream): the host loop of a caller without a set, in one call.  Distinct handles that
 * share ONE stream (members of a set) are alternated between that stream and an auxiliary
 * one, forked and joined inside the call, so that consecutive launches overlap their
 * boundaries; everything is ordered behind / in front of the shared stream as before. */
int raftq_sweep_many_async(raftq_t* const* handles, uint32_t n, unsigned flags);
/* device-to-device copy of the quorum state (match, commit index, term gate, votes) of
 * `src` into `dst` (same device, groups and peers): fork a population without a host
 * round trip.  Tick / Step node state is not copied. */
int raftq_clone_state(raftq_t* dst, raftq_t* src);

/* ---- measurement hooks (bench harness) --------------------------------- */
/* HIP events recorded on the handle's own stream, so the elapsed time covers
 * exactly the kernels enqueued between begin and end. */
int raftq_timer_begin(raftq_t* h);
int raftq_timer_end(raftq_t* h, float* elapsed_ms);

/* ---- page-locked host memory for the caller's bulk buffers ------------- */
/* Every entry point that takes or fills a caller-owned host buffer accepts any
 * memory; given a buffer from raftq_host_alloc the transfer is a direct DMA at
 * PCIe speed instead of a staged pageable copy (bulk loads, read-backs, the
 * wire / WAL codecs of raftq_wire.h).  Needs a GPU runtime: RAFTQ_ENODEV /
 * RAFTQ_ENOMEM otherwise.  A cgo caller wraps the pointer with unsafe.Slice;
 * the memory is not Go-managed, so the cgo pointer rules do not apply to it. */
int raftq_host_alloc(void** p, uint64_t bytes);
void raftq_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* RAFTQ_H */
