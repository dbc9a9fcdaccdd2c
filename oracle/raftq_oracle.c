/*
 * raftq_oracle.c -- CPU restatement of the quorum hot path.  TEST
 * INFRASTRUCTURE ONLY (see raftq_oracle.h): never linked into libraftq.so.
 *
 * PARITY UNPINNED: the algorithm lives in github.com/coreos/etcd/raft, which
 * /root/reference imports (raft.go:27-34) but does not vendor or pin.  Each
 * function names the upstream function it restates and the reference call
 * site that reaches it; the shapes follow the published 2015-era etcd code
 * and the Raft paper (sections 5.2, 5.3, 5.4.2).
 */
#define _GNU_SOURCE
#include "raftq_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* a5 -- etcd raft.q(): `return len(r.prs)/2 + 1`.
 * Reached from raft.go:269 (Step) and raft.go:214 (Propose). */
int rq_oracle_quorum(int n) { return n / 2 + 1; }

/* Go's sort.Sort on <= 12 elements is an insertion sort; descending because
 * upstream wraps the slice in sort.Reverse. */
static void insertion_sort_desc(uint64_t* a, int n) {
  for (int i = 1; i < n; ++i) {
    for (int j = i; j > 0 && a[j] > a[j - 1]; --j) {
      uint64_t t = a[j];
      a[j] = a[j - 1];
      a[j - 1] = t;
    }
  }
}

/* a6 -- etcd raft.maybeCommit():
 *     mis := make(uint64Slice, 0, len(r.prs))
 *     for id := range r.prs { mis = append(mis, r.prs[id].Match) }
 *     sort.Sort(sort.Reverse(mis))
 *     mci := mis[r.q()-1]
 * Reached once per MsgAppResp (raft.go:268-270) and per local append
 * (raft.go:211-215).  The allocation is part of the shape being restated. */
uint64_t rq_oracle_mci_sort(const uint64_t* match, int n) {
  uint64_t* mis = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
  if (!mis) return 0;
  for (int p = 0; p < n; ++p) mis[p] = match[p];
  insertion_sort_desc(mis, n);
  uint64_t mci = mis[rq_oracle_quorum(n) - 1];
  free(mis);
  return mci;
}

/* Independent of any sort: Raft 5.3/5.4 define the commit candidate as the
 * largest index i such that a majority has matchIndex >= i.  Only the match
 * values themselves can be that maximum (and 0 always qualifies). */
uint64_t rq_oracle_mci_count(const uint64_t* match, int n) {
  const int q = rq_oracle_quorum(n);
  uint64_t best = 0;
  for (int c = 0; c < n; ++c) {
    int ge = 0;
    for (int p = 0; p < n; ++p) ge += (match[p] >= match[c]);
    if (ge >= q && match[c] > best) best = match[c];
  }
  return best;
}

/* etcd raftLog.term(i) + zeroTermOnErrCompacted: the dummy index, compacted
 * indices and indices past the last entry have term 0. */
uint64_t rq_oracle_log_term(const uint64_t* run_start, const uint64_t* run_term,
                            int nruns, uint64_t last_index, uint64_t i) {
  if (nruns <= 0 || i == 0 || i < run_start[0] || i > last_index) return 0;
  uint64_t t = 0;
  for (int r = 0; r < nruns; ++r) {
    if (run_start[r] <= i) t = run_term[r];
    else break;
  }
  return t;
}

/* a7 -- etcd raftLog.maybeCommit(maxIndex, term):
 *     if maxIndex > l.committed && l.zeroTermOnErrCompacted(l.term(maxIndex)) == term {
 *         l.commitTo(maxIndex); return true }
 * The ungated variant (BASELINE config 2) drops the term test. */
uint64_t rq_oracle_maybe_commit(uint64_t mci, uint64_t committed, int gated,
                                uint64_t term_of_mci, uint64_t cur_term) {
  if (mci > committed && (!gated || term_of_mci == cur_term)) return mci;
  return committed;
}

/* a8 -- etcd raft.poll() counts r.votes[id]==true; the candidate's
 * MsgVoteResp handler then does
 *     switch r.q() { case gr: becomeLeader; case len(r.votes) - gr: becomeFollower }
 * i.e. won when granted reaches q, lost when rejections reach q (2015-era
 * rule; SURVEY.md 8a even-N caveat).  Reached from raft.go:268-270. */
uint8_t rq_oracle_poll(const uint8_t* votes, int n) {
  const int q = rq_oracle_quorum(n);
  int granted = 0, rejected = 0;
  for (int p = 0; p < n; ++p) {
    if (votes[p] == 1) ++granted;
    else if (votes[p] == 2) ++rejected;
  }
  if (granted >= q) return 1;
  if (rejected >= q) return 2;
  return 0;
}

uint64_t rq_oracle_commit_advance(const uint64_t* match, size_t ld, int n, size_t G,
                                  const uint64_t* committed, int gated,
                                  const uint64_t* first_idx_cur_term,
                                  uint64_t* committed_out) {
  uint64_t changed = 0;
  uint64_t col[16];
  for (size_t g = 0; g < G; ++g) {
    for (int p = 0; p < n; ++p) col[p] = match[(size_t)p * ld + g];
    const uint64_t mci = rq_oracle_mci_sort(col, n);
    uint64_t nc;
    if (gated) {
      /* compact encoding: terms are non-decreasing along the log and no entry
       * is newer than the leader's term, so term(i)==cur_term <=> i >= first
       * index of cur_term (0 = the log has no entry of cur_term). */
      const uint64_t f = first_idx_cur_term[g];
      const int term_ok = (f != 0 && mci >= f);
      nc = rq_oracle_maybe_commit(mci, committed[g], 1, term_ok ? 1 : 0, 1);
    } else {
      nc = rq_oracle_maybe_commit(mci, committed[g], 0, 0, 0);
    }
    changed += (nc != committed[g]);
    committed_out[g] = nc;
  }
  return changed;
}

uint64_t rq_oracle_commit_advance_log(const uint64_t* match, size_t ld, int n, size_t G,
                                      const uint64_t* committed, const uint64_t* cur_term,
                                      const uint64_t* run_off, const uint64_t* run_start,
                                      const uint64_t* run_term, const uint64_t* last_index,
                                      uint64_t* committed_out) {
  uint64_t changed = 0;
  uint64_t col[16];
  for (size_t g = 0; g < G; ++g) {
    for (int p = 0; p < n; ++p) col[p] = match[(size_t)p * ld + g];
    const uint64_t mci = rq_oracle_mci_count(col, n);
    const uint64_t o = run_off[g];
    const int nr = (int)(run_off[g + 1] - o);
    const uint64_t t = rq_oracle_log_term(run_start + o, run_term + o, nr, last_index[g], mci);
    /* only a leader runs maybeCommit and a leader's term is >= 1; a group
     * with cur_term 0 has never seen an election and commits nothing (this
     * keeps "term 0 never equals a live term" true at the 0 == 0 corner). */
    const uint64_t nc = cur_term[g] == 0
                            ? committed[g]
                            : rq_oracle_maybe_commit(mci, committed[g], 1, t, cur_term[g]);
    changed += (nc != committed[g]);
    committed_out[g] = nc;
  }
  return changed;
}

void rq_oracle_first_idx_cur_term(size_t G, const uint64_t* cur_term,
                                  const uint64_t* run_off, const uint64_t* run_start,
                                  const uint64_t* run_term, uint64_t* first_idx_out) {
  for (size_t g = 0; g < G; ++g) {
    uint64_t f = 0;
    for (uint64_t r = run_off[g]; r < run_off[g + 1]; ++r) {
      if (run_term[r] == cur_term[g] && cur_term[g] != 0) {
        f = run_start[r];
        break;
      }
    }
    first_idx_out[g] = f;
  }
}

void rq_oracle_vote_tally(const uint8_t* votes, size_t ld, int n, size_t G,
                          uint8_t* outcome_out, uint64_t* n_won, uint64_t* n_lost) {
  uint64_t w = 0, l = 0;
  uint8_t col[16];
  for (size_t g = 0; g < G; ++g) {
    for (int p = 0; p < n; ++p) col[p] = votes[(size_t)p * ld + g];
    const uint8_t o = rq_oracle_poll(col, n);
    w += (o == 1);
    l += (o == 2);
    outcome_out[g] = o;
  }
  if (n_won) *n_won = w;
  if (n_lost) *n_lost = l;
}

/* etcd Progress.maybeUpdate(n): `if pr.Match < n { pr.Match = n }` --
 * the leader-side effect of a MsgAppResp (raft.go:268-270). */
void rq_oracle_apply_deltas(uint64_t* match, size_t ld, int n, size_t G,
                            const uint64_t* d_group, const uint32_t* d_peer,
                            const uint64_t* d_match, size_t nd) {
  for (size_t i = 0; i < nd; ++i) {
    if (d_group[i] >= G || d_peer[i] >= (uint32_t)n) continue;
    uint64_t* slot = &match[(size_t)d_peer[i] * ld + d_group[i]];
    if (*slot < d_match[i]) *slot = d_match[i];
  }
}

/* etcd raft.poll(): `if _, ok := r.votes[id]; !ok { r.votes[id] = v }` --
 * the first response from a peer wins; later ones are ignored. */
void rq_oracle_apply_vote_deltas(uint8_t* votes, size_t ld, int n, size_t G,
                                 const uint64_t* d_group, const uint32_t* d_peer,
                                 const uint8_t* d_vote, size_t nd) {
  for (size_t i = 0; i < nd; ++i) {
    if (d_group[i] >= G || d_peer[i] >= (uint32_t)n) continue;
    if (d_vote[i] != 1 && d_vote[i] != 2) continue;
    uint8_t* slot = &votes[(size_t)d_peer[i] * ld + d_group[i]];
    if (*slot != 1 && *slot != 2) *slot = d_vote[i];
  }
}

/* ------------------------------------------------------------------------ */
/* batched Tick: rc.node.Tick() every 100 ms (raft.go:207, 223-224)          */

/* The timeout draw's stream (round 6: redefined, oracle + kernel + tests together).  Go's math/rand cannot be reproduced
 * without the Go runtime, so the stream is this repo's own definition; what matters is that it is uniform and that the
 * device can afford it for every follower of every group on every tick.  Round 5's splitmix64 of (seed, tick, group)
 * was three 64-bit multiplies per GROUP.  Now: one splitmix64 finaliser per TICK makes a 64-bit key (the same for
 * every group: scalar work on the device), and a group's draw is murmur3's 32-bit finaliser -- two 32-bit multiplies --
 * of (group ^ key.lo), xored with key.hi. */
uint64_t rq_oracle_tick_key(uint64_t seed, uint64_t tick_no) {
  uint64_t z = seed ^ (tick_no * 0xD1B54A32D192ED03ull);
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
uint32_t rq_oracle_tick_rand(uint64_t seed, uint64_t tick_no, uint64_t group) {
  const uint64_t key = rq_oracle_tick_key(seed, tick_no);
  uint32_t x = ((uint32_t)group ^ (uint32_t)(group >> 32)) ^ (uint32_t)key;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x ^ (uint32_t)(key >> 32);
}

/* etcd raft.tickHeartbeat (leaders):
 *     r.elapsed++; if r.elapsed >= r.heartbeatTimeout { r.elapsed = 0; Step(MsgBeat) }
 * etcd raft.tickElection (followers, candidates):
 *     r.elapsed++; if r.isElectionTimeout() { r.elapsed = 0; Step(MsgHup) }
 * etcd raft.isElectionTimeout:
 *     d := r.elapsed - r.electionTimeout; if d < 0 { return false }
 *     return d > r.rand.Int() % r.electionTimeout                              */
void rq_oracle_tick(const uint8_t* role, uint32_t* elapsed, size_t G, uint32_t election_tick,
                    uint32_t heartbeat_tick, uint64_t seed, uint64_t tick_no, uint8_t* action_out,
                    uint64_t* n_hup, uint64_t* n_beat) {
  uint64_t hup = 0, beat = 0;
  for (size_t g = 0; g < G; ++g) {
    uint8_t act = 0;
    uint32_t e = elapsed[g] + 1;
    if (role[g] == 2) {
      if (e >= heartbeat_tick) {
        e = 0;
        act = 2;
      }
    } else {
      const int64_t d = (int64_t)e - (int64_t)election_tick;
      if (d >= 0 && d > (int64_t)(rq_oracle_tick_rand(seed, tick_no, g) % election_tick)) {
        e = 0;
        act = 1;
      }
    }
    elapsed[g] = e;
    action_out[g] = act;
    hup += (act == 1);
    beat += (act == 2);
  }
  if (n_hup) *n_hup = hup;
  if (n_beat) *n_beat = beat;
}

/* etcd raft.campaign -> becomeCandidate: r.reset(term+1) clears r.votes and
 * r.elapsed, r.Vote = r.id, state = candidate; then poll(r.id, true). */
void rq_oracle_campaign(uint8_t* role, uint32_t* elapsed, uint8_t* votes, size_t ld, int n, size_t G,
                        const uint64_t* groups, size_t ng, uint32_t self_peer) {
  for (size_t i = 0; i < ng; ++i) {
    const uint64_t g = groups[i];
    if (g >= G) continue;
    role[g] = 1;
    elapsed[g] = 0;
    for (int p = 0; p < n; ++p) votes[(size_t)p * ld + g] = ((uint32_t)p == self_peer) ? 1 : 0;
  }
}

/* ------------------------------------------------------------------------ */
/* timed CPU baselines                                                      */

#define CE_DESC(a, b)            \
  do {                           \
    uint64_t _x = (a), _y = (b); \
    (a) = _x > _y ? _x : _y;     \
    (b) = _x > _y ? _y : _x;     \
  } while (0)

/* tight kernel: branch-free descending sorting networks, no allocation */
static inline uint64_t mci_network(uint64_t* v, int n) {
  switch (n) {
    case 1: break;
    case 2: CE_DESC(v[0], v[1]); break;
    case 3: CE_DESC(v[0], v[1]); CE_DESC(v[1], v[2]); CE_DESC(v[0], v[1]); break;
    case 5:
      CE_DESC(v[0], v[1]); CE_DESC(v[3], v[4]); CE_DESC(v[2], v[4]); CE_DESC(v[2], v[3]);
      CE_DESC(v[0], v[3]); CE_DESC(v[0], v[2]); CE_DESC(v[1], v[4]); CE_DESC(v[1], v[3]);
      CE_DESC(v[1], v[2]);
      break;
    default:
      for (int i = 0; i < n; ++i)
        for (int j = (i & 1); j + 1 < n; j += 2) CE_DESC(v[j], v[j + 1]);
      break;
  }
  return v[n / 2];
}

typedef struct {
  int kind, sweeps, n, gated;
  size_t g0, g1, ld, ldv;
  const uint64_t *match, *committed, *first_idx;
  const uint8_t* votes;
  uint64_t* committed_out;
  uint8_t* outcome_out;
} sweep_job_t;

static void* sweep_worker(void* arg) {
  sweep_job_t* j = (sweep_job_t*)arg;
  const int n = j->n;
  uint64_t col[16];
  uint8_t vc[16];
  for (int s = 0; s < j->sweeps; ++s) {
    for (size_t g = j->g0; g < j->g1; ++g) {
      for (int p = 0; p < n; ++p) col[p] = j->match[(size_t)p * j->ld + g];
      const uint64_t mci = j->kind == 0 ? rq_oracle_mci_sort(col, n) : mci_network(col, n);
      uint64_t c = j->committed[g];
      if (mci > c && (!j->gated || (j->first_idx[g] != 0 && mci >= j->first_idx[g]))) c = mci;
      j->committed_out[g] = c;
      if (j->votes) {
        for (int p = 0; p < n; ++p) vc[p] = j->votes[(size_t)p * j->ldv + g];
        j->outcome_out[g] = rq_oracle_poll(vc, n);
      }
    }
  }
  return NULL;
}

double rq_oracle_timed_sweeps(int kind, int threads, int sweeps,
                              const uint64_t* match, size_t ld, int n, size_t G,
                              const uint64_t* committed, int gated,
                              const uint64_t* first_idx_cur_term,
                              const uint8_t* votes, size_t ldv,
                              uint64_t* committed_out, uint8_t* outcome_out) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t tid[256];
  sweep_job_t job[256];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; ++t) {
    sweep_job_t* j = &job[t];
    j->kind = kind; j->sweeps = sweeps; j->n = n; j->gated = gated;
    j->g0 = G * (size_t)t / (size_t)threads;
    j->g1 = G * (size_t)(t + 1) / (size_t)threads;
    j->ld = ld; j->ldv = ldv;
    j->match = match; j->committed = committed; j->first_idx = first_idx_cur_term;
    j->votes = votes; j->committed_out = committed_out; j->outcome_out = outcome_out;
    if (t + 1 < threads) pthread_create(&tid[t], NULL, sweep_worker, j);
  }
  sweep_worker(&job[threads - 1]);
  for (int t = 0; t + 1 < threads; ++t) pthread_join(tid[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
