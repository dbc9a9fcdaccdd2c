/*
 * raftq_step_oracle.c -- CPU restatement of etcd raft.Step for the message kinds
 * that carry no entries.  TEST INFRASTRUCTURE ONLY (see raftq_oracle.h): never
 * linked into libraftq.so; only tests/ and __graft_entry__.smoke() call it.
 *
 * PARITY UNPINNED.  The reference enters this code through
 *     raftNode.Process -> rc.node.Step(ctx, m)      raft.go:268-270
 *     rc.node.Tick() -> MsgHup / MsgBeat            raft.go:223-224
 * but the functions themselves (raft.Step, stepLeader, stepCandidate,
 * stepFollower, becomeFollower/Candidate/Leader, reset, poll, campaign,
 * handleHeartbeat, raftLog.isUpToDate/commitTo, Progress.maybeUpdate) are in the
 * un-vendored, un-pinned module github.com/coreos/etcd/raft.
 * They are restated from the published 2015-era (v2.2-v2.3 line) code and the Raft
 * paper, sections 5.1-5.4; every function below names the upstream function it
 * follows.  Deliberate restatement choices are marked CHOICE.
 *
 * One message at a time, in batch order, one group at a time: the shape of the
 * original (a single goroutine draining a channel), not of the GPU kernel.
 */
#include <stdlib.h>
#include <string.h>

#include "raftq_oracle.h"

#define ROLE_FOLLOWER 0
#define ROLE_CANDIDATE 1
#define ROLE_LEADER 2

/* one group's view into the SoA state */
typedef struct {
  rq_node_state_t* s;
  size_t g;
} node_t;

#define F(field) (r->s->field[r->g])
#define MATCH(p) (r->s->match[(size_t)(p) * r->s->ld + r->g])
#define VOTES(p) (r->s->votes[(size_t)(p) * r->s->ld + r->g])

/* etcd raft.reset(term):
 *     if r.Term != term { r.Term = term; r.Vote = None }
 *     r.lead = None; r.elapsed = 0; r.votes = make(map[uint64]bool)
 *     for i := range r.prs { r.prs[i] = &Progress{Next: lastIndex+1, ...}
 *                            if i == r.id { r.prs[i].Match = lastIndex } }
 * (Progress.Next and the inflight window are the caller's, see raftq_step.h.) */
static void reset(node_t* r, uint64_t term) {
  if (F(term) != term) {
    F(term) = term;
    F(vote) = 0;
  }
  F(lead) = 0;
  F(elapsed) = 0;
  for (int p = 0; p < r->s->n; ++p) {
    VOTES(p) = 0;
    MATCH(p) = ((uint32_t)p == r->s->self) ? F(last_index) : 0;
  }
}

/* etcd raft.becomeFollower(term, lead) */
static void become_follower(node_t* r, uint64_t term, uint32_t lead) {
  reset(r, term);
  F(lead) = lead;
  F(role) = ROLE_FOLLOWER;
  F(first_idx) = 0; /* not a leader: the current-term gate of maybeCommit is closed */
}

/* etcd raft.becomeCandidate(): reset(Term+1); Vote = id; state = Candidate */
static void become_candidate(node_t* r) {
  reset(r, F(term) + 1);
  F(vote) = r->s->self + 1;
  F(role) = ROLE_CANDIDATE;
  F(first_idx) = 0;
}

/* etcd raft.maybeCommit() + raftLog.maybeCommit(mci, r.Term) with the compact
 * gate (DESIGN.md "term gate"; equivalence with the full term lookup is
 * tests/test_oracle.py::test_compact_gate_equals_full_log_lookup). */
static int maybe_commit(node_t* r) {
  uint64_t m[RQ_MAX_PEERS];
  for (int p = 0; p < r->s->n; ++p) m[p] = MATCH(p);
  const uint64_t mci = rq_oracle_mci_sort(m, r->s->n);
  if (mci > F(committed) && F(first_idx) != 0 && mci >= F(first_idx)) {
    F(committed) = mci;
    return 1;
  }
  return 0;
}

/* etcd raft.becomeLeader(): reset(Term); lead = id; state = Leader;
 * appendEntry(pb.Entry{Data: nil}) -- the empty entry of the new term, which
 * is what opens the commit gate (Raft 5.4.2); appendEntry ends with
 * prs[id].maybeUpdate(lastIndex) and maybeCommit(). */
static void become_leader(node_t* r) {
  reset(r, F(term));
  F(lead) = r->s->self + 1;
  F(role) = ROLE_LEADER;
  F(last_index) += 1;
  F(last_term) = F(term);
  F(first_idx) = F(last_index);
  MATCH(r->s->self) = F(last_index);
  (void)maybe_commit(r); /* commits at once in a single-voter group */
}

/* etcd raft.poll(id, v): first response of a peer wins; returns #granted */
static int poll(node_t* r, uint32_t from, int granted) {
  if (VOTES(from) != 1 && VOTES(from) != 2) VOTES(from) = granted ? 1 : 2;
  int gr = 0;
  for (int p = 0; p < r->s->n; ++p) gr += (VOTES(p) == 1);
  return gr;
}

static int votes_recorded(node_t* r) {
  int c = 0;
  for (int p = 0; p < r->s->n; ++p) c += (VOTES(p) == 1 || VOTES(p) == 2);
  return c;
}

/* etcd raftLog.isUpToDate(lasti, term) -- Raft 5.4.1 */
static int is_up_to_date(node_t* r, uint64_t lasti, uint64_t term) {
  return term > F(last_term) || (term == F(last_term) && lasti >= F(last_index));
}

/* etcd raftLog.commitTo(tocommit): never decreases.  CHOICE: upstream panics
 * when tocommit > lastIndex (a correct leader never sends that: it clamps to
 * Progress.Match); a vector engine cannot panic per lane, so it clamps. */
static void commit_to(node_t* r, uint64_t tocommit) {
  if (tocommit > F(last_index)) tocommit = F(last_index);
  if (F(committed) < tocommit) F(committed) = tocommit;
}

/* etcd handleAppendEntries after the header was accepted.  With RAFTQ_MSGF_ENTRIES the message says how many entries it
 * carries and the last one's term; when it appends at the tail, raftLog.maybeAppend reduces to
 *   matchTerm(m.Index, m.LogTerm) -> findConflict: nothing behind the tail -> append -> commitTo(min(m.Commit, lastnewi))
 * and is done here (raftq_step.h); every other MsgApp is left to the log's owner (RAFTQ_OUT_APPEND). */
static void handle_append(node_t* r, const raftq_msg_t* m, raftq_step_out_t* o) {
  o->type = RAFTQ_OUT_APPEND;
  if ((m->_pad[1] & RAFTQ_MSGF_ENTRIES) && m->index == F(last_index) && m->log_term == F(last_term)) {
    const uint64_t k = m->_resv & 0xffffffffull;
    if (k) {
      F(last_index) = m->index + k;
      F(last_term) = m->reject_hint;
    }
    commit_to(r, m->commit); /* clamps to lastIndex = lastnewi */
    o->type = RAFTQ_OUT_APPENDED;
    o->index = F(last_index);
  }
}

static void out_common(node_t* r, const raftq_msg_t* m, raftq_step_out_t* o) {
  o->group = m->group;
  o->term = F(term);
  o->commit = F(committed);
  o->last_index = F(last_index);
  o->to = m->from;
  o->vote = F(vote);
  o->lead = F(lead);
  o->role = F(role);
}

/* etcd raft.Step(m) followed by r.step(r, m) */
static void step(node_t* r, const raftq_msg_t* m, raftq_step_out_t* o) {
  const int q = rq_oracle_quorum(r->s->n);
  const uint64_t term0 = F(term), commit0 = F(committed);
  const uint32_t vote0 = F(vote);
  const uint8_t role0 = F(role);
  memset(o, 0, sizeof(*o));

  if (m->type == RAFTQ_MSG_HUP) {
    /* Step: `if m.Type == pb.MsgHup { if r.state != StateLeader { r.campaign() } }`
     * campaign: becomeCandidate(); if q == poll(id, true) { becomeLeader() }
     *           else send MsgVote{Index: lastIndex, LogTerm: lastTerm} to the others */
    if (F(role) != ROLE_LEADER) {
      become_candidate(r);
      if (q == poll(r, r->s->self, 1)) {
        become_leader(r);
        o->type = RAFTQ_OUT_BECAME_LEADER;
        o->index = F(last_index);
        o->log_term = F(last_term);
      } else {
        o->type = RAFTQ_OUT_CAMPAIGN;
        o->index = F(last_index);
        o->log_term = F(last_term);
      }
    }
    goto done;
  }

  if (m->term == 0) {
    /* local message */
  } else if (m->term > F(term)) {
    /* `lead := m.From; if m.Type == pb.MsgVote { lead = None }; r.becomeFollower(m.Term, lead)` */
    become_follower(r, m->term, m->type == RAFTQ_MSG_VOTE ? 0 : m->from + 1);
  } else if (m->term < F(term)) {
    goto done; /* ignored */
  }

  switch (F(role)) {
    case ROLE_LEADER: /* etcd stepLeader */
      switch (m->type) {
        case RAFTQ_MSG_BEAT:
          o->type = RAFTQ_OUT_BCAST_HEARTBEAT;
          break;
        case RAFTQ_MSG_VOTE:
          o->type = RAFTQ_OUT_VOTE_RESP;
          o->reject = 1;
          break;
        case RAFTQ_MSG_APP_RESP:
          /* `if m.Reject { pr.maybeDecrTo(...) -> sendAppend }` is flow control (caller's);
           * else `if pr.maybeUpdate(m.Index) { ...; if r.maybeCommit() { r.bcastAppend() } }` */
          o->type = RAFTQ_OUT_PROGRESS;
          o->reject = m->reject;
          if (!m->reject) {
            /* CHOICE: an ack beyond the leader's own last index cannot come from a correct
             * follower; it is clamped so Match never exceeds lastIndex. */
            const uint64_t idx = m->index > F(last_index) ? F(last_index) : m->index;
            if (MATCH(m->from) < idx) { /* Progress.maybeUpdate */
              MATCH(m->from) = idx;
              o->flags |= RAFTQ_OUTF_UPDATED;
              (void)maybe_commit(r);
            }
          }
          o->index = MATCH(m->from);
          break;
        case RAFTQ_MSG_HEARTBEAT_RESP:
          /* `if pr.Match < r.raftLog.lastIndex() { r.sendAppend(m.From) }` -- the caller
           * decides from index (= Match) and last_index */
          o->type = RAFTQ_OUT_PROGRESS;
          o->index = MATCH(m->from);
          break;
        default:
          break; /* MsgApp / MsgHeartbeat / MsgVoteResp at a leader of the same term: no case upstream */
      }
      break;
    case ROLE_CANDIDATE: /* etcd stepCandidate */
      switch (m->type) {
        case RAFTQ_MSG_APP:
          become_follower(r, F(term), m->from + 1);
          handle_append(r, m, o);
          break;
        case RAFTQ_MSG_HEARTBEAT:
          become_follower(r, F(term), m->from + 1);
          commit_to(r, m->commit); /* handleHeartbeat */
          o->type = RAFTQ_OUT_HEARTBEAT_RESP;
          break;
        case RAFTQ_MSG_VOTE:
          o->type = RAFTQ_OUT_VOTE_RESP;
          o->reject = 1;
          break;
        case RAFTQ_MSG_VOTE_RESP: {
          /* `gr := r.poll(m.From, !m.Reject)
           *  switch r.q() { case gr: becomeLeader(); bcastAppend()
           *                 case len(r.votes) - gr: becomeFollower(r.Term, None) }` */
          const int gr = poll(r, m->from, !m->reject);
          if (q == gr) {
            become_leader(r);
            o->type = RAFTQ_OUT_BECAME_LEADER;
            o->index = F(last_index);
            o->log_term = F(last_term);
          } else if (q == votes_recorded(r) - gr) {
            become_follower(r, F(term), 0);
          }
          break;
        }
        default:
          break;
      }
      break;
    default: /* etcd stepFollower */
      switch (m->type) {
        case RAFTQ_MSG_APP:
          F(elapsed) = 0;
          F(lead) = m->from + 1;
          handle_append(r, m, o);
          break;
        case RAFTQ_MSG_HEARTBEAT:
          F(elapsed) = 0;
          F(lead) = m->from + 1;
          commit_to(r, m->commit); /* handleHeartbeat */
          o->type = RAFTQ_OUT_HEARTBEAT_RESP;
          break;
        case RAFTQ_MSG_VOTE:
          /* `if (r.Vote == None || r.Vote == m.From) && r.raftLog.isUpToDate(m.Index, m.LogTerm)` */
          o->type = RAFTQ_OUT_VOTE_RESP;
          if ((F(vote) == 0 || F(vote) == m->from + 1) && is_up_to_date(r, m->index, m->log_term)) {
            F(elapsed) = 0;
            F(vote) = m->from + 1;
          } else {
            o->reject = 1;
          }
          break;
        default:
          break;
      }
      break;
  }

done:
  out_common(r, m, o);
  if (F(term) != term0 || F(vote) != vote0 || F(committed) != commit0) o->flags |= RAFTQ_OUTF_HARDSTATE;
  if (F(committed) != commit0) o->flags |= RAFTQ_OUTF_COMMITTED;
  if (role0 != ROLE_FOLLOWER && F(role) == ROLE_FOLLOWER) o->flags |= RAFTQ_OUTF_STEPPED_DOWN;
}

void rq_oracle_step_batch(rq_node_state_t* s, const raftq_msg_t* msgs, size_t n, raftq_step_out_t* out) {
  /* RAFTQ_MSGF_BARRIER (raftq_step.h): once a MsgApp that carries it is left to the log's owner (RAFTQ_OUT_APPEND), the
   * group's later messages OF THIS BATCH are not applied (RAFTQ_OUT_DEFERRED); RAFTQ_MSGF_HOLD does the same unconditionally
   * and is itself not stepped; RAFTQ_MSGF_SKIP records are nobody's.  held[]: the groups in that state, one byte
   * each, only allocated when a message of the batch carries the flag. */
  uint8_t* held = NULL;
  for (size_t i = 0; i < n && !held; ++i)
    if ((msgs[i].type == RAFTQ_MSG_APP && (msgs[i]._pad[1] & RAFTQ_MSGF_BARRIER)) || (msgs[i]._pad[1] & RAFTQ_MSGF_HOLD))
      held = (uint8_t*)calloc(s->G ? s->G : 1, 1);
  for (size_t i = 0; i < n; ++i) {
    if (msgs[i]._pad[1] & RAFTQ_MSGF_SKIP) { /* not a message: no field is looked at */
      memset(&out[i], 0, sizeof(out[i]));
      out[i].type = RAFTQ_OUT_SKIPPED;
      continue;
    }
    node_t r = {s, (size_t)msgs[i].group};
    if (msgs[i]._pad[1] & RAFTQ_MSGF_HOLD) { /* the caller's (MsgProp): answered where it stands, the rest of its group waits */
      memset(&out[i], 0, sizeof(out[i]));
      out_common(&r, &msgs[i], &out[i]);
      out[i].type = RAFTQ_OUT_HELD;
      held[r.g] = 1;
      continue;
    }
    if (held && held[r.g]) {
      memset(&out[i], 0, sizeof(out[i]));
      out_common(&r, &msgs[i], &out[i]);
      out[i].type = RAFTQ_OUT_DEFERRED;
      continue;
    }
    step(&r, &msgs[i], &out[i]);
    if (held && out[i].type == RAFTQ_OUT_APPEND && msgs[i].type == RAFTQ_MSG_APP && (msgs[i]._pad[1] & RAFTQ_MSGF_BARRIER)) held[r.g] = 1;
  }
  free(held);
}

/* the log's owner reports its new tail: leader = appendEntry's bookkeeping
 * (prs[id].maybeUpdate(lastIndex); maybeCommit()), follower = the tail of
 * handleAppendEntries (`commitTo(min(m.Commit, lastnewi))`). */
void rq_oracle_apply_log_deltas(rq_node_state_t* s, const raftq_log_delta_t* d, size_t n, uint64_t* committed_out) {
  for (size_t i = 0; i < n; ++i) {
    node_t rr = {s, (size_t)d[i].group};
    node_t* r = &rr;
    F(last_index) = d[i].last_index;
    F(last_term) = d[i].last_term;
    if (F(role) == ROLE_LEADER) {
      if (MATCH(s->self) < F(last_index)) MATCH(s->self) = F(last_index);
      (void)maybe_commit(r);
    } else if (d[i].commit_to != 0) {
      commit_to(r, d[i].commit_to);
    }
    if (committed_out) committed_out[i] = F(committed);
  }
}
