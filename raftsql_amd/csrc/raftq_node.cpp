// raftq_node.cpp -- one raft node for G groups (include/raftq_node.h): the C++ stand-in for
// the reference's raftNode + serveChannels (raft.go:36-78, 204-246), G-fold.
//
// Division of labour.  Every consensus decision is the GPU engine's: raftq_step_batch (Step for
// all payload-free message kinds + the MsgApp header), raftq_tick, raftq_apply_log_deltas.  This
// file owns what the reference's raftNode owns around raft.Node: the log entries
// (raft.MemoryStorage, raft.go:70), the replication cursor (Progress.Next), queues and channels.
// It computes no quorum, no vote tally and no commit index.
#include "raftq_node.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

struct Entry {
  uint64_t term;
  std::string data;
};

struct Item {
  int kind;
  std::string data;
};

struct InMsg {
  raftq_msg_t h;
  std::vector<Entry> ents;
};

struct Group {
  std::vector<Entry> log;  // log[i] holds index i + 1 (no compaction)
  uint64_t committed = 0, applied = 0, term = 0;
  uint32_t lead = 0, vote = 0;
  uint8_t role = RAFTQ_ROLE_FOLLOWER;
  std::vector<uint64_t> next, match;  // leader only: Progress.Next / mirror of Progress.Match
  std::vector<Item> q;                // commit channel
  size_t qhead = 0;
  uint64_t hs_term = 0, hs_commit = 0;  // raftq_node_set_hard_state
  uint32_t hs_vote = 0;
};

constexpr uint64_t kMaxEntriesPerMsg = 1024;     // with kMaxBytesPerMsg: raft.Config.MaxSizePerMsg (raft.go:157)
constexpr uint64_t kMaxBytesPerMsg = 1 << 20;

}  // namespace

struct raftq_node {
  raftq_t* h = nullptr;
  uint64_t G = 0;
  uint32_t N = 0, self = 0;
  std::vector<Group> groups;
  std::mutex mu;  // guards inbound / proposals / pending_ticks / outbound / commit channels / status
  std::condition_variable cv_commit;
  std::vector<InMsg> inbound;
  std::vector<std::pair<uint64_t, std::string>> proposals;
  uint32_t pending_ticks = 0;
  std::vector<std::vector<std::string>> outbound;  // [peer] -> frames
  bool started = false, closed = false;
  int error = 0;
  std::string errtext;
  std::mutex turn_mu;  // one advance() at a time
  raftq_node_stats_t stats{};
  // scratch of advance()
  std::vector<uint8_t> action;
  std::vector<raftq_log_delta_t> deltas;
  std::vector<uint64_t> delta_commit;
};

namespace {

int nfail(raftq_node_t* n, int code, const std::string& msg) {
  if (n) {
    std::lock_guard<std::mutex> lk(n->mu);
    n->errtext = msg;
  }
  return code;
}

// writeError (raft.go:136-142): record the error, close the commit side
int poison(raftq_node_t* n, int code, const std::string& what) {
  const char* m = raftq_last_error(n->h);
  std::lock_guard<std::mutex> lk(n->mu);
  if (n->error == 0) {
    n->error = code;
    n->errtext = what + ": " + (m ? m : "?");
  }
  n->closed = true;
  n->cv_commit.notify_all();
  return code;
}

uint64_t term_at(const Group& g, uint64_t index) {
  return (index == 0 || index > g.log.size()) ? 0 : g.log[index - 1].term;
}

void put_frame(raftq_node_t* n, uint32_t to, const raftq_msg_t& hdr, const Entry* ents, size_t n_ents) {
  std::string f;
  size_t bytes = sizeof(raftq_msg_t);
  for (size_t i = 0; i < n_ents; ++i) bytes += 16 + (ents[i].data.size() + 7) / 8 * 8;
  f.reserve(bytes);
  raftq_msg_t h = hdr;
  h._resv = n_ents;
  f.append((const char*)&h, sizeof(h));
  for (size_t i = 0; i < n_ents; ++i) {
    const uint64_t term = ents[i].term;
    const uint32_t len = (uint32_t)ents[i].data.size(), zero = 0;
    f.append((const char*)&term, 8);
    f.append((const char*)&len, 4);
    f.append((const char*)&zero, 4);
    f.append(ents[i].data);
    f.append((8 - len % 8) % 8, '\0');
  }
  n->outbound[to].push_back(std::move(f));
  n->stats.msgs_sent++;
}

raftq_msg_t header(const raftq_node_t* n, uint64_t group, uint8_t type, uint64_t term) {
  raftq_msg_t m;
  std::memset(&m, 0, sizeof(m));
  m.group = group;
  m.type = type;
  m.term = term;  // raft.send: every non-MsgProp message carries r.Term
  m.from = n->self;
  return m;
}

// publishEntries (raft.go:82-96): entries (applied, upto] go to the commit channel; empty
// payloads (a new leader's no-op) are skipped.
void publish(raftq_node_t* n, Group& g, uint64_t upto) {
  upto = std::min<uint64_t>(upto, g.log.size());
  for (uint64_t idx = g.applied + 1; idx <= upto; ++idx) {
    const Entry& e = g.log[idx - 1];
    if (e.data.empty()) continue;
    g.q.push_back(Item{RAFTQ_NODE_ENTRY, e.data});
    n->stats.entries_published++;
  }
  if (upto > g.applied) g.applied = upto;
}

void note_commit(raftq_node_t* n, Group& g, uint64_t commit) {
  if (commit > g.committed) {
    g.committed = commit;
    publish(n, g, commit);
  }
}

// raft.sendAppend(to): entries from Progress.Next on; an empty MsgApp still carries the commit
// index.  Optimistic cursor (ProgressStateReplicate): Next jumps past what was sent.
void send_append(raftq_node_t* n, uint64_t gi, Group& g, uint32_t to) {
  const uint64_t last = g.log.size();
  uint64_t nx = std::max<uint64_t>(g.next[to], 1);
  if (nx > last + 1) nx = last + 1;
  raftq_msg_t m = header(n, gi, RAFTQ_MSG_APP, g.term);
  m.index = nx - 1;
  m.log_term = term_at(g, nx - 1);
  m.commit = g.committed;
  uint64_t cnt = 0, bytes = 0;
  while (nx + cnt <= last && cnt < kMaxEntriesPerMsg && bytes < kMaxBytesPerMsg) {
    bytes += g.log[nx + cnt - 1].data.size();
    ++cnt;
  }
  put_frame(n, to, m, cnt ? &g.log[nx - 1] : nullptr, cnt);
  g.next[to] = nx + cnt;
}

void bcast_append(raftq_node_t* n, uint64_t gi, Group& g) {
  for (uint32_t p = 0; p < n->N; ++p)
    if (p != n->self) send_append(n, gi, g, p);
}

// raft.bcastHeartbeat: `commit := min(r.prs[to].Match, r.raftLog.committed)`
void bcast_heartbeat(raftq_node_t* n, uint64_t gi, Group& g) {
  for (uint32_t p = 0; p < n->N; ++p) {
    if (p == n->self) continue;
    raftq_msg_t m = header(n, gi, RAFTQ_MSG_HEARTBEAT, g.term);
    m.commit = std::min(g.match.empty() ? 0 : g.match[p], g.committed);
    put_frame(n, p, m, nullptr, 0);
  }
}

// the leader's (or a forwarded) proposal: appendEntry on the leader, forward / drop elsewhere.
// Returns true when the log grew (the caller reports the tail and broadcasts).
bool handle_proposal(raftq_node_t* n, uint64_t gi, Group& g, std::vector<Entry>& ents) {
  if (g.role == RAFTQ_ROLE_LEADER) {
    for (Entry& e : ents) g.log.push_back(Entry{g.term, std::move(e.data)});
    return !ents.empty();
  }
  if (g.lead != 0 && g.lead - 1 != n->self) {  // stepFollower MsgProp: `m.To = r.lead; r.send(m)`
    raftq_msg_t m = header(n, gi, RAFTQ_MSG_PROP, 0);
    put_frame(n, g.lead - 1, m, ents.data(), ents.size());
  } else {
    n->stats.proposals_dropped += ents.size();  // no leader: etcd drops the proposal
  }
  return false;
}

// handleAppendEntries on the log's owner, after Step accepted the header (RAFTQ_OUT_APPEND)
void follower_append(raftq_node_t* n, uint64_t gi, Group& g, InMsg& im, std::vector<raftq_log_delta_t>& deltas) {
  const raftq_msg_t& m = im.h;
  raftq_msg_t r = header(n, gi, RAFTQ_MSG_APP_RESP, g.term);
  if (m.index < g.committed) {  // `if m.Index < r.raftLog.committed { send MsgAppResp{Index: committed} }`
    r.index = g.committed;
    put_frame(n, m.from, r, nullptr, 0);
    return;
  }
  if (m.index <= g.log.size() && term_at(g, m.index) == m.log_term) {  // raftLog.maybeAppend
    size_t k = 0;
    for (; k < im.ents.size(); ++k) {  // findConflict
      const uint64_t idx = m.index + 1 + k;
      if (idx > g.log.size()) break;
      if (g.log[idx - 1].term != im.ents[k].term) {
        g.log.resize(idx - 1);  // a conflicting suffix is never committed (Raft 5.3)
        break;
      }
    }
    for (; k < im.ents.size(); ++k) g.log.push_back(std::move(im.ents[k]));
    const uint64_t lastnewi = m.index + im.ents.size();
    r.index = lastnewi;
    put_frame(n, m.from, r, nullptr, 0);
    raftq_log_delta_t d;
    d.group = gi;
    d.last_index = g.log.size();
    d.last_term = term_at(g, g.log.size());
    d.commit_to = std::min(m.commit, lastnewi);  // `commitTo(min(m.Commit, lastnewi))`
    deltas.push_back(d);
  } else {  // reject with the hint
    r.index = m.index;
    r.reject = 1;
    r.reject_hint = g.log.size();
    put_frame(n, m.from, r, nullptr, 0);
  }
}

// report log tails to the engine, then publish whatever the reports committed.  Called with
// `lk` held; the device call runs unlocked.  On failure returns with `lk` RELEASED (so the caller
// can poison()), on success with it held again.
int flush_deltas(raftq_node_t* n, std::unique_lock<std::mutex>& lk) {
  if (n->deltas.empty()) return RAFTQ_OK;
  n->delta_commit.resize(n->deltas.size());
  lk.unlock();
  const int rc = raftq_apply_log_deltas(n->h, n->deltas.data(), n->deltas.size(), n->delta_commit.data());
  if (rc != RAFTQ_OK) return rc;
  lk.lock();
  for (size_t i = 0; i < n->deltas.size(); ++i) note_commit(n, n->groups[n->deltas[i].group], n->delta_commit[i]);
  n->deltas.clear();
  return RAFTQ_OK;
}

// what one Step result means for the node (the "Ready" consequences of one message)
void apply_result(raftq_node_t* n, const raftq_step_out_t& o, InMsg& im) {
  Group& g = n->groups[o.group];
  const uint64_t gi = o.group;
  g.term = o.term;
  g.lead = o.lead;
  g.vote = o.vote;
  g.role = o.role;
  if (o.flags & RAFTQ_OUTF_HARDSTATE) n->stats.hard_states++;  // wal.Save(rd.HardState, ...) (raft.go:228)
  if (o.flags & RAFTQ_OUTF_STEPPED_DOWN) {
    g.next.clear();
    g.match.clear();
  }
  note_commit(n, g, o.commit);
  switch (o.type) {
    case RAFTQ_OUT_VOTE_RESP: {
      raftq_msg_t r = header(n, gi, RAFTQ_MSG_VOTE_RESP, o.term);
      r.reject = o.reject;
      put_frame(n, o.to, r, nullptr, 0);
      break;
    }
    case RAFTQ_OUT_HEARTBEAT_RESP: {
      raftq_msg_t r = header(n, gi, RAFTQ_MSG_HEARTBEAT_RESP, o.term);
      put_frame(n, o.to, r, nullptr, 0);
      break;
    }
    case RAFTQ_OUT_CAMPAIGN:
      for (uint32_t p = 0; p < n->N; ++p) {
        if (p == n->self) continue;
        raftq_msg_t r = header(n, gi, RAFTQ_MSG_VOTE, o.term);
        r.index = o.index;
        r.log_term = o.log_term;
        put_frame(n, p, r, nullptr, 0);
      }
      break;
    case RAFTQ_OUT_BECAME_LEADER:
      // becomeLeader's appendEntry(pb.Entry{Data: nil}): the engine already counted it
      g.log.resize(std::min<uint64_t>(g.log.size(), o.index - 1));
      g.log.push_back(Entry{o.term, std::string()});
      g.next.assign(n->N, o.index);  // reset(): Next = lastIndex + 1 (before the empty entry)
      g.match.assign(n->N, 0);
      g.match[n->self] = o.index;
      g.next[n->self] = o.index + 1;
      note_commit(n, g, o.commit);
      bcast_append(n, gi, g);
      break;
    case RAFTQ_OUT_PROGRESS:
      if (g.next.empty()) break;
      if (im.h.type == RAFTQ_MSG_APP_RESP && im.h.reject) {
        // Progress.maybeDecrTo: a stale rejection is ignored, else back off to the hint
        if (im.h.index > g.match[o.to]) {
          g.next[o.to] = std::max<uint64_t>(std::min(im.h.index, im.h.reject_hint + 1), g.match[o.to] + 1);
          send_append(n, gi, g, o.to);
        }
      } else {
        g.match[o.to] = o.index;
        if (g.next[o.to] < o.index + 1) g.next[o.to] = o.index + 1;
        if (im.h.type == RAFTQ_MSG_APP_RESP) {
          if (o.flags & RAFTQ_OUTF_COMMITTED) bcast_append(n, gi, g);  // `if r.maybeCommit() { r.bcastAppend() }`
          else if (g.next[o.to] <= g.log.size()) send_append(n, gi, g, o.to);
        } else if (o.index < g.log.size()) {  // MsgHeartbeatResp: `if pr.Match < lastIndex { sendAppend }`
          if (g.next[o.to] > o.index + 1) g.next[o.to] = o.index + 1;  // whatever was in flight is lost: resend
          send_append(n, gi, g, o.to);
        }
      }
      break;
    case RAFTQ_OUT_APPEND:
      follower_append(n, gi, g, im, n->deltas);
      break;
    default:
      break;
  }
}

}  // namespace

extern "C" {

int raftq_node_create(int device, uint64_t n_groups, uint32_t n_peers, uint32_t self_peer, raftq_node_t** out) {
  if (!out) return RAFTQ_EINVAL;
  *out = nullptr;
  if (n_peers == 0 || self_peer >= n_peers) return RAFTQ_EINVAL;
  raftq_node_t* n = new (std::nothrow) raftq_node();
  if (!n) return RAFTQ_ENOMEM;
  int rc = raftq_create(device, n_groups, n_peers, &n->h);
  if (rc == RAFTQ_OK) rc = raftq_set_self(n->h, self_peer);
  if (rc != RAFTQ_OK) {
    if (n->h) raftq_destroy(n->h);
    delete n;
    return rc;  // text is in raftq_last_error(NULL)
  }
  n->G = n_groups;
  n->N = n_peers;
  n->self = self_peer;
  try {
    n->groups.resize(n_groups);
    n->outbound.resize(n_peers);
    n->action.resize(n_groups);
  } catch (...) {
    raftq_destroy(n->h);
    delete n;
    return RAFTQ_ENOMEM;
  }
  *out = n;
  return RAFTQ_OK;
}

int raftq_node_replay(raftq_node_t* n, uint64_t group, const uint64_t* terms, const void* const* data,
                      const uint32_t* lens, uint64_t count) {
  if (!n) return RAFTQ_EINVAL;
  if (group >= n->G) return nfail(n, RAFTQ_EINVAL, "replay: group out of range");
  if (count && (!terms || !data || !lens)) return nfail(n, RAFTQ_EINVAL, "replay: null argument");
  std::lock_guard<std::mutex> lk(n->mu);
  if (n->started) {
    n->errtext = "replay: node already started";
    return RAFTQ_ESTATE;
  }
  Group& g = n->groups[group];
  uint64_t prev = g.log.empty() ? 1 : g.log.back().term;
  for (uint64_t i = 0; i < count; ++i) {
    if (terms[i] < prev) {
      n->errtext = "replay: terms must be non-decreasing and >= 1";
      return RAFTQ_EINVAL;
    }
    prev = terms[i];
    g.log.push_back(Entry{terms[i], std::string((const char*)data[i], lens[i])});
  }
  return RAFTQ_OK;
}

int raftq_node_set_hard_state(raftq_node_t* n, uint64_t group, uint64_t term, uint32_t vote, uint64_t commit) {
  if (!n) return RAFTQ_EINVAL;
  if (group >= n->G || vote > n->N) return nfail(n, RAFTQ_EINVAL, "set_hard_state: argument out of range");
  std::lock_guard<std::mutex> lk(n->mu);
  if (n->started) {
    n->errtext = "set_hard_state: node already started";
    return RAFTQ_ESTATE;
  }
  Group& g = n->groups[group];
  g.hs_term = term;
  g.hs_vote = vote;
  g.hs_commit = commit;
  return RAFTQ_OK;
}

int raftq_node_start(raftq_node_t* n, uint32_t election_tick, uint32_t heartbeat_tick, uint64_t seed) {
  if (!n) return RAFTQ_EINVAL;
  std::vector<uint64_t> term, last_index, last_term, committed, match;
  std::vector<uint32_t> vote;
  {
    std::lock_guard<std::mutex> lk(n->mu);
    if (n->started) {
      n->errtext = "start: already started";
      return RAFTQ_ESTATE;
    }
    try {
      term.assign(n->G, 0);
      last_index.assign(n->G, 0);
      last_term.assign(n->G, 0);
      committed.assign(n->G, 0);
      vote.assign(n->G, 0);
      match.assign((size_t)n->N * n->G, 0);
    } catch (...) {
      n->errtext = "start: host allocation failed";
      return RAFTQ_ENOMEM;
    }
    for (uint64_t gi = 0; gi < n->G; ++gi) {
      Group& g = n->groups[gi];
      // replayWAL (raft.go:122-134): every logged entry goes out, then the nil sentinel
      publish(n, g, g.log.size());
      g.q.push_back(Item{RAFTQ_NODE_SENTINEL, std::string()});
      g.term = term[gi] = g.hs_term;
      g.vote = vote[gi] = g.hs_vote;
      g.committed = committed[gi] = std::min<uint64_t>(g.hs_commit, g.log.size());
      last_index[gi] = g.log.size();
      last_term[gi] = term_at(g, g.log.size());
      match[(size_t)n->self * n->G + gi] = g.log.size();  // newRaft: prs[id].Match = lastIndex
    }
    n->stats.entries_published = 0;  // replayed entries are not "live" publications
  }
  int rc = raftq_load_match(n->h, match.data(), committed.data());
  if (rc == RAFTQ_OK) rc = raftq_load_node(n->h, term.data(), vote.data(), nullptr, last_index.data(), last_term.data());
  if (rc == RAFTQ_OK) rc = raftq_set_timers(n->h, election_tick ? election_tick : 10, heartbeat_tick ? heartbeat_tick : 1, seed);
  if (rc != RAFTQ_OK) return poison(n, rc, "start");
  {
    std::lock_guard<std::mutex> lk(n->mu);
    n->started = true;
  }
  n->cv_commit.notify_all();
  return RAFTQ_OK;
}

int raftq_node_propose(raftq_node_t* n, uint64_t group, const void* data, uint32_t len) {
  if (!n) return RAFTQ_EINVAL;
  if (group >= n->G) return nfail(n, RAFTQ_EINVAL, "propose: group out of range");
  if (len && !data) return nfail(n, RAFTQ_EINVAL, "propose: null payload");
  std::lock_guard<std::mutex> lk(n->mu);
  if (!n->started || n->closed) {
    n->errtext = n->closed ? "propose: node is closed" : "propose: node not started";
    return RAFTQ_ESTATE;
  }
  n->proposals.emplace_back(group, std::string((const char*)data, len));
  return RAFTQ_OK;
}

int raftq_node_tick(raftq_node_t* n) {
  if (!n) return RAFTQ_EINVAL;
  std::lock_guard<std::mutex> lk(n->mu);
  if (!n->started || n->closed) {
    n->errtext = "tick: node not running";
    return RAFTQ_ESTATE;
  }
  n->pending_ticks++;
  return RAFTQ_OK;
}

int raftq_node_deliver(raftq_node_t* n, const void* frames, uint64_t len) {
  if (!n) return RAFTQ_EINVAL;
  if (len && !frames) return nfail(n, RAFTQ_EINVAL, "deliver: null buffer");
  std::vector<InMsg> parsed;
  const uint8_t* p = (const uint8_t*)frames;
  uint64_t off = 0;
  while (off < len) {
    if (len - off < sizeof(raftq_msg_t)) return nfail(n, RAFTQ_EINVAL, "deliver: truncated frame header");
    InMsg im;
    std::memcpy(&im.h, p + off, sizeof(raftq_msg_t));
    off += sizeof(raftq_msg_t);
    const uint64_t n_ents = im.h._resv;
    if (im.h.group >= n->G || im.h.from >= n->N || n_ents > (len - off) / 16)
      return nfail(n, RAFTQ_EINVAL, "deliver: malformed frame (group / from / entry count)");
    for (uint64_t i = 0; i < n_ents; ++i) {
      if (len - off < 16) return nfail(n, RAFTQ_EINVAL, "deliver: truncated entry header");
      uint64_t term;
      uint32_t l;
      std::memcpy(&term, p + off, 8);
      std::memcpy(&l, p + off + 8, 4);
      off += 16;
      const uint64_t padded = ((uint64_t)l + 7) / 8 * 8;
      if (len - off < padded) return nfail(n, RAFTQ_EINVAL, "deliver: truncated entry payload");
      im.ents.push_back(Entry{term, std::string((const char*)p + off, l)});
      off += padded;
    }
    im.h._resv = 0;
    parsed.push_back(std::move(im));
  }
  std::lock_guard<std::mutex> lk(n->mu);
  if (!n->started || n->closed) {
    n->errtext = "deliver: node not running";
    return RAFTQ_ESTATE;
  }
  for (InMsg& im : parsed) n->inbound.push_back(std::move(im));
  return RAFTQ_OK;
}

int raftq_node_advance(raftq_node_t* n, uint64_t* n_published) {
  if (!n) return RAFTQ_EINVAL;
  std::lock_guard<std::mutex> turn(n->turn_mu);
  std::vector<InMsg> work;
  std::vector<std::pair<uint64_t, std::string>> props;
  uint32_t ticks = 0;
  {
    std::lock_guard<std::mutex> lk(n->mu);
    if (!n->started) return RAFTQ_ESTATE;
    if (n->error) return n->error;
    work.swap(n->inbound);
    props.swap(n->proposals);
    ticks = n->pending_ticks;
    n->pending_ticks = 0;
  }
  // The commit channels, the status mirror and the outbound queues are only written below, under
  // mu, one short critical section per phase.
  std::unique_lock<std::mutex> lk(n->mu);
  const uint64_t published0 = n->stats.entries_published;
  bool did = !work.empty() || !props.empty() || ticks != 0;

  // -- rc.node.Tick() (raft.go:223-224) for every group: the engine advances the clocks and says
  // which groups' election timers fired (MsgHup -> through Step) and which leaders owe a heartbeat
  std::vector<InMsg> hups;
  for (uint32_t t = 0; t < ticks; ++t) {
    lk.unlock();
    int rc = raftq_tick(n->h, nullptr);
    if (rc == RAFTQ_OK) rc = raftq_read_tick(n->h, n->action.data(), nullptr, nullptr);
    if (rc != RAFTQ_OK) return poison(n, rc, "tick");
    lk.lock();
    for (uint64_t gi = 0; gi < n->G; ++gi) {
      if (n->action[gi] == 1) {
        InMsg im;
        im.h = header(n, gi, RAFTQ_MSG_HUP, 0);
        hups.push_back(std::move(im));
      } else if (n->action[gi] == 2 && n->groups[gi].role == RAFTQ_ROLE_LEADER) {
        bcast_heartbeat(n, gi, n->groups[gi]);  // stepLeader MsgBeat has no state effect: host only
      }
    }
  }
  if (!hups.empty()) {
    for (InMsg& im : work) hups.push_back(std::move(im));
    work.swap(hups);
  }

  // -- rc.Process -> Step, in rounds.  A message that changes a group's log (MsgApp, MsgProp) must
  // be the last one of its group in a Step batch: what follows it has to see the new log tail.
  std::vector<InMsg> batch, deferred;
  std::unordered_set<uint64_t> blocked, dirty;
  while (!work.empty()) {
    batch.clear();
    deferred.clear();
    blocked.clear();
    dirty.clear();
    for (InMsg& im : work) {
      if (blocked.count(im.h.group)) {
        deferred.push_back(std::move(im));
        continue;
      }
      if (im.h.type == RAFTQ_MSG_PROP || im.h.type == RAFTQ_MSG_APP) blocked.insert(im.h.group);
      batch.push_back(std::move(im));
    }
    // MsgProp never reaches Step; everything else does, in arrival order
    std::vector<size_t> idx;  // batch position of each stepped message
    raftq_msg_t* staged = nullptr;
    lk.unlock();
    size_t n_step = 0;
    for (const InMsg& im : batch) n_step += im.h.type != RAFTQ_MSG_PROP;
    const raftq_step_out_t* outs = nullptr;
    if (n_step) {
      int rc = raftq_step_stage(n->h, n_step, &staged);
      if (rc != RAFTQ_OK) return poison(n, rc, "step_stage");
      idx.reserve(n_step);
      for (size_t i = 0; i < batch.size(); ++i)
        if (batch[i].h.type != RAFTQ_MSG_PROP) {
          staged[idx.size()] = batch[i].h;
          idx.push_back(i);
        }
      rc = raftq_step_batch(n->h, staged, n_step, nullptr, nullptr);
      uint64_t n_out = 0;
      if (rc == RAFTQ_OK) rc = raftq_step_results(n->h, &outs, &n_out);
      if (rc != RAFTQ_OK) return poison(n, rc, "step_batch");
    }
    lk.lock();
    n->stats.msgs_stepped += n_step;
    // consequences, in arrival order (stepped results and proposals interleaved as they came)
    size_t k = 0;
    for (size_t i = 0; i < batch.size(); ++i) {
      InMsg& im = batch[i];
      if (im.h.type == RAFTQ_MSG_PROP) {
        Group& g = n->groups[im.h.group];
        if (handle_proposal(n, im.h.group, g, im.ents)) dirty.insert(im.h.group);
      } else {
        apply_result(n, outs[k++], im);
      }
    }
    for (uint64_t gi : dirty) {
      Group& g = n->groups[gi];
      raftq_log_delta_t d{gi, g.log.size(), g.term, 0};
      n->deltas.push_back(d);
    }
    if (int rc = flush_deltas(n, lk)) return poison(n, rc, "apply_log_deltas");
    for (uint64_t gi : dirty) bcast_append(n, gi, n->groups[gi]);
    work.swap(deferred);
  }

  // -- proposeC (raft.go:211-215)
  if (!props.empty()) {
    dirty.clear();
    for (auto& pr : props) {
      Group& g = n->groups[pr.first];
      std::vector<Entry> one;
      one.push_back(Entry{0, std::move(pr.second)});
      if (handle_proposal(n, pr.first, g, one)) dirty.insert(pr.first);
    }
    for (uint64_t gi : dirty) {
      Group& g = n->groups[gi];
      raftq_log_delta_t d{gi, g.log.size(), g.term, 0};
      n->deltas.push_back(d);
    }
    if (int rc = flush_deltas(n, lk)) return poison(n, rc, "apply_log_deltas");
    for (uint64_t gi : dirty) bcast_append(n, gi, n->groups[gi]);
  }
  if (did) n->stats.turns++;
  const uint64_t pub = n->stats.entries_published - published0;
  lk.unlock();
  if (pub) n->cv_commit.notify_all();
  if (n_published) *n_published = pub;
  return RAFTQ_OK;
}

int raftq_node_poll(raftq_node_t* n, uint32_t to_peer, void* buf, uint64_t cap, uint64_t* len) {
  if (!n || !len) return RAFTQ_EINVAL;
  *len = 0;
  if (to_peer >= n->N) return nfail(n, RAFTQ_EINVAL, "poll: peer out of range");
  if (cap && !buf) return nfail(n, RAFTQ_EINVAL, "poll: null buffer");
  std::lock_guard<std::mutex> lk(n->mu);
  auto& q = n->outbound[to_peer];
  size_t taken = 0;
  uint64_t off = 0;
  while (taken < q.size() && off + q[taken].size() <= cap) {
    std::memcpy((uint8_t*)buf + off, q[taken].data(), q[taken].size());
    off += q[taken].size();
    ++taken;
  }
  q.erase(q.begin(), q.begin() + taken);
  *len = off;
  if (taken == 0 && !q.empty()) {
    n->errtext = "poll: buffer smaller than the next frame";
    return RAFTQ_EINVAL;
  }
  return RAFTQ_OK;
}

int raftq_node_recv(raftq_node_t* n, uint64_t group, int timeout_ms, void* buf, uint32_t cap, uint32_t* len,
                    int* kind) {
  if (!n || !kind) return RAFTQ_EINVAL;
  if (group >= n->G) return nfail(n, RAFTQ_EINVAL, "recv: group out of range");
  std::unique_lock<std::mutex> lk(n->mu);
  Group& g = n->groups[group];
  auto ready = [&] { return g.qhead < g.q.size() || n->closed; };
  if (!ready()) {
    if (timeout_ms < 0) n->cv_commit.wait(lk, ready);
    else if (timeout_ms > 0) n->cv_commit.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready);
  }
  if (g.qhead < g.q.size()) {
    Item& it = g.q[g.qhead];
    *kind = it.kind;
    if (len) *len = (uint32_t)it.data.size();
    if (buf && cap) std::memcpy(buf, it.data.data(), std::min<size_t>(cap, it.data.size()));
    g.qhead++;
    if (g.qhead == g.q.size()) {
      g.q.clear();
      g.qhead = 0;
    }
    return RAFTQ_OK;
  }
  if (len) *len = 0;
  *kind = n->closed ? RAFTQ_NODE_CLOSED : RAFTQ_NODE_TIMEOUT;
  return RAFTQ_OK;
}

int raftq_node_status(raftq_node_t* n, uint64_t group, raftq_node_status_t* st) {
  if (!n || !st) return RAFTQ_EINVAL;
  if (group >= n->G) return nfail(n, RAFTQ_EINVAL, "status: group out of range");
  std::lock_guard<std::mutex> lk(n->mu);
  const Group& g = n->groups[group];
  std::memset(st, 0, sizeof(*st));
  st->term = g.term;
  st->commit = g.committed;
  st->last_index = g.log.size();
  st->applied = g.applied;
  st->lead = g.lead;
  st->vote = g.vote;
  st->role = g.role;
  return RAFTQ_OK;
}

int raftq_node_stats(raftq_node_t* n, raftq_node_stats_t* st) {
  if (!n || !st) return RAFTQ_EINVAL;
  std::lock_guard<std::mutex> lk(n->mu);
  *st = n->stats;
  return RAFTQ_OK;
}

int raftq_node_entry(raftq_node_t* n, uint64_t group, uint64_t index, void* buf, uint32_t cap, uint32_t* len,
                     uint64_t* term) {
  if (!n) return RAFTQ_EINVAL;
  if (group >= n->G) return nfail(n, RAFTQ_EINVAL, "entry: group out of range");
  std::lock_guard<std::mutex> lk(n->mu);
  const Group& g = n->groups[group];
  if (index == 0 || index > g.log.size()) {
    n->errtext = "entry: index out of range";
    return RAFTQ_EINVAL;
  }
  const Entry& e = g.log[index - 1];
  if (len) *len = (uint32_t)e.data.size();
  if (term) *term = e.term;
  if (buf && cap) std::memcpy(buf, e.data.data(), std::min<size_t>(cap, e.data.size()));
  return RAFTQ_OK;
}

raftq_t* raftq_node_engine(raftq_node_t* n) { return n ? n->h : nullptr; }

int raftq_node_close(raftq_node_t* n) {
  if (!n) return RAFTQ_EINVAL;
  std::lock_guard<std::mutex> turn(n->turn_mu);
  std::lock_guard<std::mutex> lk(n->mu);
  n->closed = true;
  n->cv_commit.notify_all();
  return n->error;  // 0 = the nil error of `return <-rp.ErrorC` (raftpipe.go:16)
}

int raftq_node_error(const raftq_node_t* n) { return n ? n->error : RAFTQ_EINVAL; }

const char* raftq_node_last_error(const raftq_node_t* n) { return n ? n->errtext.c_str() : "null node"; }

void raftq_node_destroy(raftq_node_t* n) {
  if (!n) return;
  raftq_node_close(n);
  raftq_destroy(n->h);
  delete n;
}

}  // extern "C"
