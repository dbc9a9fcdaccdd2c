// raftq_pipe.cpp -- multi-group propose -> commit pipeline over the raftq C-ABI
// (include/raftq_pipe.h).  The C++ stand-in for the Go batching goroutine in
// go/raftq/batcher.go; keeps the per-group contract of the reference's
// raftPipe / newRaftNode (raftpipe.go:3-17, raft.go:57-62, 82-96, 122-134).
//
// This node leads every group it drives (peer slot 0).  All quorum arithmetic
// is done by the GPU sweep (raftq_cycle); nothing here computes a commit index.
#include "raftq_pipe.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "raftq.h"

namespace {

struct Entry {
  uint64_t term;
  std::string data;
};

struct Item {
  int kind;  // RAFTQ_PIPE_ENTRY / RAFTQ_PIPE_SENTINEL
  std::string data;
};

struct Group {
  std::vector<Entry> log;  // log[i] holds index i + 1
  uint64_t committed = 0;
  uint64_t term = 0;
  std::vector<Item> q;     // commit channel (FIFO: q[qhead..])
  size_t qhead = 0;
};

}  // namespace

struct raftq_pipe {
  raftq_t* h = nullptr;
  uint64_t G = 0;
  uint32_t N = 0;
  std::vector<Group> groups;
  std::mutex mu;                     // guards everything below
  std::condition_variable cv_commit; // commit channels got items / closed
  std::condition_variable cv_work;   // batching thread: work pending / closing
  // the turn's acks, in the engine's 16-byte record whenever the group ids fit 32 bits (every realistic G):
  // a third fewer bytes over PCIe per turn; ranges are checked where the records are made (propose / process),
  // so the engine ingests them in one pass (RAFTQ_CYCLE_TRUSTED)
  bool packed = true;
  std::vector<raftq_delta_t> pending;
  std::vector<raftq_delta16_t> pending16;
  std::vector<raftq_append_t> outbox;
  std::chrono::steady_clock::time_point first_pending;
  bool started = false, closed = false;
  int error = 0;
  std::string errtext;
  std::mutex flush_mu;               // one batching turn at a time
  std::vector<raftq_advance_t> advbuf;
  std::vector<raftq_delta_t> turn;
  std::vector<raftq_delta16_t> turn16;
  std::thread worker;
  uint32_t max_batch = 1 << 16, max_wait_us = 200;
  uint64_t turns = 0;
};

namespace {

int pfail(raftq_pipe_t* p, int code, const std::string& msg) {
  if (p) {
    std::lock_guard<std::mutex> lk(p->mu);
    p->errtext = msg;
  }
  return code;
}

// writeError (raft.go:136-142): record the error and close the commit side
void poison(raftq_pipe_t* p, int code, const std::string& msg) {
  std::lock_guard<std::mutex> lk(p->mu);
  if (p->error == 0) {
    p->error = code;
    p->errtext = msg;
  }
  p->closed = true;
  p->cv_commit.notify_all();
  p->cv_work.notify_all();
}

// publishEntries (raft.go:82-96): committed entries (from, to] of one group go
// to its commit channel; empty payloads (the leader's no-op) are skipped.
void publish_locked(Group& g, uint64_t from, uint64_t to) {
  for (uint64_t idx = from + 1; idx <= to && idx <= g.log.size(); ++idx) {
    const Entry& e = g.log[idx - 1];
    if (e.data.empty()) continue;
    g.q.push_back(Item{RAFTQ_PIPE_ENTRY, e.data});
  }
}

int flush_turn(raftq_pipe_t* p, uint64_t* n_advanced) {
  std::lock_guard<std::mutex> turn_lk(p->flush_mu);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->started) return RAFTQ_ESTATE;
    if (p->error) return p->error;
    p->turn.clear();
    p->turn.swap(p->pending);
    p->turn16.clear();
    p->turn16.swap(p->pending16);
  }
  uint64_t n_adv = 0;
  const unsigned flags = RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED | RAFTQ_CYCLE_TRUSTED;
  int rc;
  bool overflowed = false;                      // packed, contiguous and longer than the buffer: the 24-byte list instead
  const raftq_advance16_t* seg_recs = nullptr;  // packed: the list is read where the turn's sweep left it, a segment per tile
  const uint32_t* seg_counts = nullptr;
  uint32_t n_segs = 0;
  uint64_t seg_stride = 0;
  if (p->packed) {
    // (cap only matters to a turn that cannot leave segments -- a handle of more than 4M groups -- and produces the contiguous
    // list: that one is bounded by the pipe's buffer as before, and taken again through raftq_collect_changed when it overflows)
    rc = raftq_cycle_packed(p->h, p->turn16.data(), p->turn16.size(), nullptr, 0, flags | RAFTQ_CYCLE_SEGMENTED, nullptr, p->advbuf.size(), &n_adv,
                            nullptr);
    if (rc == RAFTQ_OK) rc = raftq_last_advance_segments(p->h, &seg_recs, &seg_counts, &n_segs, &seg_stride);
    if (rc == RAFTQ_OK && n_segs == 1 && seg_counts[0] < n_adv) {
      p->advbuf.resize(std::min<uint64_t>(p->G, std::max<uint64_t>(n_adv, p->advbuf.size() * 2)));
      rc = raftq_collect_changed(p->h, p->advbuf.data(), p->advbuf.size(), &n_adv);
      overflowed = true;
    }
  } else {
    rc = raftq_cycle(p->h, p->turn.data(), p->turn.size(), nullptr, 0, flags, p->advbuf.data(), p->advbuf.size(), &n_adv,
                     nullptr);
    if (rc == RAFTQ_OK && n_adv > p->advbuf.size()) {
      p->advbuf.resize(std::min<uint64_t>(p->G, std::max<uint64_t>(n_adv, p->advbuf.size() * 2)));
      rc = raftq_collect_changed(p->h, p->advbuf.data(), p->advbuf.size(), &n_adv);
    }
  }
  if (rc != RAFTQ_OK) {
    const char* m = raftq_last_error(p->h);
    poison(p, rc, std::string("batching turn failed: ") + (m ? m : "?"));
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(p->mu);
    auto advance_to = [&](uint64_t gi, uint64_t nc) {
      Group& g = p->groups[gi];
      publish_locked(g, g.committed, nc);  // the pipe's own cursor IS the old commit index of the record
      g.committed = nc;
    };
    if (p->packed && !overflowed) {
      for (uint32_t sgm = 0; sgm < n_segs; ++sgm) {
        const raftq_advance16_t* r = seg_recs + (uint64_t)sgm * seg_stride;
        for (uint32_t i = 0; i < seg_counts[sgm]; ++i) advance_to(r[i].group, r[i].new_commit);
      }
    } else {
      for (uint64_t i = 0; i < n_adv; ++i) advance_to(p->advbuf[i].group, p->advbuf[i].new_commit);
    }
    p->turns++;
  }
  if (n_adv) p->cv_commit.notify_all();
  if (n_advanced) *n_advanced = n_adv;
  return RAFTQ_OK;
}

void worker_loop(raftq_pipe_t* p) {
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->closed || !p->pending.empty() || !p->pending16.empty(); });
      if (p->closed) return;
      // let the batch fill: until max_batch messages or max_wait_us after the first one
      const auto deadline = p->first_pending + std::chrono::microseconds(p->max_wait_us);
      p->cv_work.wait_until(lk, deadline, [&] { return p->closed || p->pending.size() + p->pending16.size() >= p->max_batch; });
      if (p->closed) return;
    }
    if (flush_turn(p, nullptr) != RAFTQ_OK) return;
  }
}

void push_delta_locked(raftq_pipe_t* p, uint64_t group, uint32_t peer, uint64_t match) {
  if (p->pending.empty() && p->pending16.empty()) p->first_pending = std::chrono::steady_clock::now();
  if (p->packed) {
    raftq_delta16_t d;
    d.match = match;
    d.group = (uint32_t)group;
    d.peer = peer;
    p->pending16.push_back(d);
  } else {
    raftq_delta_t d;
    d.group = group;
    d.match = match;
    d.peer = peer;
    d._pad = 0;
    p->pending.push_back(d);
  }
  const size_t n = p->pending.size() + p->pending16.size();
  if (n == 1 || n >= p->max_batch) p->cv_work.notify_one();
}

}  // namespace

extern "C" {

int raftq_pipe_create(int device, uint64_t n_groups, uint32_t n_peers, raftq_pipe_t** out) {
  if (!out) return RAFTQ_EINVAL;
  *out = nullptr;
  raftq_pipe_t* p = new (std::nothrow) raftq_pipe();
  if (!p) return RAFTQ_ENOMEM;
  const int rc = raftq_create(device, n_groups, n_peers, &p->h);
  if (rc != RAFTQ_OK) {
    delete p;
    return rc;  // text is in raftq_last_error(NULL)
  }
  p->G = n_groups;
  p->N = n_peers;
  try {
    p->groups.resize(n_groups);
    p->advbuf.resize(std::min<uint64_t>(n_groups, 1 << 16));
    p->packed = n_groups <= (1ull << 32);
  } catch (...) {
    raftq_destroy(p->h);
    delete p;
    return RAFTQ_ENOMEM;
  }
  *out = p;
  return RAFTQ_OK;
}

int raftq_pipe_replay(raftq_pipe_t* p, uint64_t group, const uint64_t* terms, const void* const* data,
                      const uint32_t* lens, uint64_t n) {
  if (!p) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "replay: group out of range");
  if (n && (!terms || !data || !lens)) return pfail(p, RAFTQ_EINVAL, "replay: null argument");
  std::lock_guard<std::mutex> lk(p->mu);
  if (p->started) {
    p->errtext = "replay: pipe already started";
    return RAFTQ_ESTATE;
  }
  Group& g = p->groups[group];
  uint64_t prev = g.log.empty() ? 1 : g.log.back().term;
  for (uint64_t i = 0; i < n; ++i) {
    if (terms[i] < prev) {
      p->errtext = "replay: terms must be non-decreasing and >= 1";
      return RAFTQ_EINVAL;
    }
    prev = terms[i];
    g.log.push_back(Entry{terms[i], std::string((const char*)data[i], lens[i])});
  }
  return RAFTQ_OK;
}

int raftq_pipe_start(raftq_pipe_t* p, uint32_t max_batch, uint32_t max_wait_us, int background) {
  if (!p) return RAFTQ_EINVAL;
  std::vector<uint64_t> match, committed, cur_term, first_idx;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->started) {
      p->errtext = "start: already started";
      return RAFTQ_ESTATE;
    }
    if (max_batch) p->max_batch = max_batch;
    p->max_wait_us = max_wait_us;
    try {
      match.assign((size_t)p->N * p->G, 0);
      committed.assign(p->G, 0);
      cur_term.assign(p->G, 0);
      first_idx.assign(p->G, 0);
    } catch (...) {
      p->errtext = "start: host allocation failed";
      return RAFTQ_ENOMEM;
    }
    for (uint64_t gi = 0; gi < p->G; ++gi) {
      Group& g = p->groups[gi];
      // replayWAL: every logged entry goes out, then the nil sentinel (raft.go:129-132)
      const uint64_t replayed = g.log.size();
      publish_locked(g, 0, replayed);
      g.q.push_back(Item{RAFTQ_PIPE_SENTINEL, std::string()});
      g.committed = replayed;
      // becomeLeader: new term, append the empty entry of that term
      g.term = (g.log.empty() ? 0 : g.log.back().term) + 1;
      g.log.push_back(Entry{g.term, std::string()});
      const uint64_t last = g.log.size();
      raftq_append_t ap;
      ap.group = gi;
      ap.index = last;
      ap.term = g.term;
      ap.len = 0;
      ap._pad = 0;
      p->outbox.push_back(ap);
      match[gi] = last;  // peer slot 0 = this node: Match == its last index
      committed[gi] = replayed;
      cur_term[gi] = g.term;
      first_idx[gi] = last;
    }
  }
  int rc = raftq_load_match(p->h, match.data(), committed.data());
  if (rc == RAFTQ_OK) rc = raftq_load_terms(p->h, cur_term.data(), first_idx.data());
  if (rc != RAFTQ_OK) {
    const char* m = raftq_last_error(p->h);
    poison(p, rc, std::string("start: ") + (m ? m : "?"));
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->started = true;
  }
  p->cv_commit.notify_all();
  if (background) p->worker = std::thread(worker_loop, p);
  return RAFTQ_OK;
}

int raftq_pipe_propose(raftq_pipe_t* p, uint64_t group, const void* data, uint32_t len) {
  if (!p) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "propose: group out of range");
  if (len && !data) return pfail(p, RAFTQ_EINVAL, "propose: null payload");
  std::lock_guard<std::mutex> lk(p->mu);
  if (!p->started || p->closed) {
    p->errtext = p->closed ? "propose: pipe is closed" : "propose: pipe not started";
    return RAFTQ_ESTATE;
  }
  Group& g = p->groups[group];
  g.log.push_back(Entry{g.term, std::string((const char*)data, len)});
  const uint64_t last = g.log.size();
  raftq_append_t ap;
  ap.group = group;
  ap.index = last;
  ap.term = g.term;
  ap.len = len;
  ap._pad = 0;
  p->outbox.push_back(ap);
  push_delta_locked(p, group, 0, last);  // the leader's own Match follows its log
  return RAFTQ_OK;
}

int raftq_pipe_process_app_resp(raftq_pipe_t* p, uint64_t group, uint32_t from, uint64_t index) {
  if (!p) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "process: group out of range");
  if (from == 0 || from >= p->N) return pfail(p, RAFTQ_EINVAL, "process: `from` must be a follower slot 1..N-1");
  std::lock_guard<std::mutex> lk(p->mu);
  if (!p->started || p->closed) {
    p->errtext = p->closed ? "process: pipe is closed" : "process: pipe not started";
    return RAFTQ_ESTATE;
  }
  if (index > p->groups[group].log.size()) {
    p->errtext = "process: ack beyond the leader's last index";
    return RAFTQ_EINVAL;
  }
  push_delta_locked(p, group, from, index);
  return RAFTQ_OK;
}

int raftq_pipe_flush(raftq_pipe_t* p, uint64_t* n_advanced) {
  if (!p) return RAFTQ_EINVAL;
  return flush_turn(p, n_advanced);
}

int raftq_pipe_recv(raftq_pipe_t* p, uint64_t group, int timeout_ms, void* buf, uint32_t cap, uint32_t* len,
                    int* kind) {
  if (!p || !kind) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "recv: group out of range");
  std::unique_lock<std::mutex> lk(p->mu);
  Group& g = p->groups[group];
  auto ready = [&] { return g.qhead < g.q.size() || p->closed; };
  if (!ready()) {
    if (timeout_ms < 0) p->cv_commit.wait(lk, ready);
    else if (timeout_ms > 0) p->cv_commit.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready);
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
  *kind = p->closed ? RAFTQ_PIPE_CLOSED : RAFTQ_PIPE_TIMEOUT;
  return RAFTQ_OK;
}

int raftq_pipe_take_appends(raftq_pipe_t* p, raftq_append_t* out, uint64_t cap, uint64_t* n) {
  if (!p || !n) return RAFTQ_EINVAL;
  if (cap && !out) return pfail(p, RAFTQ_EINVAL, "take_appends: null out");
  std::lock_guard<std::mutex> lk(p->mu);
  const uint64_t take = std::min<uint64_t>(cap, p->outbox.size());
  if (take) std::memcpy(out, p->outbox.data(), take * sizeof(raftq_append_t));
  p->outbox.erase(p->outbox.begin(), p->outbox.begin() + take);
  *n = take;
  return RAFTQ_OK;
}

int raftq_pipe_entry(raftq_pipe_t* p, uint64_t group, uint64_t index, void* buf, uint32_t cap, uint32_t* len,
                     uint64_t* term) {
  if (!p) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "entry: group out of range");
  std::lock_guard<std::mutex> lk(p->mu);
  const Group& g = p->groups[group];
  if (index == 0 || index > g.log.size()) {
    p->errtext = "entry: index out of range";
    return RAFTQ_EINVAL;
  }
  const Entry& e = g.log[index - 1];
  if (len) *len = (uint32_t)e.data.size();
  if (term) *term = e.term;
  if (buf && cap) std::memcpy(buf, e.data.data(), std::min<size_t>(cap, e.data.size()));
  return RAFTQ_OK;
}

int raftq_pipe_last_index(raftq_pipe_t* p, uint64_t group, uint64_t* index) {
  if (!p || !index) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "group out of range");
  std::lock_guard<std::mutex> lk(p->mu);
  *index = p->groups[group].log.size();
  return RAFTQ_OK;
}

int raftq_pipe_committed(raftq_pipe_t* p, uint64_t group, uint64_t* index) {
  if (!p || !index) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "group out of range");
  std::lock_guard<std::mutex> lk(p->mu);
  *index = p->groups[group].committed;
  return RAFTQ_OK;
}

int raftq_pipe_term(raftq_pipe_t* p, uint64_t group, uint64_t* term) {
  if (!p || !term) return RAFTQ_EINVAL;
  if (group >= p->G) return pfail(p, RAFTQ_EINVAL, "group out of range");
  std::lock_guard<std::mutex> lk(p->mu);
  *term = p->groups[group].term;
  return RAFTQ_OK;
}

int raftq_pipe_close(raftq_pipe_t* p) {
  if (!p) return RAFTQ_EINVAL;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->closed = true;
  }
  p->cv_work.notify_all();
  p->cv_commit.notify_all();
  if (p->worker.joinable()) p->worker.join();
  std::lock_guard<std::mutex> lk(p->mu);
  return p->error;  // 0 = the nil error of `return <-rp.ErrorC`
}

int raftq_pipe_error(const raftq_pipe_t* p) { return p ? p->error : RAFTQ_EINVAL; }

const char* raftq_pipe_last_error(const raftq_pipe_t* p) { return p ? p->errtext.c_str() : "null pipe"; }

void raftq_pipe_destroy(raftq_pipe_t* p) {
  if (!p) return;
  raftq_pipe_close(p);
  raftq_destroy(p->h);
  delete p;
}

}  // extern "C"
