// raftq_node.cpp -- one raft node for G groups (include/raftq_node.h): the C++ stand-in for
// the reference's raftNode + serveChannels (raft.go:36-78, 204-246), G-fold.
//
// Division of labour.  Every consensus decision is the GPU engine's: raftq_step_batch (Step for
// all payload-free message kinds + the MsgApp header), raftq_tick, raftq_apply_log_deltas.  This
// file owns what the reference's raftNode owns around raft.Node: the log entries
// (raft.MemoryStorage, raft.go:70), the replication cursor (Progress.Next), queues and channels.
// It computes no quorum, no vote tally and no commit index.
#include "raftq_node.h"

#include "raftq_wire.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#endif

namespace {

// 2 MB-aligned memory for the node's arena and pool chunks, with a transparent huge page asked for per 2 MB: when 32,768
// groups outgrow a block size in the same turn the pool hands out 8-32 MB of memory nobody has touched yet, and with 4 KB
// pages that turn takes 8,000 page faults -- measured as waves of 8 / 11 / 15 ms among waves of 5.4 (bench.py -> node,
// ms_per_wave_each; profiles/r03/node_chunk_ab.txt: 4.98 -> 5.66e6 proposals/s with huge pages).  RAFTQ_NODE_THP=0 turns the
// request off (a VM whose huge-page fault compacts memory synchronously took twice as long per turn with it).
// (Its own mapping, not aligned_alloc: memory the allocator has had before comes back with its 4 KB pages already in place, and
// the request then changes nothing until khugepaged gets round to it -- inside bench.py, after the other legs had freed
// hundreds of MB, the node leg ran as if huge pages were off.)
constexpr size_t kHugePage = (size_t)2 << 20;
inline size_t huge_round(size_t bytes) { return (bytes + kHugePage - 1) / kHugePage * kHugePage; }
inline void* huge_alloc(size_t bytes) {
  bytes = huge_round(bytes);
#if defined(__linux__)
  static const bool thp = [] {
    const char* e = std::getenv("RAFTQ_NODE_THP");
    return !(e && e[0] == '0');
  }();
  char* raw = (char*)mmap(nullptr, bytes + kHugePage, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == (char*)MAP_FAILED) return nullptr;
  char* p = (char*)(((uintptr_t)raw + kHugePage - 1) & ~(uintptr_t)(kHugePage - 1));
  if (p != raw) (void)munmap(raw, (size_t)(p - raw));
  const size_t tail = (size_t)(raw + bytes + kHugePage - (p + bytes));
  if (tail) (void)munmap(p + bytes, tail);
  if (thp) (void)madvise(p, bytes, MADV_HUGEPAGE);
  return p;
#else
  return std::aligned_alloc(kHugePage, bytes);
#endif
}
inline void huge_free(void* p, size_t bytes) {
  if (!p) return;
#if defined(__linux__)
  (void)munmap(p, huge_round(bytes));
#else
  (void)bytes;
  std::free(p);
#endif
}

// Entry payloads live in the node's arena: append-only chunks that never move, so a log entry, an item on a commit
// channel and an entry of an outbound message are (pointer, length) views and nothing on the per-message path
// allocates (round 1 kept a std::string per entry, per queued item and per decoded message: the allocator was a third
// of raftq_node_advance's host time).  A truncated suffix leaves its bytes behind; the log is never compacted either.
struct Arena {
  static constexpr size_t kChunk = (size_t)4 << 20;
  struct Chunk {
    char* p;
    size_t bytes;
  };
  std::vector<Chunk> chunks;
  Arena() = default;
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  ~Arena() {
    for (const Chunk& c : chunks) huge_free(c.p, c.bytes);
  }
  char* cur = nullptr;
  size_t left = 0;
  // a stable copy of [src, src + len); nullptr when memory ran out
  const char* put(const void* src, size_t len) {
    if (len == 0) return "";
    char* at;
    if (len > left) {
      const size_t cap = std::max(len, kChunk);
      char* c = (char*)huge_alloc(cap);
      if (!c) return nullptr;
      try {
        chunks.push_back(Chunk{c, cap});
      } catch (...) {
        huge_free(c, cap);
        return nullptr;
      }
      if (cap - len < left) {  // a large payload gets a chunk of its own; the open chunk stays open
        std::memcpy(c, src, len);
        return c;
      }
      cur = c;
      left = cap;
    }
    at = cur;
    std::memcpy(at, src, len);
    cur += len;
    left -= len;
    return at;
  }
};

struct Entry {  // a view: the log's entries point into the arena, a message's into the turn's receive buffer
  uint64_t term;
  const char* data;
  uint32_t len;
};

struct Item {  // 16 bytes
  const char* data;
  uint32_t len;
  int32_t kind;
};

// The node's own allocator for the per-group arrays (logs, commit channels): power-of-two blocks carved from 2 MB chunks,
// freed blocks recycled per size, everything returned at once when the node goes.  Not thread-safe and it need not be: the
// arrays only grow inside advance() (one at a time) or before the node starts.  glibc's realloc was the single most
// expensive thing in the turn: the blocks belong to the arena of the thread that created the node, every node's turn
// thread fought for that one lock (4,000 cycles per growing push_back with three nodes in a process, measured).
struct Pool {
  static constexpr size_t kChunk = (size_t)2 << 20;
  static constexpr int kClasses = 15;  // 64 B .. 1 MB
  std::vector<void*> chunks, big;
  char* cur = nullptr;
  size_t left = 0;
  void* free_list[kClasses] = {};
  Pool() = default;
  Pool(const Pool&) = delete;
  Pool& operator=(const Pool&) = delete;
  ~Pool() {
    for (void* c : chunks) huge_free(c, kChunk);
    for (void* b : big) std::free(b);
  }
  // a block of at least `bytes`; *got = its real size.  Throws std::bad_alloc.
  void* alloc(size_t bytes, size_t* got) {
    int c = 0;
    size_t sz = 64;
    while (sz < bytes) sz <<= 1, ++c;
    if (c >= kClasses) {  // larger than any class: its own allocation
      big.reserve(big.size() + 1);  // (so that the push below cannot throw with the block in hand)
      void* p = std::malloc(bytes);
      if (!p) throw std::bad_alloc();
      big.push_back(p);
      *got = bytes;
      return p;
    }
    *got = sz;
    if (void* p = free_list[c]) {
      std::memcpy(&free_list[c], p, sizeof(void*));
      return p;
    }
    if (left < sz) {
      chunks.reserve(chunks.size() + 1);  // (so that the push below cannot throw with the chunk in hand)
      void* chunk = huge_alloc(kChunk);
      if (!chunk) throw std::bad_alloc();
      chunks.push_back(chunk);
      cur = (char*)chunk;
      left = kChunk;
    }
    void* p = cur;
    cur += sz;
    left -= sz;
    return p;
  }
  // `bytes`: anything that rounds up to the block's size (an array hands back capacity x element size)
  void release(void* p, size_t bytes) {
    if (!p || bytes == 0) return;
    int c = 0;
    size_t sz = 64;
    while (sz < bytes) sz <<= 1, ++c;
    if (c >= kClasses) {  // a `big` block: handed back to malloc (there are few of them: a log of > 40,000 entries)
      for (size_t i = big.size(); i-- > 0;)
        if (big[i] == p) {
          std::free(p);
          big[i] = big.back();
          big.pop_back();
          break;
        }
      return;
    }
    std::memcpy(p, &free_list[c], sizeof(void*));
    free_list[c] = p;
  }
};

// A growable array in 16 bytes (std::vector takes 24) for the two arrays every group has, so that both headers share the
// group's one hot cache line; its blocks come from the node's Pool (which frees them).  Trivially copyable elements only.
template <typename T> struct TinyVec {
  T* p = nullptr;
  uint32_t n = 0, cap = 0;
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T& back() { return p[n - 1]; }
  const T& back() const { return p[n - 1]; }
  T* data() { return p; }
  void clear() { n = 0; }
  void truncate(size_t k) {
    if (k < n) n = (uint32_t)k;
  }
  void push_back(Pool& pool, const T& v) {
    if (n == cap) {
      if (cap >= 0x40000000u) throw std::bad_alloc();
      size_t got = 0;
      T* q = (T*)pool.alloc(std::max<size_t>((size_t)cap * 2 * sizeof(T), 128), &got);
      if (n) std::memcpy(q, p, (size_t)n * sizeof(T));
      pool.release(p, (size_t)cap * sizeof(T));  // the block's class is what capacity x element size rounds up to
      p = q;
      cap = (uint32_t)(got / sizeof(T));
    }
    p[n++] = v;
  }
};

// A received (or locally raised) message is a raftq_wire_msg_t: the decoder's record, which is also what Step takes (the
// layouts agree in every field Step reads, raftq_wire.h).  The turn's lists hold POSITIONS: a decoded message's index in
// the decoder's output, or kLocal | index into the turn's locally raised ones (MsgHup) -- nothing is copied per round.
constexpr uint32_t kLocal = 0x80000000u;

// proposals queued by raftq_node_propose[_batch] until the next advance(): one blob, no string per proposal
struct PropBuf {
  std::vector<uint64_t> group, off;  // off has size() + 1 entries once anything is queued
  std::vector<char> blob;
  size_t size() const { return group.size(); }
  void clear() {
    group.clear();
    off.clear();
    blob.clear();
  }
  // room for k more proposals of `bytes` payload bytes in all; throws std::bad_alloc with nothing changed
  void reserve_more(size_t k, size_t bytes) {
    auto grow = [](auto& v, size_t want) {  // geometric, as push_back would: reserve(size + 1) per call is quadratic
      if (v.capacity() < want) v.reserve(std::max(want, v.capacity() * 2));
    };
    grow(group, group.size() + k);
    grow(off, off.size() + k + 1);
    grow(blob, blob.size() + bytes);
  }
  void add(uint64_t g, const void* data, size_t len) {  // after reserve_more: cannot fail half-way
    if (off.empty()) off.push_back(0);
    group.push_back(g);
    if (len) blob.insert(blob.end(), (const char*)data, (const char*)data + len);
    off.push_back(blob.size());
  }
};

// Growable page-locked byte buffer (raftq_host_alloc) for everything the codecs read or fill: the GPU
// moves such memory by direct DMA.  Measured on MI355X (tests/soak/wire_fresh_buffers.py): a 64K-frame decode
// into freshly allocated pageable arrays takes 2.8 ms, into the same pinned arrays every turn 0.21 ms.
struct PinBuf {
  uint8_t* p = nullptr;
  size_t size = 0, cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { raftq_host_free(p); }
  bool reserve(size_t want) {
    if (want <= cap) return true;
    const size_t nc = std::max<size_t>(std::max(want, cap * 2), (size_t)1 << 16);
    void* q = nullptr;
    if (raftq_host_alloc(&q, nc) != RAFTQ_OK) return false;
    if (size) std::memcpy(q, p, size);
    raftq_host_free(p);
    p = (uint8_t*)q;
    cap = nc;
    return true;
  }
  bool append(const void* src, size_t n) {
    if (!reserve(size + n)) return false;
    if (n) std::memcpy(p + size, src, n);
    size += n;
    return true;
  }
  void clear() { size = 0; }
  void swap(PinBuf& o) {
    std::swap(p, o.p);
    std::swap(size, o.size);
    std::swap(cap, o.cap);
  }
  template <typename T> T* as() { return (T*)p; }
  template <typename T> size_t count() const { return size / sizeof(T); }
};

// u64 offsets in page-locked memory: the codecs copy them to / from the device, and a copy from pageable memory goes
// through the runtime's staging buffer, synchronously
struct PinU64 {
  PinBuf b;
  size_t size() const { return b.size / 8; }
  bool empty() const { return b.size == 0; }
  uint64_t* data() { return (uint64_t*)b.p; }
  void clear() { b.size = 0; }
  void swap(PinU64& o) { b.swap(o.b); }
  bool resize(size_t k) {  // new elements are not initialised
    if (!b.reserve(k * 8)) return false;
    b.size = k * 8;
    return true;
  }
};

struct alignas(64) Group {
  // -- the cache line every message of the group touches
  uint64_t term = 0, committed = 0, applied = 0;
  TinyVec<Entry> log;  // log[i] holds index i + 1 (no compaction)
  TinyVec<Item> q;     // commit channel: q[qhead ..) waits for raftq_node_recv
  uint16_t lead = 0, vote = 0;  // 0 = None, else peer slot + 1
  uint8_t role = RAFTQ_ROLE_FOLLOWER;
  bool leading = false;    // this node keeps the group's Progress (raftq_node::prog): it is (or was just elected) leader
  bool wal_dirty = false;
  // -- the rest
  uint32_t qhead = 0;
  uint32_t hs_vote = 0;
  uint64_t hs_term = 0, hs_commit = 0;  // raftq_node_set_hard_state
  // WAL bookkeeping (raftq_node_wal_enable): what wal.Save has been handed so far
  uint64_t wal_upto = 0;                 // entries [1, wal_upto] are in the WAL as they stand in `log`
  uint64_t wal_term = 0, wal_commit = 0;  // the last HardState record written
  uint32_t wal_vote = 0;
};
static_assert(sizeof(Group) == 128 && offsetof(Group, qhead) == 64, "a group's hot state is one cache line");

// frames queued for one peer: rafthttp stream frames (u64 big-endian length | raftpb.Message), back to back
struct PeerQueue {
  std::string bytes;
  size_t head = 0;  // bytes before `head` were polled
  // where every frame of `bytes` ends (the encoder's offsets): raftq_node_forward hands them to the receiving node, which
  // then has nothing to scan.  Kept only while nothing was polled from the queue (`ends_ok`).
  std::vector<uint64_t> ends;
  bool ends_ok = true;
  // what raftq_node_forward took away comes back empty: a turn's 0.5 MB per peer is then written into memory the queue
  // already owns instead of a fresh allocation (and its page faults) every turn
  std::string spare;
  std::vector<uint64_t> spare_ends;
  void reset() {
    bytes.clear();
    ends.clear();
    head = 0;
    ends_ok = true;
  }
  void reuse_spares() {  // before appending to an empty queue
    if (bytes.capacity() < spare.capacity()) {
      spare.clear();
      bytes.swap(spare);
    }
    if (ends.capacity() < spare_ends.capacity()) {
      spare_ends.clear();
      ends.swap(spare_ends);
    }
  }
};

// this turn's messages for one peer, in the order of the sends, and an upper bound of their encoded size
struct OutLane {
  std::vector<raftq_wire_msg_t> msgs;
  uint64_t cap = 0;
};

constexpr uint64_t kMaxEntriesPerMsg = 1024;     // with kMaxBytesPerMsg: raft.Config.MaxSizePerMsg (raft.go:157)
constexpr uint64_t kMaxBytesPerMsg = 1 << 20;

}  // namespace

struct raftq_node {
  raftq_t* h = nullptr;
  uint64_t G = 0;
  uint32_t N = 0, self = 0;
  std::vector<Group> groups;
  std::mutex mu;  // guards in_bytes / proposals / pending_ticks / outbound / commit channels / status
  std::condition_variable cv_commit;
  PropBuf proposals, turn_props;  // queued / being worked through by advance()
  Arena arena;                    // entry payloads (logs, commit channels)
  Pool pool;                      // the groups' log and commit-channel arrays
  bool oom = false;               // an arena or queue allocation failed this turn: advance() ends in ENOMEM
  bool tail_appends = true;       // MsgApps are staged with RAFTQ_MSGF_ENTRIES (RAFTQ_NODE_TAIL_APPENDS=0: headers only, as round 2)
  bool deltas_nowait = true;      // RAFTQ_NODE_DELTAS_NOWAIT=0: every tail report is waited for
  bool split_wal = true;          // RAFTQ_NODE_SPLIT_WAL=0: raftq_wal_encode as a call of its own, waited for, ahead of the outbound marshal
  bool deltas_need_result = false;  // a report in n->deltas may move the commit index (a follower's): flush_deltas waits
  bool wal_begun = false;         // flush_wal_begin .. flush_wal_end
  size_t wal_inflight = 0;
  raftq_wal_counts_t wal_cnt{};
  // Round 6: what a leader is asked to propose goes to the device as two 16-byte records -- which group, which payload bytes -- and
  // appendEntry + bcastAppend (the N - 1 MsgApp headers, the entry headers) are the device's, written into the encoder's input
  // (raftq_propose_frames) -- for every group whose followers all have their Progress.Next at the log's tail, which is every group
  // outside a catch-up.  RAFTQ_NODE_PROPOSE_DEVICE=0: handle_proposal + bcast_append on the host for everybody, as round 5.
  bool propose_device = true;
  PinBuf prop_recs, prop_ents;           // raftq_prop_t[], raftq_prop_ent_t[] of the turn (page-locked: the kernels read them in place)
  std::vector<uint32_t> prop_slot;       // [group] index of the group's record in prop_recs, valid where prop_mark == the turn's epoch
  std::vector<uint32_t> prop_mark;       // [group] epoch | kind: the group proposes this turn through the device (fast) or the host (slow)
  uint64_t prop_cap = 0;                 // upper bound of the device-built frames' bytes, per addressee
  bool fuse_inbound = true;       // a turn's frames are decoded AND stepped by one submission (raftq_step_frames) whenever nothing was
                                  // raised locally ahead of them (RAFTQ_NODE_FUSE_INBOUND=0: raftq_wire_decode, then the staged rounds)
  bool broken = false;            // the engine's view of a log and the log itself disagree: advance() ends in ESTATE
  // the turn's decoded inbound entries and the bytes their payloads sit in (valid inside advance())
  const raftq_wire_ent_t* cur_ents = nullptr;
  const uint8_t* cur_bytes = nullptr;
  uint32_t pending_ticks = 0;
  std::vector<uint64_t> pending_hups, turn_hups;  // raftq_node_campaign: groups to raise MsgHup for at the next advance
  std::vector<PeerQueue> outbound;  // [peer]
  // inbound stream frames as delivered (decoded on the GPU at the next advance)
  PinBuf in_bytes;
  PinU64 in_off;  // frame boundaries in in_bytes; empty or [0, ..., in_bytes.size]
  // this turn's outbound messages, one lane per addressee, marshalled in one raftq_wire_encode at the end of advance()
  std::vector<OutLane> out_lane;
  // Progress.Next / Progress.Match mirror of the groups this node leads: prog[(g * N + peer) * 2 + {0: Next, 1: Match}]
  std::vector<uint64_t> prog;
  PinBuf out_ents;  // raftq_wire_ent_t[]
  PinBuf out_pool;  // entry payload bytes
  bool out_oom = false;
  // advance()'s own scratch (only touched under turn_mu): the other half of the inbound double buffer,
  // decoded records, the sorted outbound batch and its stream
  PinBuf turn_bytes, turn_msgs, turn_ents, enc_msgs, enc_out, wal_recs, wal_pool, wal_enc;
  PinU64 turn_off, enc_off;
  std::vector<uint64_t> lane_first;
  // WAL (off unless raftq_node_wal_enable): encoded records waiting for raftq_node_wal_poll
  bool wal_on = false, wal_head_written = false;
  uint32_t wal_crc = 0;
  PeerQueue wal_out;
  // groups the WAL has something to say about this turn: a bitmap walked in group order (round 6: a vector in touch order that
  // flush_wal_begin SORTED -- 32K ids a turn, most of the 1.4 ms the WAL cost a turn of the one-node leg) and the range of words set
  std::vector<uint64_t> wal_bits;
  uint64_t wal_lo = ~0ull, wal_hi = 0, wal_n_dirty = 0;
  // entries of the last send_append, so that a broadcast marshals one copy of a shared suffix
  uint64_t shared_group = ~0ull, shared_first_idx = 0, shared_cnt = 0;
  uint32_t shared_ent_first = 0;
  // RAFTQ_PROFILE=1: host time of advance()'s phases, printed at destroy
  enum { kPhDecode, kPhInbound, kPhTick, kPhStage, kPhStep, kPhApply, kPhDeltas, kPhProps, kPhWal, kPhEncode, kPhDevDeltas,
         kPhDevEncode, kPhN };  // the last two: the device calls inside the deltas / props and the encode phases
  double prof[kPhN] = {0};
  struct PhaseClock* clock = nullptr;  // the running advance()'s phase clock (RAFTQ_PROFILE)
  uint64_t prof_turns = 0, prof_every = 0;  // RAFTQ_PROFILE_EVERY=k: print and reset every k turns
  bool profiling = false;
  bool started = false, closed = false;
  int error = 0;
  std::string errtext;
  std::mutex turn_mu;  // one advance() at a time
  raftq_node_stats_t stats{};
  // scratch of advance()
  std::vector<uint64_t> tick_list;  // MsgHup groups of a tick whose in-place list was too short (grown on demand); its size is the cap asked for
  std::vector<uint32_t> work, batch, deferred;  // positions (see kLocal)
  std::vector<raftq_wire_msg_t> local;          // the turn's locally raised messages (MsgHup)
  std::vector<Entry> ent_tmp;
  // per-group marks of one Step round: blocked (a log-changing message of the group is in this batch) and dirty
  // (its log grew); a mark is set when it equals the round's epoch -- no hashing, no clearing
  std::vector<uint32_t> blocked_mark, dirty_mark, defer_mark;  // defer_mark: a message of the group went to the next round
  std::vector<uint64_t> dirty_list;
  uint32_t epoch = 0;
  std::vector<raftq_log_delta_t> deltas;
  std::vector<uint64_t> delta_commit;
  uint64_t& next_of(uint64_t g, uint32_t peer) { return prog[(g * N + peer) * 2]; }
  uint64_t& match_of(uint64_t g, uint32_t peer) { return prog[(g * N + peer) * 2 + 1]; }
};

struct PhaseClock {  // accumulates wall time into n->prof[which] when RAFTQ_PROFILE is set
  raftq_node_t* n;
  int which;
  std::chrono::steady_clock::time_point t0;
  PhaseClock(raftq_node_t* node, int w) : n(node), which(w) {
    if (n->profiling) t0 = std::chrono::steady_clock::now();
    n->clock = this;
  }
  void next(int w) {
    if (!n->profiling) return;
    const auto t1 = std::chrono::steady_clock::now();
    n->prof[which] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    which = w;
    t0 = t1;
  }
  ~PhaseClock() {
    next(which);
    n->clock = nullptr;
  }
};

namespace {

// a device call inside a host phase: its time goes to its own bucket
struct DevCall {
  raftq_node_t* n;
  int back;
  DevCall(raftq_node_t* node, int bucket) : n(node), back(node->clock ? node->clock->which : 0) {
    if (n->clock) n->clock->next(bucket);
  }
  ~DevCall() {
    if (n->clock) n->clock->next(back);
  }
};

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

// r.send(m): queue one raftpb.Message for peer `to`.  Nothing is marshalled here -- the headers, entry ranges and
// payload bytes of the whole turn go through one raftq_wire_encode at the end of advance().  Returns the queued record
// (valid until the next send to the same peer): the caller sets what the kind carries beyond group / type / term.
raftq_wire_msg_t& send(raftq_node_t* n, uint32_t to, uint64_t group, uint8_t type, uint64_t term) {
  OutLane& lane = n->out_lane[to];
  lane.msgs.emplace_back();  // zeroed
  lane.cap += 160;           // 12 varint fields + tags + the empty snapshot
  raftq_wire_msg_t& m = lane.msgs.back();
  m.group = group;
  m.term = term;  // raft.send: every non-MsgProp message carries r.Term
  m.from = n->self;
  m.type = type;
  m.to = (uint8_t)to;
  n->stats.msgs_sent++;
  return m;
}

// the entries of the message queued last for `to`.  They carry explicit indices on the wire: m.index + 1 + k for MsgApp,
// 0 for a forwarded MsgProp.
void attach(raftq_node_t* n, uint32_t to, raftq_wire_msg_t& m, const Entry* ents, size_t n_ents) {
  if (n_ents == 0) return;
  OutLane& lane = n->out_lane[to];
  m.n_ents = (uint32_t)n_ents;
  // bcastAppend sends most followers the same suffix: the encoder takes arbitrary entry ranges, so the
  // second and later messages point at the first one's entries instead of queueing copies
  const bool app = m.type == RAFTQ_MSG_APP;
  if (app && n->shared_group == m.group && n->shared_first_idx == m.index + 1 && n->shared_cnt == n_ents) {
    m.ent_first = n->shared_ent_first;
    for (size_t i = 0; i < n_ents; ++i) lane.cap += 48 + ents[i].len;
    return;
  }
  m.ent_first = (uint32_t)n->out_ents.count<raftq_wire_ent_t>();
  if (app) {
    n->shared_group = m.group;
    n->shared_first_idx = m.index + 1;
    n->shared_cnt = n_ents;
    n->shared_ent_first = m.ent_first;
  }
  for (size_t i = 0; i < n_ents; ++i) {
    raftq_wire_ent_t e;
    e.term = ents[i].term;
    e.index = app ? m.index + 1 + i : 0;
    e.data_len = ents[i].len;
    e.data_off = e.data_len ? n->out_pool.size : 0;
    e.type = 0;
    lane.cap += 48 + ents[i].len;
    if (!n->out_pool.append(ents[i].data, ents[i].len) || !n->out_ents.append(&e, sizeof(e))) n->out_oom = true;
  }
}

void wal_touch(raftq_node_t* n, uint64_t gi, Group& g) {
  if (n->wal_on && !g.wal_dirty) {
    g.wal_dirty = true;
    const uint64_t w = gi >> 6;
    n->wal_bits[w] |= 1ull << (gi & 63);
    n->wal_lo = std::min(n->wal_lo, w);
    n->wal_hi = std::max(n->wal_hi, w);
    n->wal_n_dirty++;
  }
}

// a locally raised message (MsgHup): term 0 marks it local
raftq_wire_msg_t local_msg(const raftq_node_t* n, uint64_t group, uint8_t type) {
  raftq_wire_msg_t m;
  std::memset(&m, 0, sizeof(m));
  m.group = group;
  m.type = type;
  m.from = n->self;
  m.to = (uint8_t)n->self;
  return m;
}

// publishEntries (raft.go:82-96): entries (applied, upto] go to the commit channel; empty
// payloads (a new leader's no-op) are skipped.
void publish(raftq_node_t* n, Group& g, uint64_t upto) {
  upto = std::min<uint64_t>(upto, g.log.size());
  for (uint64_t idx = g.applied + 1; idx <= upto; ++idx) {
    const Entry& e = g.log[idx - 1];
    if (e.len == 0) continue;
    g.q.push_back(n->pool, Item{e.data, e.len, RAFTQ_NODE_ENTRY});
    n->stats.entries_published++;
  }
  if (upto > g.applied) g.applied = upto;
}

// the lines a publication of g's next entries will touch (g's own line is expected in cache): the log from `applied` on,
// the commit channel's tail
inline void ahead_of_publish(const Group& g) {
  if (g.applied < g.log.size()) __builtin_prefetch(&g.log[g.applied]);
  if (g.q.p) __builtin_prefetch(g.q.p + g.q.n, 1);
}

void note_commit(raftq_node_t* n, Group& g, uint64_t commit) {
  if (commit > g.committed) {
    g.committed = commit;
    publish(n, g, commit);
    if (n->wal_on) wal_touch(n, (uint64_t)(&g - n->groups.data()), g);  // HardState.Commit moved
  }
}

// raft.sendAppend(to): entries from Progress.Next on; an empty MsgApp still carries the commit
// index.  Optimistic cursor (ProgressStateReplicate): Next jumps past what was sent.
void send_append(raftq_node_t* n, uint64_t gi, Group& g, uint32_t to) {
  const uint64_t last = g.log.size();
  uint64_t& next = n->next_of(gi, to);
  uint64_t nx = std::max<uint64_t>(next, 1);
  if (nx > last + 1) nx = last + 1;
  raftq_wire_msg_t& m = send(n, to, gi, RAFTQ_MSG_APP, g.term);
  m.index = nx - 1;
  m.log_term = term_at(g, nx - 1);
  m.commit = g.committed;
  uint64_t cnt = 0, bytes = 0;
  while (nx + cnt <= last && cnt < kMaxEntriesPerMsg && bytes < kMaxBytesPerMsg) {
    bytes += g.log[nx + cnt - 1].len;
    ++cnt;
  }
  if (cnt) attach(n, to, m, &g.log[nx - 1], cnt);
  next = nx + cnt;
}

void bcast_append(raftq_node_t* n, uint64_t gi, Group& g) {
  for (uint32_t p = 0; p < n->N; ++p)
    if (p != n->self) send_append(n, gi, g, p);
}

// raft.bcastHeartbeat: `commit := min(r.prs[to].Match, r.raftLog.committed)`
void bcast_heartbeat(raftq_node_t* n, uint64_t gi, Group& g) {
  for (uint32_t p = 0; p < n->N; ++p) {
    if (p == n->self) continue;
    raftq_wire_msg_t& m = send(n, p, gi, RAFTQ_MSG_HEARTBEAT, g.term);
    m.commit = std::min(g.leading ? n->match_of(gi, p) : 0, g.committed);
  }
}

// the leader's (or a forwarded) proposal: appendEntry on the leader, forward / drop elsewhere.
// Returns true when the log grew (the caller reports the tail and broadcasts).
// `ents` are views (into the proposal blob or the receive buffer): the leader copies the payloads into the arena.
bool handle_proposal(raftq_node_t* n, uint64_t gi, Group& g, const Entry* ents, size_t n_ents) {
  if (g.role == RAFTQ_ROLE_LEADER) {
    for (size_t i = 0; i < n_ents; ++i) {
      const char* at = n->arena.put(ents[i].data, ents[i].len);
      if (!at) {
        n->oom = true;
        return i != 0;
      }
      g.log.push_back(n->pool, Entry{g.term, at, ents[i].len});
    }
    if (n_ents) wal_touch(n, gi, g);
    return n_ents != 0;
  }
  if (g.lead != 0 && (uint32_t)(g.lead - 1) != n->self) {  // stepFollower MsgProp: `m.To = r.lead; r.send(m)`
    raftq_wire_msg_t& m = send(n, g.lead - 1, gi, RAFTQ_MSG_PROP, 0);
    attach(n, g.lead - 1, m, ents, n_ents);
  } else {
    n->stats.proposals_dropped += n_ents;  // no leader: etcd drops the proposal
  }
  return false;
}

// RAFTQ_OUT_APPENDED: Step found the MsgApp on the log's tail and did raftLog.maybeAppend's bookkeeping; the entries
// themselves go into the log here.  false: they could not be stored (the node poisons itself at the end of the turn).
bool store_at_tail(raftq_node_t* n, uint64_t gi, Group& g, const raftq_wire_msg_t& m) {
  if (m.index != g.log.size()) {  // the engine's tail and the log's have parted: nothing this node says can be trusted
    n->broken = true;
    return false;
  }
  if (m.n_ents == 0) return true;
  const raftq_wire_ent_t* ents = n->cur_ents + m.ent_first;
  for (uint32_t k = 0; k < m.n_ents; ++k) {
    const char* at = n->arena.put(n->cur_bytes + ents[k].data_off, ents[k].data_len);
    if (!at) {
      n->oom = true;
      return false;
    }
    g.log.push_back(n->pool, Entry{ents[k].term, at, ents[k].data_len});
  }
  wal_touch(n, gi, g);
  return true;
}

// handleAppendEntries on the log's owner, after Step accepted the header (RAFTQ_OUT_APPEND)
void follower_append(raftq_node_t* n, uint64_t gi, Group& g, const raftq_wire_msg_t& m) {
  const raftq_wire_ent_t* ents = n->cur_ents + m.ent_first;
  raftq_wire_msg_t& r = send(n, m.from, gi, RAFTQ_MSG_APP_RESP, g.term);
  if (m.index < g.committed) {  // `if m.Index < r.raftLog.committed { send MsgAppResp{Index: committed} }`
    r.index = g.committed;
    return;
  }
  const uint64_t have = g.log.size();
  if (m.index <= have && term_at(g, m.index) == m.log_term) {  // raftLog.maybeAppend
    size_t k = 0;
    for (; k < m.n_ents; ++k) {  // findConflict
      const uint64_t idx = m.index + 1 + k;
      if (idx > g.log.size()) break;
      if (g.log[idx - 1].term != ents[k].term) {
        g.log.truncate(idx - 1);  // a conflicting suffix is never committed (Raft 5.3)
        g.wal_upto = std::min<uint64_t>(g.wal_upto, idx - 1);  // the WAL gets the replacement entries again
        // replayWAL published the whole log, committed or not (raft.go:122-134), so `applied` may sit beyond the
        // truncation point: the replacement entries at those indices must reach the commit channel once they
        // commit (the reference re-publishes them through rd.Entries), so the cursor comes back with the log
        g.applied = std::min<uint64_t>(g.applied, idx - 1);
        n->shared_group = ~0ull;
        break;
      }
    }
    for (; k < m.n_ents; ++k) {
      const char* at = n->arena.put(n->cur_bytes + ents[k].data_off, ents[k].data_len);
      if (!at) {  // out of memory: the node poisons itself at the end of this turn; do not acknowledge what is not stored
        n->oom = true;
        n->out_lane[m.from].msgs.pop_back();
        n->stats.msgs_sent--;
        return;
      }
      g.log.push_back(n->pool, Entry{ents[k].term, at, ents[k].data_len});
    }
    wal_touch(n, gi, g);
    const uint64_t lastnewi = m.index + m.n_ents;
    r.index = lastnewi;
    raftq_log_delta_t d;
    d.group = gi;
    d.last_index = g.log.size();
    d.last_term = term_at(g, g.log.size());
    d.commit_to = std::min(m.commit, lastnewi);  // `commitTo(min(m.Commit, lastnewi))`
    n->deltas.push_back(d);
    n->deltas_need_result = true;  // a follower's commitTo: what the engine makes of it is published
  } else {  // reject with the hint
    r.index = m.index;
    r.reject = 1;
    r.reject_hint = have;
  }
}

// report log tails to the engine, then publish whatever the reports committed.  Called with
// `lk` held; the device call runs unlocked.  On failure returns with `lk` RELEASED (so the caller
// can poison()), on success with it held again.
int flush_deltas(raftq_node_t* n, std::unique_lock<std::mutex>& lk) {
  if (n->deltas.empty()) return RAFTQ_OK;
  if (n->deltas_nowait && n->N > 1 && !n->deltas_need_result) {
    // every report is a leader's appendEntry and there is more than one peer: raftLog.committed cannot move (the leader's own
    // Match is the largest, the quorum-th largest is somebody else's) -- enqueued and left (raftq_apply_log_deltas_nowait);
    // the engine's state has moved by the time the next Step runs
    lk.unlock();
    int rc;
    {
      DevCall dev(n, raftq_node::kPhDevDeltas);
      rc = raftq_apply_log_deltas_nowait(n->h, n->deltas.data(), n->deltas.size());
    }
    if (rc != RAFTQ_OK) return rc;
    lk.lock();
    n->deltas.clear();
    return RAFTQ_OK;
  }
  n->deltas_need_result = false;
  n->delta_commit.resize(n->deltas.size());
  lk.unlock();
  int rc;
  {
    DevCall dev(n, raftq_node::kPhDevDeltas);
    rc = raftq_apply_log_deltas(n->h, n->deltas.data(), n->deltas.size(), n->delta_commit.data());
  }
  if (rc != RAFTQ_OK) return rc;
  lk.lock();
  const size_t nd = n->deltas.size();
  for (size_t i = 0; i < nd; ++i) {
    if (i + 16 < nd) __builtin_prefetch(&n->groups[n->deltas[i + 16].group]);
    if (i + 8 < nd) ahead_of_publish(n->groups[n->deltas[i + 8].group]);
    note_commit(n, n->groups[n->deltas[i].group], n->delta_commit[i]);
  }
  n->deltas.clear();
  return RAFTQ_OK;
}

// what one Step result means for the node (the "Ready" consequences of one message)
// The node reads Step's 32-byte result records (raftq_step_set_compact(h, 2), round 6: what a result says beyond the message it
// answers and beyond what the log's owner knows anyway -- 32 bytes per message less on the link than the full record, which is
// what bounds the inbound half of a turn): group and addressee are the message's own, raftLog.lastIndex() after the message is
// this node's own log's business, a new leader's log_term is its term and a campaign's rides in `commit` (include/raftq_step.h).
inline raftq_step_out_t widen(const raftq_step_out_s_t& c, const raftq_wire_msg_t& im) {
  raftq_step_out_t o;
  o.group = im.group;
  o.term = c.term;
  o.index = c.index;
  const bool campaign = c.type == RAFTQ_OUT_CAMPAIGN;
  o.commit = campaign ? 0 : c.commit;  // (a campaign moves no commit index: note_commit(0) is a no-op)
  o.log_term = campaign ? c.commit : c.type == RAFTQ_OUT_BECAME_LEADER ? c.term : 0;
  o.last_index = 0;  // (nothing below reads it)
  o.to = im.from;
  o.vote = c.vote;
  o.lead = c.lead;
  o.type = c.type;
  o.reject = c.reject;
  o.flags = c.flags;
  o.role = c.role;
  return o;
}

void apply_result(raftq_node_t* n, const raftq_step_out_t& o, const raftq_wire_msg_t& im) {
  const uint64_t gi = im.group;
  Group& g = n->groups[gi];
  g.term = o.term;
  g.lead = (uint16_t)o.lead;
  g.vote = (uint16_t)o.vote;
  g.role = o.role;
  if (o.flags & (RAFTQ_OUTF_HARDSTATE | RAFTQ_OUTF_STEPPED_DOWN)) {
    if (o.flags & RAFTQ_OUTF_HARDSTATE) {
      n->stats.hard_states++;  // wal.Save(rd.HardState, ...) (raft.go:228)
      wal_touch(n, gi, g);
    }
    if (o.flags & RAFTQ_OUTF_STEPPED_DOWN) g.leading = false;
  }
  // Step appended the message's entries at the tail itself (RAFTQ_MSGF_ENTRIES): they go into the log BEFORE the commit
  // index it reports is published
  bool stored = true;
  if (o.type == RAFTQ_OUT_APPENDED) stored = store_at_tail(n, gi, g, im);
  note_commit(n, g, o.commit);
  switch (o.type) {
    case RAFTQ_OUT_APPENDED:
      if (stored) send(n, o.to, gi, RAFTQ_MSG_APP_RESP, o.term).index = o.index;  // MsgAppResp{Index: lastnewi}
      break;
    case RAFTQ_OUT_VOTE_RESP:
      send(n, o.to, gi, RAFTQ_MSG_VOTE_RESP, o.term).reject = o.reject;
      break;
    case RAFTQ_OUT_HEARTBEAT_RESP:
      send(n, o.to, gi, RAFTQ_MSG_HEARTBEAT_RESP, o.term);
      break;
    case RAFTQ_OUT_CAMPAIGN:
      for (uint32_t p = 0; p < n->N; ++p) {
        if (p == n->self) continue;
        raftq_wire_msg_t& r = send(n, p, gi, RAFTQ_MSG_VOTE, o.term);
        r.index = o.index;
        r.log_term = o.log_term;
      }
      break;
    case RAFTQ_OUT_BECAME_LEADER:
      // becomeLeader's appendEntry(pb.Entry{Data: nil}): the engine already counted it
      g.log.truncate(o.index - 1);
      g.wal_upto = std::min<uint64_t>(g.wal_upto, g.log.size());
      n->shared_group = ~0ull;
        g.log.push_back(n->pool, Entry{o.term, "", 0});
      wal_touch(n, gi, g);
      for (uint32_t p = 0; p < n->N; ++p) {  // reset(): Next = lastIndex + 1 (before the empty entry)
        n->next_of(gi, p) = o.index;
        n->match_of(gi, p) = 0;
      }
      g.leading = true;
      n->match_of(gi, n->self) = o.index;
      n->next_of(gi, n->self) = o.index + 1;
      note_commit(n, g, o.commit);
      bcast_append(n, gi, g);
      break;
    case RAFTQ_OUT_PROGRESS: {
      if (!g.leading) break;
      uint64_t& next = n->next_of(gi, o.to);
      uint64_t& match = n->match_of(gi, o.to);
      if (im.type == RAFTQ_MSG_APP_RESP && im.reject) {
        // Progress.maybeDecrTo: a stale rejection is ignored, else back off to the hint
        if (im.index > match) {
          next = std::max<uint64_t>(std::min(im.index, im.reject_hint + 1), match + 1);
          send_append(n, gi, g, o.to);
        }
      } else {
        match = o.index;
        if (next < o.index + 1) next = o.index + 1;
        if (im.type == RAFTQ_MSG_APP_RESP) {
          if (o.flags & RAFTQ_OUTF_COMMITTED) bcast_append(n, gi, g);  // `if r.maybeCommit() { r.bcastAppend() }`
          else if (next <= g.log.size()) send_append(n, gi, g, o.to);
        } else if (o.index < g.log.size()) {  // MsgHeartbeatResp: `if pr.Match < lastIndex { sendAppend }`
          if (next > o.index + 1) next = o.index + 1;  // whatever was in flight is lost: resend
          send_append(n, gi, g, o.to);
        }
      }
      break;
    }
    case RAFTQ_OUT_APPEND:
      follower_append(n, gi, g, im);
      break;
    default:
      break;
  }
}

uint64_t load_len(const uint8_t* p, bool big_endian) {
  uint64_t v;
  std::memcpy(&v, p, 8);
  return big_endian ? __builtin_bswap64(v) : v;
}

// hand out whole frames of q, at most cap bytes
int poll_queue(raftq_node_t* n, PeerQueue& q, bool big_endian, void* buf, uint64_t cap, uint64_t* len) {
  const uint64_t avail = q.bytes.size() - q.head;
  const uint8_t* p = (const uint8_t*)q.bytes.data() + q.head;
  uint64_t pos = 0;
  while (avail - pos >= 8) {
    const uint64_t body = load_len(p + pos, big_endian);
    if (body > avail - pos - 8 || 8 + body > cap - pos) break;
    pos += 8 + body;
  }
  if (pos) std::memcpy(buf, p, pos);
  q.head += pos;
  if (pos) q.ends_ok = false;  // the frame ends no longer start at the queue's first byte
  if (q.head == q.bytes.size()) q.reset();
  *len = pos;
  if (pos == 0 && avail != 0) {
    n->errtext = "poll: buffer smaller than the next frame";
    return RAFTQ_EINVAL;
  }
  return RAFTQ_OK;
}

// The second half of the WAL encode, in two steps so that a turn's WAL bytes are PUBLISHED BEFORE its outbound frames
// (wal.Save before transport.Send, raft.go:228-230; the poll calls may run on a transport thread during advance(): a
// thread that takes turn T's MsgAppResp off raftq_node_poll must find T's HardState / entries in raftq_node_wal_poll).
// wal_end_device: the device wait, with mu released; wal_publish: the bytes onto wal_out, with mu held.
int wal_end_device(raftq_node_t* n) {
  if (n->wal_inflight == 0 || !n->wal_begun) return RAFTQ_OK;
  const int rc = raftq_wal_encode_end(n->h, &n->wal_cnt);
  n->wal_begun = false;
  return rc;
}
void wal_publish(raftq_node_t* n) {
  if (n->wal_inflight == 0) return;
  n->wal_out.bytes.append((const char*)n->wal_enc.p, (size_t)n->wal_cnt.bytes);
  n->wal_crc = n->wal_cnt.last_crc;
  n->wal_head_written = true;
  n->stats.wal_records += n->wal_inflight;
  n->wal_inflight = 0;
}
// a turn with nothing to send: the WAL encode's wait is its own
int flush_wal_end(raftq_node_t* n, std::unique_lock<std::mutex>& lk) {
  if (n->wal_inflight == 0) return RAFTQ_OK;
  if (n->wal_begun) {
    lk.unlock();
    if (const int rc = wal_end_device(n)) return rc;
    lk.lock();
  }
  wal_publish(n);
  return RAFTQ_OK;
}

// rc.transport.Send(rd.Messages) (raft.go:230) for the whole turn: one batched marshal on the GPU, then
// every peer's slice of the stream goes onto its queue.  Lock convention as flush_deltas.
int flush_outbound(raftq_node_t* n, std::unique_lock<std::mutex>& lk) {
  size_t nm = 0;
  uint64_t cap = 0;  // an upper bound of the stream, kept by send() / attach()
  for (const OutLane& lane : n->out_lane) {
    nm += lane.msgs.size();
    cap += lane.cap;
  }
  // the groups whose proposals go through the device (raftq_propose_frames): N - 1 MsgApps each that exist only in HBM, one run per
  // peer behind the frames of what this turn queued on the host
  const size_t n_props = n->prop_recs.count<raftq_prop_t>(), n_pents = n->prop_ents.count<raftq_prop_ent_t>();
  const size_t n_dev = n_props * (n->N - 1);
  cap += n_props ? n->prop_cap * (n->N - 1) : 0;
  if (nm + n_dev == 0) return RAFTQ_OK;
  if (n->out_oom) {
    lk.unlock();
    return RAFTQ_ENOMEM;
  }
  // the lanes back to back: per-peer order is the order of the sends, and every peer's frames are one slice of the stream
  n->enc_msgs.clear();
  if (!n->enc_off.resize(nm + n_dev + 1) || !n->enc_msgs.reserve(std::max<size_t>(nm, 1) * sizeof(raftq_wire_msg_t)) || !n->enc_out.reserve(cap)) {
    lk.unlock();
    return RAFTQ_ENOMEM;
  }
  std::vector<uint64_t>& first = n->lane_first;
  first.assign(n->N + 1, 0);
  for (uint32_t p = 0; p < n->N; ++p) {
    OutLane& lane = n->out_lane[p];
    first[p + 1] = first[p] + lane.msgs.size();
    n->enc_msgs.append(lane.msgs.data(), lane.msgs.size() * sizeof(raftq_wire_msg_t));
    lane.msgs.clear();
    lane.cap = 0;
  }
  const raftq_wire_msg_t* sorted = n->enc_msgs.as<raftq_wire_msg_t>();
  const raftq_wire_ent_t* ents = n->out_ents.as<raftq_wire_ent_t>();
  const size_t n_ents = n->out_ents.count<raftq_wire_ent_t>();
  n->shared_group = ~0ull;
  raftq_wire_counts_t cnt;
  lk.unlock();  // send() only runs inside advance(), which this thread holds (turn_mu): out_* are safe to read unlocked
  int rc;
  {
    DevCall dev(n, raftq_node::kPhDevEncode);
    if (n_props)
      rc = raftq_propose_frames(n->h, n->prop_recs.as<raftq_prop_t>(), n_props, n->prop_ents.as<raftq_prop_ent_t>(), n_pents, nm ? sorted : nullptr, nm,
                                n_ents ? ents : nullptr, n_ents, n->out_pool.p, n->out_pool.size, n->enc_out.p, cap, n->enc_off.data(), &cnt);
    else
      rc = raftq_wire_encode(n->h, sorted, nm, ents, n_ents, n->out_pool.p, n->out_pool.size, n->enc_out.p, cap, n->enc_off.data(), &cnt);
  }
  if (rc != RAFTQ_OK) return rc;
  // the marshal's wait covered the WAL encode enqueued in front of it: take its verdict now, still unlocked, and publish the
  // WAL bytes first, the frames second, under ONE hold of mu (the frames used to be visible a device call earlier)
  {
    DevCall dev(n, raftq_node::kPhDevEncode);
    rc = wal_end_device(n);
  }
  if (rc != RAFTQ_OK) return rc;
  lk.lock();
  wal_publish(n);
  n->out_ents.clear();
  n->out_pool.clear();
  n->prop_recs.clear();
  n->prop_ents.clear();
  const uint64_t* off = n->enc_off.data();
  // peer p's bytes: the slice of what the host queued for it, then (p != self) its run of the device-built MsgApps -- the order
  // of the sends: the proposals are the turn's last
  auto take = [&](PeerQueue& q, uint64_t k0, uint64_t k1) {
    if (k1 == k0) return;
    if (q.bytes.empty()) q.reuse_spares();
    const uint64_t from = off[k0], base = q.bytes.size();
    q.bytes.append((const char*)n->enc_out.p + from, (size_t)(off[k1] - from));
    if (q.ends_ok) {
      const size_t at = q.ends.size();
      q.ends.resize(at + (size_t)(k1 - k0));
      for (uint64_t k = k0; k < k1; ++k) q.ends[at + (k - k0)] = base + (off[k + 1] - from);
    }
  };
  uint32_t run = 0;
  for (uint32_t p = 0; p < n->N; ++p) {
    PeerQueue& q = n->outbound[p];
    take(q, first[p], first[p + 1]);
    if (n_props && p != n->self) {
      take(q, nm + (uint64_t)run * n_props, nm + (uint64_t)(run + 1) * n_props);
      ++run;
    }
  }
  return RAFTQ_OK;
}

// rc.wal.Save(rd.HardState, rd.Entries) (raft.go:228) for every group this turn touched: per group its new
// entries, then its HardState if it changed (wal.Save's order), groups ascending; one batched encode with
// the segment's running CRC carried across turns.  The first call also writes what wal.Create writes at
// the head of a file: the crc record, the (empty) metadata, the empty snapshot.
// In two halves: _begin builds the records and ENQUEUES the encode (raftq_wal_encode_begin), the wait of the outbound marshal that
// follows covers it, _end takes the bytes -- the turn's two encodes are one submission.
int flush_wal_begin(raftq_node_t* n, std::unique_lock<std::mutex>& lk) {
  n->wal_inflight = 0;
  if (!n->wal_on || (n->wal_n_dirty == 0 && n->wal_head_written)) return RAFTQ_OK;
  PinBuf& recs = n->wal_recs;
  PinBuf& pool = n->wal_pool;
  recs.clear();
  pool.clear();
  bool oom = false;
  // records are built in place in the page-locked array (room is made a group at a time: no staging copy, no per-record
  // capacity check -- 65K records a turn of the one-node leg)
  size_t n_put = 0;
  raftq_wal_rec_t* rbase = recs.as<raftq_wal_rec_t>();
  auto room = [&](size_t k) {  // room for k more records
    if ((n_put + k) * sizeof(raftq_wal_rec_t) > recs.cap) {
      recs.size = n_put * sizeof(raftq_wal_rec_t);
      if (!recs.reserve(std::max<size_t>((n_put + k) * 2, 4096) * sizeof(raftq_wal_rec_t))) oom = true;
      rbase = recs.as<raftq_wal_rec_t>();
    }
    return !oom;
  };
  auto put = [&](uint8_t kind, uint64_t group, uint64_t term, uint64_t index, uint32_t vote, const Entry* data) {
    if (oom) return;
    raftq_wal_rec_t& r = rbase[n_put++];
    r.group = group;
    r.term = term;
    r.index = index;
    r.data_off = 0;
    r.data_len = 0;
    r.vote = vote;
    r.crc = 0;
    r.kind = kind;
    r.entry_type = 0;
    r.flags = 0;
    r._pad = 0;
    if (data && data->len) {
      r.data_len = data->len;
      r.data_off = pool.size;
      oom |= !pool.append(data->data, data->len);
    }
  };
  (void)room(3 + n->wal_n_dirty * 2);
  if (!n->wal_head_written) {
    put(RAFTQ_WAL_CRC, 0, 0, 0, 0, nullptr);
    put(RAFTQ_WAL_METADATA, 0, 0, 0, 0, nullptr);
    put(RAFTQ_WAL_SNAPSHOT, 0, 0, 0, 0, nullptr);
  }
  for (uint64_t w = n->wal_lo; n->wal_n_dirty != 0 && w <= n->wal_hi; ++w)
   for (uint64_t bits = n->wal_bits[w]; bits != 0; bits &= bits - 1) {  // groups ascending (wal.Save's order within a group: entries, then HardState)
    const uint64_t gi = w * 64 + (uint64_t)__builtin_ctzll(bits);
    Group& g = n->groups[gi];
    g.wal_dirty = false;
    if (!room(g.log.size() - std::min<uint64_t>(g.wal_upto, g.log.size()) + 1)) break;
    for (uint64_t idx = g.wal_upto + 1; idx <= g.log.size(); ++idx) {
      const Entry& e = g.log[idx - 1];
      put(RAFTQ_WAL_ENTRY, gi, e.term, idx, 0, &e);
    }
    g.wal_upto = g.log.size();
    const bool empty_hs = g.term == 0 && g.vote == 0 && g.committed == 0;  // `if !raft.IsEmptyHardState(st)`
    if (!empty_hs && (g.term != g.wal_term || g.vote != g.wal_vote || g.committed != g.wal_commit)) {
      // raft IDs are 1-based peer positions (raft.go:148-151), 0 = None: Vote goes out as it is
      put(RAFTQ_WAL_STATE, gi, g.wal_term = g.term, g.wal_commit = g.committed, g.wal_vote = g.vote, nullptr);
    }
  }
  for (uint64_t w = n->wal_lo; n->wal_n_dirty != 0 && w <= n->wal_hi; ++w) n->wal_bits[w] = 0;
  n->wal_lo = ~0ull;
  n->wal_hi = 0;
  n->wal_n_dirty = 0;
  recs.size = n_put * sizeof(raftq_wal_rec_t);
  const size_t n_recs = n_put;
  if (n_recs == 0) return RAFTQ_OK;
  const uint64_t cap = (uint64_t)n_recs * 80 + pool.size;
  if (oom || !n->wal_enc.reserve(cap)) {
    lk.unlock();
    return RAFTQ_ENOMEM;
  }
  const uint32_t prev = n->wal_crc;
  lk.unlock();
  int rc = n->split_wal ? raftq_wal_encode_begin(n->h, recs.as<raftq_wal_rec_t>(), n_recs, pool.p, pool.size, prev, n->wal_enc.p, cap, nullptr) : RAFTQ_EINVAL;
  n->wal_begun = rc == RAFTQ_OK;
  if (rc == RAFTQ_EINVAL) {  // buffers the device cannot address (or RAFTQ_NODE_SPLIT_WAL=0): the whole call, now
    rc = raftq_wal_encode(n->h, recs.as<raftq_wal_rec_t>(), n_recs, pool.p, pool.size, prev, n->wal_enc.p, cap, nullptr, &n->wal_cnt);
  }
  if (rc != RAFTQ_OK) return rc;
  lk.lock();
  n->wal_inflight = n_recs;
  return RAFTQ_OK;
}
// one entry into a group's log under the WAL's rule (wal.ReadAll: a later entry with an index already
// seen replaces it and everything after it)
bool log_put(raftq_node_t* n, Group& g, uint64_t index, uint64_t term, const char* data, uint32_t len) {
  if (index == 0 || index > g.log.size() + 1) return false;
  if (index <= g.log.size()) g.log.truncate(index - 1);
  const char* at = n->arena.put(data, len);
  if (!at) return false;
  g.log.push_back(n->pool, Entry{term, at, len});
  return true;
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
  if (rc == RAFTQ_OK) rc = raftq_step_set_msg_flags(n->h, 1);  // this node fills every byte of every record it stages (RAFTQ_MSGF_*)
  if (rc == RAFTQ_OK) rc = raftq_step_set_compact(n->h, 2);    // ... and reads 32-byte result records (widen())
  if (rc != RAFTQ_OK) {
    if (n->h) raftq_destroy(n->h);
    delete n;
    return rc;  // text is in raftq_last_error(NULL)
  }
  n->G = n_groups;
  n->N = n_peers;
  n->self = self_peer;
  n->profiling = std::getenv("RAFTQ_PROFILE") != nullptr;
  if (const char* ta = std::getenv("RAFTQ_NODE_TAIL_APPENDS")) n->tail_appends = std::atoi(ta) != 0;
  if (const char* fi = std::getenv("RAFTQ_NODE_FUSE_INBOUND")) n->fuse_inbound = std::atoi(fi) != 0;
  if (const char* dn = std::getenv("RAFTQ_NODE_DELTAS_NOWAIT")) n->deltas_nowait = std::atoi(dn) != 0;
  if (const char* sw = std::getenv("RAFTQ_NODE_SPLIT_WAL")) n->split_wal = std::atoi(sw) != 0;
  if (const char* pd = std::getenv("RAFTQ_NODE_PROPOSE_DEVICE")) n->propose_device = std::atoi(pd) != 0;
  if (n_peers < 2) n->propose_device = false;  // (a single-peer group commits what it appends: the report has to come back)
  if (const char* ev = std::getenv("RAFTQ_PROFILE_EVERY")) n->prof_every = std::strtoull(ev, nullptr, 10);
  try {
    n->groups.resize(n_groups);
    n->outbound.resize(n_peers);
    n->out_lane.resize(n_peers);
    n->tick_list.resize(std::min<uint64_t>(n_groups, 4096));
    n->blocked_mark.assign(n_groups, 0);
    n->defer_mark.assign(n_groups, 0);
    n->dirty_mark.assign(n_groups, 0);
    n->prop_mark.assign(n_groups, 0);
    n->prop_slot.assign(n_groups, 0);
    n->wal_bits.assign((n_groups + 63) / 64, 0);
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
    const char* at = n->arena.put(data[i], lens[i]);
    if (!at) {
      n->errtext = "replay: host allocation failed";
      return RAFTQ_ENOMEM;
    }
    g.log.push_back(n->pool, Entry{terms[i], at, lens[i]});
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
      n->prog.assign((size_t)n->N * n->G * 2, 0);
    } catch (...) {
      n->errtext = "start: host allocation failed";
      return RAFTQ_ENOMEM;
    }
    for (uint64_t gi = 0; gi < n->G; ++gi) {
      Group& g = n->groups[gi];
      // replayWAL (raft.go:122-134): every logged entry goes out, then the nil sentinel
      publish(n, g, g.log.size());
        g.q.push_back(n->pool, Item{"", 0, RAFTQ_NODE_SENTINEL});
      g.term = term[gi] = g.hs_term;
      vote[gi] = g.hs_vote;
      g.vote = (uint16_t)g.hs_vote;
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
  try {
    n->proposals.reserve_more(1, len);
  } catch (...) {
    n->errtext = "propose: host allocation failed";
    return RAFTQ_ENOMEM;
  }
  n->proposals.add(group, data, len);
  return RAFTQ_OK;
}

int raftq_node_propose_batch(raftq_node_t* n, const uint64_t* groups, const uint64_t* offsets, const void* blob, uint64_t k) {
  if (!n) return RAFTQ_EINVAL;
  if (k == 0) return RAFTQ_OK;
  if (!groups || !offsets || (!blob && offsets[k] != offsets[0])) return nfail(n, RAFTQ_EINVAL, "propose_batch: null argument");
  for (uint64_t i = 0; i < k; ++i)
    if (groups[i] >= n->G || offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xffffffffull)
      return nfail(n, RAFTQ_EINVAL, "propose_batch: group out of range or offsets not ascending; nothing was queued");
  std::lock_guard<std::mutex> lk(n->mu);
  if (!n->started || n->closed) {
    n->errtext = n->closed ? "propose: node is closed" : "propose: node not started";
    return RAFTQ_ESTATE;
  }
  try {
    n->proposals.reserve_more((size_t)k, (size_t)(offsets[k] - offsets[0]));
  } catch (...) {
    n->errtext = "propose_batch: host allocation failed; nothing was queued";
    return RAFTQ_ENOMEM;
  }
  for (uint64_t i = 0; i < k; ++i) n->proposals.add(groups[i], (const char*)blob + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
  return RAFTQ_OK;
}

int raftq_node_campaign(raftq_node_t* n, const uint64_t* groups, uint64_t k) {
  if (!n) return RAFTQ_EINVAL;
  if (k == 0) return RAFTQ_OK;
  if (!groups) return nfail(n, RAFTQ_EINVAL, "campaign: null argument");
  for (uint64_t i = 0; i < k; ++i)
    if (groups[i] >= n->G) return nfail(n, RAFTQ_EINVAL, "campaign: group out of range; nothing was queued");
  std::lock_guard<std::mutex> lk(n->mu);
  if (!n->started || n->closed) {
    n->errtext = "campaign: node not running";
    return RAFTQ_ESTATE;
  }
  try {
    n->pending_hups.insert(n->pending_hups.end(), groups, groups + k);
  } catch (...) {
    n->errtext = "campaign: host allocation failed";
    return RAFTQ_ENOMEM;
  }
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

// frames -> the node's inbound buffer.  `ends` (may be null): where each of the n_ends frames ends, as the sending node's
// encoder reported them (raftq_node_forward); without it the length words are walked here.
static int deliver_impl(raftq_node_t* n, const void* frames, uint64_t len, const uint64_t* ends, size_t n_ends) {
  if (!n) return RAFTQ_EINVAL;
  if (len && !frames) return nfail(n, RAFTQ_EINVAL, "deliver: null buffer");
  if (len == 0) return RAFTQ_OK;
  // the stream reader's part of messageDecoder.decode: the length words must tile the buffer.  The
  // messages themselves are unmarshalled on the GPU, all of this turn's at once, by the next advance().
  const uint8_t* p = (const uint8_t*)frames;
  uint64_t nf = n_ends;
  if (!ends) {
    uint64_t pos = 0;
    nf = 0;
    while (len - pos >= 8) {
      const uint64_t body = load_len(p + pos, true);
      if (body > len - pos - 8) break;
      pos += 8 + body;
      ++nf;
    }
    if (pos != len) return nfail(n, RAFTQ_EINVAL, "deliver: not a whole number of stream frames (truncated or garbage)");
  }
  std::lock_guard<std::mutex> lk(n->mu);
  if (!n->started || n->closed) {
    n->errtext = "deliver: node not running";
    return RAFTQ_ESTATE;
  }
  const uint64_t base = n->in_bytes.size;
  {
    const bool was_empty = n->in_off.empty();
    const size_t at = was_empty ? 1 : n->in_off.size();
    // the offsets before the bytes: a failure leaves the buffer as it was
    if (!n->in_off.resize(at + (size_t)nf) || !n->in_bytes.append(frames, (size_t)len)) {
      n->in_off.resize(was_empty ? 0 : at);
      n->errtext = "deliver: page-locked allocation failed";
      return RAFTQ_ENOMEM;
    }
    if (at == 1) n->in_off.data()[0] = 0;
    uint64_t* off = n->in_off.data() + at;
    if (ends) {
      for (uint64_t i = 0; i < nf; ++i) off[i] = base + ends[i];
    } else {
      uint64_t pos = 0;
      for (uint64_t i = 0; i < nf; ++i) {
        pos += 8 + load_len(p + pos, true);
        off[i] = base + pos;
      }
    }
  }
  return RAFTQ_OK;
}

int raftq_node_deliver(raftq_node_t* n, const void* frames, uint64_t len) { return deliver_impl(n, frames, len, nullptr, 0); }

namespace {
void dump_profile(raftq_node_t* n) {
  static const char* names[raftq_node::kPhN] = {"decode", "inbound", "tick", "stage", "step", "apply", "deltas", "props", "wal", "encode",
                                                "dev:deltas", "dev:encode"};
  std::fprintf(stderr, "[raftq_node %u] advance phases, total ms over %llu turns (%llu msgs stepped so far):", n->self,
               (unsigned long long)n->prof_turns, (unsigned long long)n->stats.msgs_stepped);
  for (int i = 0; i < raftq_node::kPhN; ++i) std::fprintf(stderr, " %s %.1f", names[i], n->prof[i] / 1e3);
  std::fprintf(stderr, "\n");
  std::fill(n->prof, n->prof + raftq_node::kPhN, 0.0);
  n->prof_turns = 0;
}
}  // namespace

static int node_advance_impl(raftq_node_t* n, uint64_t* n_published) {
  if (!n) return RAFTQ_EINVAL;
  std::lock_guard<std::mutex> turn(n->turn_mu);
  PhaseClock ph(n, raftq_node::kPhDecode);
  if (n->profiling) n->prof_turns++;
  std::vector<uint32_t>& work = n->work;
  work.clear();
  n->local.clear();
  PropBuf& props = n->turn_props;
  props.clear();
  n->cur_ents = nullptr;
  n->cur_bytes = nullptr;
  PinBuf& in_bytes = n->turn_bytes;  // the half of the inbound double buffer this turn decodes
  PinU64& in_off = n->turn_off;
  in_bytes.clear();
  in_off.clear();
  uint32_t ticks = 0;
  {
    std::lock_guard<std::mutex> lk(n->mu);
    if (!n->started) return RAFTQ_ESTATE;
    if (n->error) return n->error;
    in_bytes.swap(n->in_bytes);
    in_off.swap(n->in_off);
    std::swap(props, n->proposals);
    ticks = n->pending_ticks;
    n->pending_ticks = 0;
    n->turn_hups.clear();
    n->turn_hups.swap(n->pending_hups);
  }
  ph.next(raftq_node::kPhTick);
  // The commit channels, the status mirror and the outbound queues are only written below, under
  // mu, one short critical section per phase.
  std::unique_lock<std::mutex> lk(n->mu);
  const uint64_t published0 = n->stats.entries_published;
  bool did = in_off.size() > 1 || props.size() != 0 || ticks != 0 || !n->turn_hups.empty();

  // -- rc.node.Tick() (raft.go:223-224) for every group: the engine advances the clocks and says
  // which groups' election timers fired (MsgHup -> through Step) and which leaders owe a heartbeat
  std::vector<raftq_wire_msg_t>& local = n->local;
  for (uint64_t gi : n->turn_hups)  // rc.node.Campaign: MsgHup through Step, ahead of what the timers raise
    local.push_back(local_msg(n, gi, RAFTQ_MSG_HUP));
  for (uint32_t t = 0; t < ticks; ++t) {
    lk.unlock();
    // the device compacts the two short lists (ascending group ids); no G-byte read-back and no loop over every
    // group under the lock (O(G) host work per 100 ms tick at 1M groups)
    // one call: the Tick and both of its lists (two launches, one wait).  A list that does not fit is fetched again alone.
    // (raftq_tick_collect_lists: 4-byte ids read where the device left them, the MsgBeat groups as a bitmap -- with
    // HeartbeatTick 1 the beat list is the set of groups this node leads, every tick)
    uint64_t n_hup = 0, n_beat = 0;
    int rc = raftq_tick_collect_lists(n->h, RAFTQ_TICK_BEAT_BITMAP, n->tick_list.size(), 0, &n_hup, &n_beat);
    const uint32_t* hups = nullptr;
    const uint64_t* beat_map = nullptr;
    uint64_t n_listed = 0, map_words = 0;
    if (rc == RAFTQ_OK) rc = raftq_last_tick_lists(n->h, &hups, &n_listed, nullptr, nullptr, &beat_map, &map_words);
    if (rc == RAFTQ_OK && n_hup > n_listed) {  // more timers fired than the list was sized for: fetch them again, alone
      n->tick_list.resize(n_hup);
      rc = raftq_collect_hups(n->h, n->tick_list.data(), n->tick_list.size(), &n_hup);
      if (rc == RAFTQ_OK)
        for (uint64_t i = 0; i < n_hup; ++i) local.push_back(local_msg(n, n->tick_list[i], RAFTQ_MSG_HUP));
    } else if (rc == RAFTQ_OK) {
      for (uint64_t i = 0; i < n_hup; ++i) local.push_back(local_msg(n, hups[i], RAFTQ_MSG_HUP));
    }
    if (rc != RAFTQ_OK) return poison(n, rc, "tick");
    lk.lock();
    if (n_beat) {
      for (uint64_t w = 0; w < map_words; ++w) {
        for (uint64_t bits = beat_map[w]; bits; bits &= bits - 1) {
          const uint64_t gi = w * 64 + (uint64_t)__builtin_ctzll(bits);
          if (n->groups[gi].role == RAFTQ_ROLE_LEADER) bcast_heartbeat(n, gi, n->groups[gi]);  // stepLeader MsgBeat: host only
        }
      }
    }
  }
  if (local.size() >= kLocal) {
    lk.unlock();
    return poison(n, RAFTQ_EINVAL, "more than 2^31 local messages in one turn");
  }
  // -- rafthttp's messageDecoder + Message.Unmarshal for everything received since the last turn, on the GPU.  Frames that
  // do not parse, are not addressed to this node's slot, come from no peer of the cluster or are of a kind a peer never sends
  // are dropped and counted -- rafthttp would log and drop the stream; a raft node must survive any bytes a peer throws at it.
  // When nothing was raised locally ahead of them (no campaign, no timer: every turn of steady-state replication) the same
  // submission also makes those checks and steps every frame, in arrival order (raftq_step_frames: one wait for the inbound
  // half of the turn); otherwise the frames are decoded here (raftq_wire_decode) and stepped in the staged rounds below, behind
  // the local messages.
  uint64_t nf = in_off.size() > 1 ? in_off.size() - 1 : 0;
  const raftq_wire_msg_t* wm = nullptr;
  const raftq_step_out_s_t* fused_outs = nullptr;  // != nullptr: round 1 has been stepped, out[i] answers frame i
  if (nf) {
    lk.unlock();
    ph.next(raftq_node::kPhDecode);
    if (nf >= kLocal) return poison(n, RAFTQ_EINVAL, "more than 2^31 frames in one turn");
    uint64_t ents_cap = std::max<uint64_t>(nf + 1024, n->turn_ents.cap / sizeof(raftq_wire_ent_t));
    if (!n->turn_msgs.reserve(nf * sizeof(raftq_wire_msg_t)) || !n->turn_ents.reserve(ents_cap * sizeof(raftq_wire_ent_t)))
      return poison(n, RAFTQ_ENOMEM, "wire_decode (page-locked result buffers)");
    raftq_wire_msg_t* out = n->turn_msgs.as<raftq_wire_msg_t>();
    raftq_wire_ent_t* we = n->turn_ents.as<raftq_wire_ent_t>();
    raftq_wire_counts_t cnt;
    const bool fuse = n->fuse_inbound && local.empty();
    int rc = fuse ? raftq_step_frames(n->h, in_bytes.p, in_bytes.size, in_off.data(), nf, n->tail_appends ? 1 : 0, out, we, ents_cap, &cnt)
                  : raftq_wire_decode(n->h, in_bytes.p, in_bytes.size, in_off.data(), nf, out, we, ents_cap, &cnt);
    if ((rc == RAFTQ_OK || rc == RAFTQ_EINVAL) && cnt.n_ents > ents_cap) {
      // more entries than messages + 1024: grow once, decode again (after raftq_step_frames only for the headers: the frames
      // have been stepped, and what the plain decode writes over the records is what Step read minus the flags it was given)
      ents_cap = cnt.n_ents;
      if (!n->turn_ents.reserve(ents_cap * sizeof(raftq_wire_ent_t))) return poison(n, RAFTQ_ENOMEM, "wire_decode (entries)");
      we = n->turn_ents.as<raftq_wire_ent_t>();
      rc = raftq_wire_decode(n->h, in_bytes.p, in_bytes.size, in_off.data(), nf, out, we, ents_cap, &cnt);
    }
    if (rc != RAFTQ_OK) return poison(n, rc, fuse ? "step_frames" : "wire_decode");
    if (fuse) {
      uint64_t n_out = 0;
      rc = raftq_step_results_s(n->h, &fused_outs, &n_out);
      if (rc != RAFTQ_OK || n_out != nf || !fused_outs) return poison(n, rc != RAFTQ_OK ? rc : RAFTQ_ESTATE, "step_frames (results)");
    }
    wm = out;
    n->cur_ents = we;
    n->cur_bytes = in_bytes.p;
    lk.lock();
  }
  ph.next(raftq_node::kPhInbound);
  // what this turn works through, in order: the locally raised messages, then the inbound ones as they arrived
  work.resize(local.size() + (size_t)nf);
  for (size_t i = 0; i < local.size(); ++i) work[i] = kLocal | (uint32_t)i;
  for (uint64_t i = 0; i < nf; ++i) work[local.size() + i] = (uint32_t)i;
  auto msg_at = [&](uint32_t ix) -> const raftq_wire_msg_t& { return (ix & kLocal) ? local[ix & ~kLocal] : wm[ix]; };

  // -- rc.Process -> Step, in rounds.  What follows a message that changes a group's log has to see the new log tail: a MsgProp
  // (which never reaches Step) keeps the rest of its group for the next round; a MsgApp only when Step could not finish it
  // itself (RAFTQ_MSGF_BARRIER).  In steady-state replication everything goes in the first round.
  std::vector<uint32_t>& batch = n->batch;
  std::vector<uint32_t>& deferred = n->deferred;
  std::vector<uint64_t>& dirty = n->dirty_list;
  auto next_epoch = [&] {  // a fresh value no mark holds
    if (++n->epoch >= 0x80000000u) {  // (below 2^31: the proposals' marks use the top bit)
      std::fill(n->blocked_mark.begin(), n->blocked_mark.end(), 0u);
      std::fill(n->dirty_mark.begin(), n->dirty_mark.end(), 0u);
      std::fill(n->defer_mark.begin(), n->defer_mark.end(), 0u);
      std::fill(n->prop_mark.begin(), n->prop_mark.end(), 0u);
      n->epoch = 1;
    }
    return n->epoch;
  };
  // the views of one inbound MsgProp's entries (payloads stay in the receive buffer)
  auto prop_entries = [&](const raftq_wire_msg_t& im) -> const Entry* {
    n->ent_tmp.clear();
    for (uint32_t k = 0; k < im.n_ents; ++k) {
      const raftq_wire_ent_t& e = n->cur_ents[im.ent_first + k];
      n->ent_tmp.push_back(Entry{e.term, (const char*)n->cur_bytes + e.data_off, e.data_len});
    }
    return n->ent_tmp.data();
  };
  // report the grown logs' tails, publish what that committed, then bcastAppend
  auto flush_dirty = [&]() -> int {
    const size_t nd = dirty.size();
    for (size_t i = 0; i < nd; ++i) {
      if (i + 16 < nd) __builtin_prefetch(&n->groups[dirty[i + 16]]);
      const uint64_t gi = dirty[i];
      Group& g = n->groups[gi];
      raftq_log_delta_t d{gi, g.log.size(), g.term, 0};
      n->deltas.push_back(d);
    }
    if (int rc = flush_deltas(n, lk)) return rc;
    for (size_t i = 0; i < nd; ++i) {  // bcastAppend: the group's line, its Progress and the entry before Next
      if (i + 16 < nd) __builtin_prefetch(&n->groups[dirty[i + 16]]);
      if (i + 8 < nd) {
        const Group& ahead = n->groups[dirty[i + 8]];
        __builtin_prefetch(&n->prog[dirty[i + 8] * n->N * 2], 1);
        if (!ahead.log.empty()) __builtin_prefetch(&ahead.log.back());
      }
      bcast_append(n, dirty[i], n->groups[dirty[i]]);
    }
    return RAFTQ_OK;
  };
  bool first_round = true;
  uint64_t dropped = 0;
  while (!work.empty()) {
    ph.next(raftq_node::kPhStage);
    batch.clear();
    deferred.clear();
    dirty.clear();
    const uint32_t ep = next_epoch();
    // one pass: what cannot go yet is deferred, the rest is written where the GPU reads it (raftq_step_stage: device
    // memory behind a large BAR) -- the decoder's 64-byte record IS Step's record
    raftq_msg_t* staged = nullptr;
    static_assert(sizeof(raftq_msg_t) == sizeof(raftq_wire_msg_t), "the decoder's record is Step's record");
    size_t n_step = 0;
    const raftq_step_out_s_t* outs = nullptr;
    const bool fused = first_round && fused_outs != nullptr;
    if (fused) {
      // round 1 has been stepped already, every frame of it, by the submission that decoded them (raftq_step_frames): what is
      // nobody's was skipped, a MsgProp held its group, a MsgApp was a barrier -- out[i] answers frame i
      batch.assign(work.begin(), work.end());
      outs = fused_outs;
      first_round = false;
    } else {
    lk.unlock();
    if (int rc = raftq_step_stage(n->h, work.size(), &staged)) return poison(n, rc, "step_stage");
    for (const uint32_t ix : work) {
      const raftq_wire_msg_t& m = msg_at(ix);
      if (first_round && !(ix & kLocal)) {
        const bool kind_ok = m.type == RAFTQ_MSG_PROP || m.type == RAFTQ_MSG_APP || m.type == RAFTQ_MSG_APP_RESP ||
                             m.type == RAFTQ_MSG_VOTE || m.type == RAFTQ_MSG_VOTE_RESP || m.type == RAFTQ_MSG_HEARTBEAT ||
                             m.type == RAFTQ_MSG_HEARTBEAT_RESP;
        if ((m.flags & RAFTQ_WIRE_F_MALFORMED) || !kind_ok || m.group >= n->G || m.from >= n->N || m.to != n->self) {
          ++dropped;
          continue;
        }
      }
      uint32_t& mark = n->blocked_mark[m.group];
      if (mark == ep) {
        deferred.push_back(ix);
        continue;
      }
      if (m.type == RAFTQ_MSG_PROP) {  // MsgProp never reaches Step; everything else does, in arrival order
        mark = ep;
      } else if (m.type == RAFTQ_MSG_APP) {
        // a MsgApp says what it carries (RAFTQ_MSGF_ENTRIES): one that lands on the log's tail -- replication's common
        // case -- is then appended and committed by Step itself, with no raftq_apply_log_deltas round trip behind it; one
        // that Step has to leave to this log (a gap, a conflict) holds back what follows it for the group
        // (RAFTQ_MSGF_BARRIER -> RAFTQ_OUT_DEFERRED): those come round again below, after the log has changed
        raftq_msg_t t;
        std::memcpy(&t, &m, sizeof(t));
        t._pad[0] = 0;
        t._pad[1] = RAFTQ_MSGF_BARRIER | (n->tail_appends ? RAFTQ_MSGF_ENTRIES : 0);
        t._resv = m.n_ents;
        t.reject_hint = m.n_ents ? n->cur_ents[m.ent_first + m.n_ents - 1].term : 0;
        std::memcpy(&staged[n_step++], &t, sizeof(t));
      } else {
        std::memcpy(&staged[n_step++], &m, sizeof(raftq_msg_t));
      }
      batch.push_back(ix);
    }
    first_round = false;
    if (n_step) {
      ph.next(raftq_node::kPhStep);
      int rc = raftq_step_batch(n->h, staged, n_step, nullptr, nullptr);
      uint64_t n_out = 0;
      if (rc == RAFTQ_OK) rc = raftq_step_results_s(n->h, &outs, &n_out);
      if (rc != RAFTQ_OK) return poison(n, rc, "step_batch");
    }
    lk.lock();
    }
    ph.next(raftq_node::kPhApply);
    n->stats.msgs_stepped += fused ? batch.size() : n_step;
    const size_t kept_back = deferred.size();  // (behind a MsgProp: decided while staging; what Step defers is added below)
    // consequences, in arrival order (stepped results and proposals interleaved as they came)
    // The groups of a batch are scattered over tens of MB of per-group state: the line of the group 16 messages ahead
    // and, 8 ahead (its line has arrived by then), its Progress and the tail of its log are asked for now.
    const size_t nb = batch.size();
    auto group_of = [&](size_t at) { return msg_at(batch[at]).group; };
    // (a skipped frame's group field may hold anything: in a fused round result `at` answers frame `at`)
    auto live_at = [&](size_t at) { return !fused || outs[at].type != RAFTQ_OUT_SKIPPED; };
    size_t k = 0;
    for (size_t bi = 0; bi < nb; ++bi) {
      if (bi + 16 < nb && live_at(bi + 16)) __builtin_prefetch(&n->groups[group_of(bi + 16)]);
      if (bi + 8 < nb && live_at(bi + 8)) {
        const uint64_t g8 = group_of(bi + 8);
        const Group& ahead = n->groups[g8];
        if (ahead.leading) __builtin_prefetch(&n->prog[g8 * n->N * 2], 1);
        if (!ahead.log.empty()) __builtin_prefetch(&ahead.log.back(), 1);  // term_at(prev), and where the next entry goes
        ahead_of_publish(ahead);
      }
      const raftq_wire_msg_t& im = msg_at(batch[bi]);
      if (fused && outs[k].type == RAFTQ_OUT_SKIPPED) {  // not a message for this node (the checks are the decoder's there)
        ++dropped;
        n->stats.msgs_stepped--;
        ++k;
        continue;
      }
      if (im.type == RAFTQ_MSG_PROP) {
        if (fused) {  // (answered RAFTQ_OUT_HELD: a result slot of its own)
          n->stats.msgs_stepped--;
          ++k;
        }
        if (n->defer_mark[im.group] == ep) {
          // something of this group that arrived earlier waits for the next round (Step deferred it behind a hold or a
          // barrier): the proposal waits behind it -- a group's messages are worked off in arrival order whichever way the
          // round was stepped ([MsgProp, MsgAppResp, MsgProp] used to run the second proposal first)
          deferred.push_back(batch[bi]);
          continue;
        }
        Group& g = n->groups[im.group];
        if (handle_proposal(n, im.group, g, prop_entries(im), im.n_ents) && n->dirty_mark[im.group] != ep) {
          n->dirty_mark[im.group] = ep;
          dirty.push_back(im.group);
        }
      } else if (outs[k].type == RAFTQ_OUT_DEFERRED) {  // behind a MsgApp this log has to work out first: next round
        deferred.push_back(batch[bi]);
        n->defer_mark[im.group] = ep;
        n->stats.msgs_stepped--;
        ++k;
      } else {
        apply_result(n, widen(outs[k++], im), im);
      }
    }
    if (deferred.size() != kept_back && kept_back != 0) std::sort(deferred.begin(), deferred.end());  // arrival order
    ph.next(raftq_node::kPhDeltas);
    if (int rc = flush_dirty()) return poison(n, rc, "apply_log_deltas");
    work.swap(deferred);
  }
  n->stats.frames_dropped += dropped;
  ph.next(raftq_node::kPhProps);

  // -- proposeC (raft.go:211-215)
  n->prop_recs.clear();
  n->prop_ents.clear();
  n->prop_cap = 0;
  if (props.size()) {
    dirty.clear();
    const size_t np_ = props.size();
    // Are every group's statements of this turn NEXT TO EACH OTHER in the queue (the common shapes: one statement per group, or
    // a client's run of statements for one group)?  One look at the queue's group ids -- no group state is touched -- says so; then
    // a group is dealt with ONCE, while its lines are in cache: classified, its statements stored, its Progress moved.
    bool runs = true;
    {
      const uint32_t seen = next_epoch();
      std::vector<uint32_t>& mk = n->prop_mark;
      for (size_t i = 0; i < np_; ++i) {
        const uint64_t gi = props.group[i];
        if (i && gi == props.group[i - 1]) continue;
        if (mk[gi] == seen) {
          runs = false;
          break;
        }
        mk[gi] = seen;
      }
    }
    const uint32_t ep = next_epoch();
    if (runs) {
      const bool dev = n->propose_device;
      bool pin_oom = dev && (!n->prop_recs.reserve(np_ * sizeof(raftq_prop_t)) || !n->prop_ents.reserve(np_ * sizeof(raftq_prop_ent_t)));
      raftq_prop_t* recs = n->prop_recs.as<raftq_prop_t>();
      raftq_prop_ent_t* pents = n->prop_ents.as<raftq_prop_ent_t>();
      size_t n_fast = 0, n_pents = 0;
      for (size_t i = 0; i < np_ && !n->oom;) {
        // (the queue is in arrival order; a run's first statement is a good enough place to look ahead from)
        if (i + 16 < np_) __builtin_prefetch(&n->groups[props.group[i + 16]]);
        if (i + 8 < np_) {
          const uint64_t g8 = props.group[i + 8];
          const Group& ahead = n->groups[g8];
          __builtin_prefetch(&n->prog[g8 * n->N * 2], 1);
          if (ahead.log.p) __builtin_prefetch(ahead.log.p + ahead.log.n, 1);  // appendEntry writes behind the log's last entry
        }
        const uint64_t gi = props.group[i];
        size_t j = i + 1;
        while (j < np_ && props.group[j] == gi) ++j;
        const uint64_t bytes = props.off[j] - props.off[i];
        Group& g = n->groups[gi];
        // through the device: a group this node leads, every follower's Progress.Next at the log's tail (what bcastAppend leaves
        // behind: every group outside a catch-up), one message's worth of statements (raft.Config.MaxSizePerMsg, raft.go:157)
        bool fast = dev && !pin_oom && g.role == RAFTQ_ROLE_LEADER && g.leading && j - i <= kMaxEntriesPerMsg && bytes < kMaxBytesPerMsg;
        const uint64_t tail = g.log.size() + 1;
        for (uint32_t p = 0; fast && p < n->N; ++p)
          if (p != n->self) fast = n->next_of(gi, p) == tail;
        if (fast) {
          recs[n_fast++] = raftq_prop_t{gi, (uint32_t)n_pents, (uint32_t)(j - i)};
          for (size_t k = i; k < j; ++k) {
            const uint32_t len = (uint32_t)(props.off[k + 1] - props.off[k]);
            const char* src = props.blob.data() + props.off[k];
            const char* at = n->arena.put(src, len);
            if (!at) {
              n->oom = true;
              break;
            }
            g.log.push_back(n->pool, Entry{g.term, at, len});
            raftq_prop_ent_t& pe = pents[n_pents++];
            pe.data_off = len ? n->out_pool.size : 0;
            pe.data_len = len;
            pe.type = 0;
            if (!n->out_pool.append(src, len)) n->out_oom = true;
            n->prop_cap += 48 + len;
          }
          const uint64_t next = g.log.size() + 1;  // bcastAppend's optimistic cursor
          for (uint32_t p = 0; p < n->N; ++p)
            if (p != n->self) n->next_of(gi, p) = next;
          wal_touch(n, gi, g);
        } else {
          bool grew = false;
          for (size_t k = i; k < j; ++k) {
            const Entry one{0, props.blob.data() + props.off[k], (uint32_t)(props.off[k + 1] - props.off[k])};
            grew |= handle_proposal(n, gi, g, &one, 1);
          }
          if (grew && n->dirty_mark[gi] != ep) {
            n->dirty_mark[gi] = ep;
            dirty.push_back(gi);
          }
        }
        i = j;
      }
      if (pin_oom) n->out_oom = true;
      n->stats.msgs_sent += n_fast * (n->N - 1);
      n->stats.msgs_built_on_device += n_fast * (n->N - 1);
      n->prop_cap += n_fast * 160;  // (per addressee: flush_outbound multiplies)
      n->prop_recs.size = n_fast * sizeof(raftq_prop_t);
      n->prop_ents.size = n_pents * sizeof(raftq_prop_ent_t);
      if (n->oom) {  // (a device record without its log entry must not go out)
        n->prop_recs.clear();
        n->prop_ents.clear();
      }
      if (int rc = flush_dirty()) return poison(n, rc, "apply_log_deltas");
    } else {
    // The general shape (a group's statements interleaved with other groups'), two passes.
    // Pass 1 (round 6): which groups go through the device.  A group this node leads whose followers all have Progress.Next at
    // the log's tail (what bcastAppend leaves behind: every group outside a catch-up) gets ONE raftq_prop_t for the turn, in the
    // order of its first proposal -- the order its MsgApps take in every peer's stream -- and a count; everybody else takes
    // round 5's way below (handle_proposal: a follower forwards, a leader with a lagging follower sends from its own log).
    // prop_mark[g] == ep: through the device, prop_slot[g] = its record; == ep with the top bit set aside: decided, host.
    std::vector<uint32_t>& pmark = n->prop_mark;
    constexpr uint32_t kSlowBit = 0x80000000u;
    const uint32_t ep_fast = ep, ep_slow = ep | kSlowBit;  // (epochs stay below 2^31: next_epoch)
    size_t n_fast = 0;
    if (n->propose_device) {
      bool pin_oom = !n->prop_recs.reserve(np_ * sizeof(raftq_prop_t)) || !n->prop_ents.reserve(np_ * sizeof(raftq_prop_ent_t));
      raftq_prop_t* recs = n->prop_recs.as<raftq_prop_t>();
      for (size_t i = 0; i < np_ && !pin_oom; ++i) {
        if (i + 16 < np_) __builtin_prefetch(&n->groups[props.group[i + 16]]);
        if (i + 8 < np_) __builtin_prefetch(&n->prog[props.group[i + 8] * n->N * 2]);
        const uint64_t gi = props.group[i];
        uint32_t& mk = pmark[gi];
        if (mk == ep_slow) continue;
        const uint64_t len = props.off[i + 1] - props.off[i];
        if (mk == ep_fast) {  // another statement for a group already on its way: one MsgApp carries them all, within raft.Config.MaxSizePerMsg
          raftq_prop_t& r = recs[n->prop_slot[gi]];
          r.n_ents++;
          r.ent_first = (uint32_t)std::min<uint64_t>((uint64_t)r.ent_first + len, 0xffffffffull);  // (the payload bytes so far; the entries' place comes below)
          continue;
        }
        const Group& g = n->groups[gi];
        bool fast = g.role == RAFTQ_ROLE_LEADER && g.leading;
        const uint64_t tail = g.log.size() + 1;
        for (uint32_t p = 0; fast && p < n->N; ++p)
          if (p != n->self) fast = n->next_of(gi, p) == tail;
        if (!fast) {
          mk = ep_slow;
          continue;
        }
        mk = ep_fast;
        n->prop_slot[gi] = (uint32_t)n_fast;
        recs[n_fast++] = raftq_prop_t{gi, (uint32_t)std::min<uint64_t>(len, 0xffffffffull), 1};
      }
      // a group whose statements outgrow one message (raft.Config.MaxSizePerMsg, raft.go:157: sendAppend would cut it) is the host's
      uint32_t ent_at = 0;
      size_t kept = 0;
      for (size_t k = 0; k < n_fast; ++k) {
        raftq_prop_t r = recs[k];
        if (r.n_ents > kMaxEntriesPerMsg || r.ent_first >= kMaxBytesPerMsg) {
          pmark[r.group] = ep_slow;
          continue;
        }
        r.ent_first = ent_at;
        ent_at += r.n_ents;
        r.n_ents = 0;  // (filled again, entry by entry, by the second pass)
        n->prop_slot[r.group] = (uint32_t)kept;
        recs[kept++] = r;
      }
      n_fast = kept;
      if (pin_oom) n->out_oom = true;
    }
    raftq_prop_t* recs = n->prop_recs.as<raftq_prop_t>();
    raftq_prop_ent_t* pents = n->prop_ents.as<raftq_prop_ent_t>();
    size_t n_pents = 0;
    // Pass 2: every statement, in arrival order.  Through the device: the payload once into the log's arena and once into the
    // turn's page-locked pool (where the encoder's readers find it), the log's entry, 16 bytes that say where the payload is.
    for (size_t i = 0; i < np_; ++i) {
      if (i + 16 < np_) __builtin_prefetch(&n->groups[props.group[i + 16]]);
      if (i + 8 < np_) {  // appendEntry writes behind the log's last entry
        const Group& ahead = n->groups[props.group[i + 8]];
        if (ahead.log.p) __builtin_prefetch(ahead.log.p + ahead.log.n, 1);
      }
      const uint64_t gi = props.group[i];
      const Entry one{0, props.blob.data() + props.off[i], (uint32_t)(props.off[i + 1] - props.off[i])};
      if (n_fast != 0 && pmark[gi] == ep_fast) {
        Group& g = n->groups[gi];
        const char* at = n->arena.put(one.data, one.len);
        if (!at) {
          n->oom = true;
          break;
        }
        g.log.push_back(n->pool, Entry{g.term, at, one.len});
        raftq_prop_t& r = recs[n->prop_slot[gi]];
        raftq_prop_ent_t& pe = pents[r.ent_first + r.n_ents++];
        pe.data_off = one.len ? n->out_pool.size : 0;
        pe.data_len = one.len;
        pe.type = 0;
        ++n_pents;
        if (!n->out_pool.append(one.data, one.len)) n->out_oom = true;
        n->prop_cap += 48 + one.len;
        continue;
      }
      if (handle_proposal(n, gi, n->groups[gi], &one, 1) && n->dirty_mark[gi] != ep) {
        n->dirty_mark[gi] = ep;
        dirty.push_back(gi);
      }
    }
    // what bcastAppend leaves on the host for the groups that went through the device: Progress.Next past what is being sent
    // (the optimistic cursor), the WAL's dirty mark, the count of sends
    for (size_t k = 0; k < n_fast; ++k) {
      if (k + 8 < n_fast) __builtin_prefetch(&n->prog[recs[k + 8].group * n->N * 2], 1);
      const uint64_t gi = recs[k].group;
      Group& g = n->groups[gi];
      const uint64_t next = g.log.size() + 1;
      for (uint32_t p = 0; p < n->N; ++p)
        if (p != n->self) n->next_of(gi, p) = next;
      wal_touch(n, gi, g);
    }
    n->stats.msgs_sent += n_fast * (n->N - 1);
    n->stats.msgs_built_on_device += n_fast * (n->N - 1);
    n->prop_cap += n_fast * 160;  // (per addressee: flush_outbound multiplies)
    n->prop_recs.size = n_fast * sizeof(raftq_prop_t);
    n->prop_ents.size = n_pents * sizeof(raftq_prop_ent_t);
    if (n->oom) {  // (a device record without its log entry must not go out)
      n->prop_recs.clear();
      n->prop_ents.clear();
    }
    if (int rc = flush_dirty()) return poison(n, rc, "apply_log_deltas");
    }
  }
  if (n->oom) {
    lk.unlock();
    return poison(n, RAFTQ_ENOMEM, "entry storage");
  }
  if (n->broken) {
    lk.unlock();
    return poison(n, RAFTQ_ESTATE, "the engine's log tail and the node's log disagree");
  }
  // -- wal.Save before transport.Send (raft.go:228-230): the caller persists what raftq_node_wal_poll
  // hands out before it transmits what raftq_node_poll hands out
  ph.next(raftq_node::kPhWal);
  if (int rc = flush_wal_begin(n, lk)) return poison(n, rc, "wal_encode");
  ph.next(raftq_node::kPhEncode);
  if (int rc = flush_outbound(n, lk)) return poison(n, rc, "wire_encode");
  ph.next(raftq_node::kPhWal);
  if (int rc = flush_wal_end(n, lk)) return poison(n, rc, "wal_encode");
  if (did) n->stats.turns++;
  const uint64_t pub = n->stats.entries_published - published0;
  lk.unlock();
  if (pub) n->cv_commit.notify_all();
  if (n_published) *n_published = pub;
  if (n->profiling && n->prof_every && n->prof_turns >= n->prof_every) {
    ph.next(raftq_node::kPhEncode);
    dump_profile(n);
  }
  return RAFTQ_OK;
}

// One Ready-loop iteration for all groups.  The turn grows host vectors (work lists, queues, entry storage): an
// allocation failure anywhere in it ends in the node's poisoned state and RAFTQ_ENOMEM, never in an exception crossing the
// extern "C" boundary.
int raftq_node_advance(raftq_node_t* n, uint64_t* n_published) {
  try {
    return node_advance_impl(n, n_published);
  } catch (const std::bad_alloc&) {
    return n ? poison(n, RAFTQ_ENOMEM, "advance (host allocation)") : RAFTQ_ENOMEM;
  } catch (...) {
    return n ? poison(n, RAFTQ_EHIP, "advance (unexpected exception)") : RAFTQ_EHIP;
  }
}

int raftq_node_poll(raftq_node_t* n, uint32_t to_peer, void* buf, uint64_t cap, uint64_t* len) {
  if (!n || !len) return RAFTQ_EINVAL;
  *len = 0;
  if (to_peer >= n->N) return nfail(n, RAFTQ_EINVAL, "poll: peer out of range");
  if (cap && !buf) return nfail(n, RAFTQ_EINVAL, "poll: null buffer");
  std::lock_guard<std::mutex> lk(n->mu);
  return poll_queue(n, n->outbound[to_peer], true, buf, cap, len);
}

int raftq_node_forward(raftq_node_t* from, uint32_t to_peer, raftq_node_t* to, uint64_t* moved) {
  if (moved) *moved = 0;
  if (!from) return RAFTQ_EINVAL;
  if (to_peer >= from->N) return nfail(from, RAFTQ_EINVAL, "forward: peer out of range");
  std::string taken;  // the queue's bytes leave under the sender's lock and arrive under the receiver's: never both held
  std::vector<uint64_t> ends;
  bool have_ends = false;
  try {
    std::lock_guard<std::mutex> lk(from->mu);
    PeerQueue& q = from->outbound[to_peer];
    if (q.head == 0) {
      taken.swap(q.bytes);
      have_ends = q.ends_ok;
      if (have_ends) ends.swap(q.ends);
    } else {
      taken.assign(q.bytes, q.head, std::string::npos);
    }
    q.reset();
  } catch (...) {
    return nfail(from, RAFTQ_ENOMEM, "forward: host allocation failed");
  }
  if (moved) *moved = taken.size();
  int rc = RAFTQ_OK;
  if (to && !taken.empty()) {
    if (have_ends && !ends.empty() && ends.back() == taken.size()) rc = deliver_impl(to, taken.data(), taken.size(), ends.data(), ends.size());
    else rc = raftq_node_deliver(to, taken.data(), taken.size());
  }
  if (taken.capacity() != 0 || ends.capacity() != 0) {  // the buffers go back to the sender's queue, empty
    std::lock_guard<std::mutex> lk(from->mu);
    PeerQueue& q = from->outbound[to_peer];
    if (q.spare.capacity() < taken.capacity()) q.spare.swap(taken);
    if (q.spare_ends.capacity() < ends.capacity()) q.spare_ends.swap(ends);
  }
  return rc;
}

int raftq_node_wal_enable(raftq_node_t* n) {
  if (!n) return RAFTQ_EINVAL;
  std::lock_guard<std::mutex> lk(n->mu);
  if (n->started) {
    n->errtext = "wal_enable: node already started";
    return RAFTQ_ESTATE;
  }
  n->wal_on = true;
  return RAFTQ_OK;
}

int raftq_node_wal_poll(raftq_node_t* n, void* buf, uint64_t cap, uint64_t* len) {
  if (!n || !len) return RAFTQ_EINVAL;
  *len = 0;
  if (cap && !buf) return nfail(n, RAFTQ_EINVAL, "wal_poll: null buffer");
  std::lock_guard<std::mutex> lk(n->mu);
  return poll_queue(n, n->wal_out, false, buf, cap, len);
}

int raftq_node_replay_wal(raftq_node_t* n, const void* wal, uint64_t len, int restore_hard_state, uint64_t* n_records) {
  if (!n) return RAFTQ_EINVAL;
  if (n_records) *n_records = 0;
  if (len && !wal) return nfail(n, RAFTQ_EINVAL, "replay_wal: null buffer");
  {
    std::lock_guard<std::mutex> lk(n->mu);
    if (n->started) {
      n->errtext = "replay_wal: node already started";
      return RAFTQ_ESTATE;
    }
  }
  if (len == 0) return RAFTQ_OK;
  // w.ReadAll() (raft.go:124): the frames that are whole (a torn tail is a crash in mid-append: ignored,
  // as wal.ReadAll tolerates io.ErrUnexpectedEOF on the last file), every record parsed and its CRC checked
  // against the chain on the GPU
  std::vector<uint64_t> off;
  std::vector<raftq_wal_rec_t> recs;
  uint64_t nf = 0, used = 0;
  try {
    off.resize((size_t)(len / 8) + 2);
    raftq_wire_scan_frames(wal, len, 0, off.data(), off.size() - 1, &nf, &used);
    recs.resize((size_t)nf);
  } catch (...) {
    return nfail(n, RAFTQ_ENOMEM, "replay_wal: host allocation failed");
  }
  if (nf == 0) return RAFTQ_OK;
  raftq_wal_counts_t cnt;
  const int rc = raftq_wal_decode(n->h, wal, used, off.data(), nf, 0, recs.data(), &cnt);
  if (rc != RAFTQ_OK) return nfail(n, rc, std::string("replay_wal: ") + raftq_last_error(n->h));
  if (cnt.n_valid != nf)  // the reference: log.Fatalf("raftsql: failed to read WAL (%v)", err) (raft.go:126)
    return nfail(n, RAFTQ_EINVAL,
                 "replay_wal: record " + std::to_string(cnt.n_valid) +
                     ((recs[cnt.n_valid].flags & RAFTQ_WAL_F_MALFORMED) ? " does not parse" : " fails its CRC (wal.ErrCRCMismatch)"));
  std::lock_guard<std::mutex> lk(n->mu);
  const char* base = (const char*)wal;
  for (uint64_t i = 0; i < nf; ++i) {
    const raftq_wal_rec_t& r = recs[i];
    if (r.kind == RAFTQ_WAL_ENTRY) {
      if (r.group >= n->G || !log_put(n, n->groups[r.group], r.index, r.term, base + r.data_off, r.data_len)) {
        n->errtext = "replay_wal: entry record " + std::to_string(i) + " has no place in its group's log";
        return RAFTQ_EINVAL;
      }
    } else if (r.kind == RAFTQ_WAL_STATE) {
      if (r.group >= n->G || r.vote > n->N) {
        n->errtext = "replay_wal: state record " + std::to_string(i) + " out of range";
        return RAFTQ_EINVAL;
      }
      Group& g = n->groups[r.group];
      g.wal_term = r.term;
      g.wal_vote = r.vote;
      g.wal_commit = r.index;
      if (restore_hard_state) {  // the reference does not: `_, _, ents, err := w.ReadAll()` (raft.go:124)
        g.hs_term = r.term;
        g.hs_vote = r.vote;
        g.hs_commit = r.index;
      }
    }
  }
  for (Group& g : n->groups) g.wal_upto = g.log.size();
  // appending continues this segment's chain
  n->wal_crc = cnt.last_crc;
  n->wal_head_written = true;
  if (n_records) *n_records = nf;
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
    if (len) *len = it.len;
    if (buf && cap) std::memcpy(buf, it.data, std::min<size_t>(cap, it.len));
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

int raftq_node_status_batch(raftq_node_t* n, uint64_t first_group, uint64_t count, raftq_node_status_t* st) {
  if (!n || (count && !st)) return RAFTQ_EINVAL;
  if (first_group > n->G || count > n->G - first_group) return nfail(n, RAFTQ_EINVAL, "status_batch: range out of bounds");
  std::lock_guard<std::mutex> lk(n->mu);
  for (uint64_t i = 0; i < count; ++i) {
    const Group& g = n->groups[first_group + i];
    raftq_node_status_t& o = st[i];
    std::memset(&o, 0, sizeof(o));
    o.term = g.term;
    o.commit = g.committed;
    o.last_index = g.log.size();
    o.applied = g.applied;
    o.lead = g.lead;
    o.vote = g.vote;
    o.role = g.role;
  }
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
  if (len) *len = e.len;
  if (term) *term = e.term;
  if (buf && cap) std::memcpy(buf, e.data, std::min<size_t>(cap, e.len));
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
  if (n->profiling && n->prof_turns) dump_profile(n);
  raftq_destroy(n->h);
  delete n;
}

}  // extern "C"

// ---- raftq_crank: the nodes of one process turned in lock-step, each on a thread of its own ------------------------------
// What the reference's tests do with three raftNodes over loopback (raftsql_test.go:11-35) and what bench.py's node leg
// measures: every live node's Tick + Ready-loop iteration at once, then every node's inbound buffer filled from the
// others' queues (per ADDRESSEE, senders in slot order from `first_sender` on: what a node receives does not depend on
// thread timing, and a caller that moves first_sender along gives no slot the first word every time).  One call
// per cluster step; the threads live as long as the crank (a Python thread pool costs a GIL hand-off per node and step,
// and its workers do not stay on the node's core).
struct raftq_crank {
  std::vector<raftq_node_t*> nodes;
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t epoch = 0;
  uint32_t pending = 0;
  int phase = 0;  // 1 = turn, 2 = transport, -1 = quit
  uint32_t live = 0, first_sender = 0;
  int tick = 0;
  const uint8_t* lost = nullptr;
  std::vector<int> rc;
  std::vector<uint64_t> published;
  double seconds[2] = {0, 0};  // wall time of the steps' two halves so far: the turns, the transport
  bool shards = false;         // raftq_shards_create: the members are shards of ONE node (no transport between them)
};

namespace {
void crank_worker(raftq_crank* c, uint32_t p, int cpu) {
#if defined(__linux__)
  if (cpu >= 0) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
  }
#else
  (void)cpu;
#endif
  uint64_t seen = 0;
  for (;;) {
    int phase;
    {
      std::unique_lock<std::mutex> lk(c->mu);
      c->cv_go.wait(lk, [&] { return c->epoch != seen; });
      seen = c->epoch;
      phase = c->phase;
    }
    if (phase < 0) return;
    const uint32_t n = (uint32_t)c->nodes.size();
    int rc = RAFTQ_OK;
    if (phase == 1 && (c->live >> p & 1u)) {
      if (c->tick) rc = raftq_node_tick(c->nodes[p]);
      uint64_t pub = 0;
      if (rc == RAFTQ_OK) rc = raftq_node_advance(c->nodes[p], &pub);
      c->published[p] = pub;
    } else if (phase == 2) {
      for (uint32_t k = 0; k < n && rc == RAFTQ_OK; ++k) {
        const uint32_t from = (c->first_sender + k) % n;
        if (from == p || !(c->live >> from & 1u)) continue;
        const bool gone = !(c->live >> p & 1u) || (c->lost && c->lost[(size_t)p * n + from]);
        rc = raftq_node_forward(c->nodes[from], p, gone ? nullptr : c->nodes[p], nullptr);
      }
    }
    {
      std::lock_guard<std::mutex> lk(c->mu);
      if (rc != RAFTQ_OK && c->rc[p] == RAFTQ_OK) c->rc[p] = rc;
      if (--c->pending == 0) c->cv_done.notify_one();
    }
  }
}

void crank_run(raftq_crank* c, int phase) {
  std::unique_lock<std::mutex> lk(c->mu);
  c->phase = phase;
  c->pending = (uint32_t)c->nodes.size();
  c->epoch++;
  c->cv_go.notify_all();
  c->cv_done.wait(lk, [&] { return c->pending == 0; });
}
}  // namespace

extern "C" {

int raftq_crank_create(raftq_node_t* const* nodes, uint32_t n, const int* cpus, raftq_crank_t** out) {
  if (!out) return RAFTQ_EINVAL;
  *out = nullptr;
  if (!nodes || n == 0 || n > 32) return RAFTQ_EINVAL;
  for (uint32_t p = 0; p < n; ++p)  // nodes[p] is peer slot p of an n-peer cluster, or NULL (a stopped node: never live)
    if (nodes[p] && (nodes[p]->N != n || nodes[p]->self != p)) return RAFTQ_EINVAL;
  raftq_crank* c = new (std::nothrow) raftq_crank();
  if (!c) return RAFTQ_ENOMEM;
  try {
    c->nodes.assign(nodes, nodes + n);
    c->rc.assign(n, RAFTQ_OK);
    c->published.assign(n, 0);
    for (uint32_t p = 0; p < n; ++p) c->threads.emplace_back(crank_worker, c, p, cpus ? cpus[p] : -1);
  } catch (...) {
    raftq_crank_destroy(c);
    return RAFTQ_ENOMEM;
  }
  *out = c;
  return RAFTQ_OK;
}

int raftq_crank_step(raftq_crank_t* c, uint32_t live_mask, int tick, const uint8_t* lost, uint32_t first_sender, uint64_t* published,
                     int* node_rc) {
  if (!c || c->shards) return RAFTQ_EINVAL;  // (shards have no transport between them: raftq_shards_turn)
  const uint32_t n = (uint32_t)c->nodes.size();
  for (uint32_t p = 0; p < 32; ++p)
    if ((live_mask >> p & 1u) && (p >= n || !c->nodes[p])) return RAFTQ_EINVAL;
  c->live = live_mask;
  c->first_sender = first_sender % n;
  c->tick = tick;
  c->lost = lost;
  std::fill(c->rc.begin(), c->rc.end(), RAFTQ_OK);
  std::fill(c->published.begin(), c->published.end(), 0);
  const auto t0 = std::chrono::steady_clock::now();
  crank_run(c, 1);
  const auto t1 = std::chrono::steady_clock::now();
  crank_run(c, 2);
  const auto t2 = std::chrono::steady_clock::now();
  c->seconds[0] += std::chrono::duration<double>(t1 - t0).count();
  c->seconds[1] += std::chrono::duration<double>(t2 - t1).count();
  int first = RAFTQ_OK;
  for (uint32_t p = 0; p < n; ++p) {
    if (published) published[p] = c->published[p];
    if (node_rc) node_rc[p] = c->rc[p];
    if (first == RAFTQ_OK) first = c->rc[p];
  }
  return first;
}

// The SHARDS of one node: K handles that are the same peer slot of the same cluster for K disjoint sets of groups, a thread each
// (the crank's threads, without its transport: a shard's frames are its own).  One handle turns its groups on one host thread
// -- 3.5 of a 4.0 ms turn at 32,768 groups are host work -- and groups are independent: the reference runs a
// raftNode goroutine per group (raft.go:204-246); this is the same, batched K ways.
int raftq_shards_create(raftq_node_t* const* shards, uint32_t k, const int* cpus, raftq_crank_t** out) {
  if (!out) return RAFTQ_EINVAL;
  *out = nullptr;
  if (!shards || k == 0 || k > 32) return RAFTQ_EINVAL;
  for (uint32_t i = 0; i < k; ++i) {
    if (!shards[i] || shards[i]->N != shards[0]->N || shards[i]->self != shards[0]->self) return RAFTQ_EINVAL;
    for (uint32_t j = 0; j < i; ++j)
      if (shards[j] == shards[i]) return RAFTQ_EINVAL;  // a handle is turned by one thread at a time
  }
  raftq_crank* c = new (std::nothrow) raftq_crank();
  if (!c) return RAFTQ_ENOMEM;
  try {
    c->nodes.assign(shards, shards + k);
    c->rc.assign(k, RAFTQ_OK);
    c->published.assign(k, 0);
    c->shards = true;
    for (uint32_t i = 0; i < k; ++i) c->threads.emplace_back(crank_worker, c, i, cpus ? cpus[i] : -1);
  } catch (...) {
    raftq_crank_destroy(c);
    return RAFTQ_ENOMEM;
  }
  *out = c;
  return RAFTQ_OK;
}

int raftq_shards_turn(raftq_crank_t* c, int tick, uint64_t* published, int* shard_rc) {
  if (!c || !c->shards) return RAFTQ_EINVAL;
  const uint32_t k = (uint32_t)c->nodes.size();
  c->live = k == 32 ? ~0u : ((1u << k) - 1u);
  c->first_sender = 0;
  c->tick = tick;
  c->lost = nullptr;
  std::fill(c->rc.begin(), c->rc.end(), RAFTQ_OK);
  std::fill(c->published.begin(), c->published.end(), 0);
  const auto t0 = std::chrono::steady_clock::now();
  crank_run(c, 1);  // every shard's Tick + Ready iteration at once; no transport phase: a shard's queues are its own
  c->seconds[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int first = RAFTQ_OK;
  for (uint32_t i = 0; i < k; ++i) {
    if (published) published[i] = c->published[i];
    if (shard_rc) shard_rc[i] = c->rc[i];
    if (first == RAFTQ_OK) first = c->rc[i];
  }
  return first;
}

void raftq_crank_seconds(const raftq_crank_t* c, double* turns, double* transport) {
  if (turns) *turns = c ? c->seconds[0] : 0;
  if (transport) *transport = c ? c->seconds[1] : 0;
}

void raftq_crank_destroy(raftq_crank_t* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->phase = -1;
    c->epoch++;
    c->cv_go.notify_all();
  }
  for (std::thread& t : c->threads)
    if (t.joinable()) t.join();
  delete c;
}

}  // extern "C"
