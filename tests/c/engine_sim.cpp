// engine_sim.cpp -- TEST INFRASTRUCTURE: the part of the C-ABI that raftq_node.cpp and raftq_pipe.cpp call, answered on the
// CPU by the oracle (oracle/*.c).  Linked with those two translation units (and nothing of the HIP side) into
// tests/c/libraftq_hostsim.so, it lets the CPU suite run the node / pipe suites -- the same Python tests the GPU box runs
// against libraftq.so -- under AddressSanitizer + UBSan: 1,900 lines of host C++ (mutexes, condition variables, a
// background thread, arenas, queues) that cannot be run under ASan beside the HIP runtime (profiles/r03/sanitizers_host_cpp.txt).
// Nothing of the product links, loads or ships this file; it is not a CPU fallback: it exists only inside the test
// library, which the product's loader never opens (tests/conftest.py swaps the path for tests/test_hostsim.py's sub-runs).
//
// Semantics follow include/*.h: same return codes, same all-or-nothing rules, same list orders.  The arithmetic is the
// oracle's, which tests/test_*_gpu.py hold equal to the device's word for word.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "raftq.h"
#include "raftq_step.h"
#include "raftq_wire.h"

extern "C" {
#include "raftq_oracle.h"
}

struct raftq {
  uint64_t G = 0;
  uint32_t N = 0, self = 0;
  bool wal_pending = false;
  int wal_rc = 0;
  raftq_wal_counts_t wal_counts{};
  std::string wal_err;
  std::vector<uint8_t> role, votes, action;
  std::vector<uint32_t> elapsed, vote, lead;
  std::vector<uint64_t> term, last_index, last_term, committed, first_idx, match;
  bool have_terms = false, ticked = false, msg_flags = false;
  std::vector<uint32_t> tl_hups, tl_beats;  // raftq_tick_collect_lists: left in place
  std::vector<uint64_t> tl_map;
  bool tl_bitmap = false, tl_valid = false;
  uint32_t election_tick = 10, heartbeat_tick = 1;
  uint64_t seed = 0x1000, tick_no = 0;
  std::vector<raftq_msg_t> stage;
  std::vector<raftq_step_out_t> outs;
  std::vector<raftq_step_out_c_t> outs_c;
  std::vector<raftq_step_out_s_t> outs_s;
  bool compact = false;
  uint64_t n_out = 0;
  std::vector<raftq_advance_t> adv;  // the advance list of the last CHANGED sweep, ascending group
  std::vector<raftq_advance16_t> adv16;  // ... of the last packed turn, in the 16-byte layout
  uint32_t adv16_count = 0;
  bool have_adv = false;
  std::string err;
  rq_node_state_t state() {
    rq_node_state_t s;
    s.G = G;
    s.ld = G;
    s.n = (int)N;
    s.self = self;
    s.role = role.data();
    s.elapsed = elapsed.data();
    s.term = term.data();
    s.vote = vote.data();
    s.lead = lead.data();
    s.last_index = last_index.data();
    s.last_term = last_term.data();
    s.committed = committed.data();
    s.first_idx = first_idx.data();
    s.match = match.data();
    s.votes = votes.data();
    return s;
  }
};

namespace {
thread_local std::string g_err;
int fail(raftq_t* h, int code, const std::string& msg) {
  g_err = msg;
  if (h) h->err = msg;
  return code;
}

// one sweep over all groups with `flags`; fills the advance list when a commit sweep ran
void sweep(raftq_t* h, unsigned flags, raftq_counts_t* counts) {
  raftq_counts_t c{0, 0, 0};
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  h->have_adv = false;
  if (commit) {
    std::vector<uint64_t> out(h->G);
    c.n_changed = rq_oracle_commit_advance(h->match.data(), h->G, (int)h->N, h->G, h->committed.data(), (flags & RAFTQ_SWEEP_GATED) ? 1 : 0,
                                           h->first_idx.data(), out.data());
    h->adv.clear();
    for (uint64_t g = 0; g < h->G; ++g)
      if (out[g] != h->committed[g]) h->adv.push_back(raftq_advance_t{g, h->committed[g], out[g]});
    h->have_adv = true;
    if (!(flags & RAFTQ_SWEEP_NO_ADOPT)) h->committed.swap(out);
  }
  if (flags & RAFTQ_SWEEP_VOTES) {
    std::vector<uint8_t> oc(h->G);
    rq_oracle_vote_tally(h->votes.data(), h->G, (int)h->N, h->G, oc.data(), &c.n_won, &c.n_lost);
  }
  if (counts) *counts = c;
}

template <typename Rec, typename Adv>
int cycle(raftq_t* h, const char* who, const Rec* d, uint64_t n, const raftq_vote_delta_t* vd, uint64_t nv, unsigned flags, Adv* out,
          uint64_t cap, uint64_t* n_adv, raftq_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if ((n && !d) || (nv && !vd)) return fail(h, RAFTQ_EINVAL, std::string(who) + ": null array with non-zero length");
  const bool trusted = flags & RAFTQ_CYCLE_TRUSTED;
  flags &= ~(RAFTQ_CYCLE_TRUSTED | RAFTQ_CYCLE_SEGMENTED);  // (segments: here the list is always one)
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  if (!commit && !(flags & RAFTQ_SWEEP_VOTES)) return fail(h, RAFTQ_EINVAL, std::string(who) + ": nothing to sweep");
  if ((flags & RAFTQ_SWEEP_GATED) && !h->have_terms) return fail(h, RAFTQ_ESTATE, std::string(who) + ": gated sweep before raftq_load_terms");
  auto ok_d = [&](const Rec& r) { return (uint64_t)r.group < h->G && r.peer < h->N; };
  auto ok_v = [&](const raftq_vote_delta_t& r) { return r.group < h->G && r.peer < h->N && (r.vote == 1 || r.vote == 2); };
  bool bad = false;
  for (uint64_t i = 0; i < n; ++i) bad |= !ok_d(d[i]);
  for (uint64_t i = 0; i < nv; ++i) bad |= !ok_v(vd[i]);
  if (bad && !trusted) {
    if (n_adv) *n_adv = 0;
    if (counts) *counts = raftq_counts_t{0, 0, 0};
    return fail(h, RAFTQ_EINVAL, "a delta is out of range; nothing applied");
  }
  for (uint64_t i = 0; i < n; ++i)
    if (ok_d(d[i])) {
      uint64_t& m = h->match[(size_t)d[i].peer * h->G + d[i].group];
      m = std::max<uint64_t>(m, d[i].match);
    }
  for (uint64_t i = 0; i < nv; ++i)
    if (ok_v(vd[i])) {
      uint8_t& v = h->votes[(size_t)vd[i].peer * h->G + vd[i].group];
      if (v != 1 && v != 2) v = vd[i].vote;  // the first response wins
    }
  sweep(h, flags, counts);
  if (commit && (out || n_adv || cap)) {
    const uint64_t total = h->adv.size();
    if (n_adv) *n_adv = total;
    if constexpr (sizeof(Adv) != sizeof(raftq_advance_t)) {  // the list left in place (raftq_last_advance_segments)
      try {
        h->adv16.resize((size_t)total);
      } catch (...) {
        return fail(h, RAFTQ_ENOMEM, std::string(who) + ": host allocation failed");
      }
      for (uint64_t i = 0; i < total; ++i) {
        const raftq_advance_t& a = h->adv[i];
        const uint64_t by = a.new_commit - a.old_commit;
        h->adv16[i] = raftq_advance16_t{a.new_commit, (uint32_t)a.group, by > 0xfffffffeull ? 0xffffffffu : (uint32_t)by};
      }
      h->adv16_count = (uint32_t)total;
    }
    if (out)
      for (uint64_t i = 0; i < std::min(total, cap); ++i) {
        const raftq_advance_t& a = h->adv[i];
        if constexpr (sizeof(Adv) == sizeof(raftq_advance_t)) {
          out[i] = a;
        } else {
          const uint64_t by = a.new_commit - a.old_commit;
          out[i].new_commit = a.new_commit;
          out[i].group = (uint32_t)a.group;
          out[i].advanced_by = by > 0xfffffffeull ? 0xffffffffu : (uint32_t)by;
        }
      }
  }
  return bad ? fail(h, RAFTQ_EINVAL, "a record was out of range; that record was dropped, every other record of the turn was applied")
             : RAFTQ_OK;
}

int tick_list(raftq_t* h, const char* who, uint8_t want, uint64_t* groups, uint64_t cap, uint64_t* n) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!n) return fail(h, RAFTQ_EINVAL, std::string(who) + ": null count");
  if (!h->ticked) return fail(h, RAFTQ_ESTATE, std::string(who) + ": no raftq_tick yet");
  if (cap && !groups) return fail(h, RAFTQ_EINVAL, std::string(who) + ": null out with cap > 0");
  uint64_t k = 0;
  for (uint64_t g = 0; g < h->G; ++g)
    if (h->action[g] == want) {
      if (k < cap) groups[k] = g;
      ++k;
    }
  *n = k;
  return RAFTQ_OK;
}
}  // namespace

extern "C" {

int raftq_abi_version(void) { return RAFTQ_ABI_VERSION; }
uint32_t raftq_quorum(uint32_t n_peers) { return n_peers / 2 + 1; }
int raftq_device_count(int* n) {
  if (!n) return RAFTQ_EINVAL;
  *n = 1;  // "device 0" = this stand-in
  return RAFTQ_OK;
}
const char* raftq_last_error(const raftq_t* h) { return h ? h->err.c_str() : g_err.c_str(); }
uint64_t raftq_groups(const raftq_t* h) { return h ? h->G : 0; }
uint32_t raftq_peers(const raftq_t* h) { return h ? h->N : 0; }

int raftq_create(int device, uint64_t n_groups, uint32_t n_peers, raftq_t** out) {
  if (!out) return fail(nullptr, RAFTQ_EINVAL, "raftq_create: null out");
  *out = nullptr;
  if (n_groups == 0 || n_groups > (1ull << 40)) return fail(nullptr, RAFTQ_EINVAL, "raftq_create: n_groups out of range");
  if (n_peers < 1 || n_peers > RAFTQ_MAX_PEERS) return fail(nullptr, RAFTQ_EINVAL, "raftq_create: n_peers must be 1..9");
  if (device != 0) return fail(nullptr, RAFTQ_ENODEV, "raftq_create: device index out of range");
  raftq_t* h = new (std::nothrow) raftq();
  if (!h) return fail(nullptr, RAFTQ_ENOMEM, "raftq_create: host allocation failed");
  try {
    h->G = n_groups;
    h->N = n_peers;
    const size_t G = n_groups, NG = (size_t)n_peers * n_groups;
    h->role.assign(G, 0);
    h->action.assign(G, 0);
    h->elapsed.assign(G, 0);
    h->vote.assign(G, 0);
    h->lead.assign(G, 0);
    h->term.assign(G, 0);
    h->last_index.assign(G, 0);
    h->last_term.assign(G, 0);
    h->committed.assign(G, 0);
    h->first_idx.assign(G, 0);
    h->match.assign(NG, 0);
    h->votes.assign(NG, 0);
  } catch (...) {
    delete h;
    return fail(nullptr, RAFTQ_ENOMEM, "raftq_create: host allocation failed");
  }
  *out = h;
  return RAFTQ_OK;
}
void raftq_destroy(raftq_t* h) { delete h; }

int raftq_host_alloc(void** p, uint64_t bytes) {
  if (!p || bytes == 0) return fail(nullptr, RAFTQ_EINVAL, "raftq_host_alloc: null pointer or zero size");
  *p = malloc((size_t)bytes);
  return *p ? RAFTQ_OK : fail(nullptr, RAFTQ_ENOMEM, "raftq_host_alloc: out of memory");
}
void raftq_host_free(void* p) { free(p); }

int raftq_load_match(raftq_t* h, const uint64_t* match, const uint64_t* committed) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!match && !committed) return fail(h, RAFTQ_EINVAL, "raftq_load_match: nothing to load");
  if (match) std::copy(match, match + (size_t)h->N * h->G, h->match.begin());
  if (committed) std::copy(committed, committed + h->G, h->committed.begin());
  return RAFTQ_OK;
}
int raftq_load_terms(raftq_t* h, const uint64_t* cur_term, const uint64_t* first_idx_cur_term) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!cur_term || !first_idx_cur_term) return fail(h, RAFTQ_EINVAL, "raftq_load_terms: null argument");
  for (uint64_t g = 0; g < h->G; ++g) h->first_idx[g] = cur_term[g] == 0 ? 0 : first_idx_cur_term[g];
  h->have_terms = true;
  return RAFTQ_OK;
}
int raftq_set_self(raftq_t* h, uint32_t self_peer) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (self_peer >= h->N) return fail(h, RAFTQ_EINVAL, "raftq_set_self: self_peer out of range");
  h->self = self_peer;
  return RAFTQ_OK;
}
int raftq_step_set_msg_flags(raftq_t* h, int on) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  h->msg_flags = on != 0;
  return RAFTQ_OK;
}
int raftq_set_timers(raftq_t* h, uint32_t election_tick, uint32_t heartbeat_tick, uint64_t seed) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (election_tick == 0 || heartbeat_tick == 0) return fail(h, RAFTQ_EINVAL, "raftq_set_timers: ticks must be >= 1");
  h->election_tick = election_tick;
  h->heartbeat_tick = heartbeat_tick;
  h->seed = seed;
  return RAFTQ_OK;
}
int raftq_load_node(raftq_t* h, const uint64_t* term, const uint32_t* vote, const uint32_t* lead, const uint64_t* last_index,
                    const uint64_t* last_term) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (term) std::copy(term, term + h->G, h->term.begin());
  if (vote) std::copy(vote, vote + h->G, h->vote.begin());
  if (lead) std::copy(lead, lead + h->G, h->lead.begin());
  if (last_index) std::copy(last_index, last_index + h->G, h->last_index.begin());
  if (last_term) std::copy(last_term, last_term + h->G, h->last_term.begin());
  h->have_terms = true;  // the gate is maintained by Step / the log-tail reports from here on
  return RAFTQ_OK;
}

int raftq_cycle(raftq_t* h, const raftq_delta_t* deltas, uint64_t n_deltas, const raftq_vote_delta_t* vote_deltas,
                uint64_t n_vote_deltas, unsigned flags, raftq_advance_t* advances_out, uint64_t cap, uint64_t* n_advanced,
                raftq_counts_t* counts) {
  return cycle(h, "raftq_cycle", deltas, n_deltas, vote_deltas, n_vote_deltas, flags, advances_out, cap, n_advanced, counts);
}
int raftq_cycle_packed(raftq_t* h, const raftq_delta16_t* deltas, uint64_t n_deltas, const raftq_vote_delta_t* vote_deltas,
                       uint64_t n_vote_deltas, unsigned flags, raftq_advance16_t* advances_out, uint64_t cap,
                       uint64_t* n_advanced, raftq_counts_t* counts) {
  if (h && h->G > (1ull << 32)) return fail(h, RAFTQ_EINVAL, "raftq_cycle_packed: more than 2^32 groups");
  return cycle(h, "raftq_cycle_packed", deltas, n_deltas, vote_deltas, n_vote_deltas, flags, advances_out, cap, n_advanced, counts);
}
int raftq_last_advance_segments(raftq_t* h, const raftq_advance16_t** recs, const uint32_t** counts, uint32_t* n_segments, uint64_t* stride) {
  if (!h || !recs || !counts || !n_segments || !stride) return fail(h, RAFTQ_EINVAL, "raftq_last_advance_segments: null argument");
  *recs = h->adv16.data();
  *counts = &h->adv16_count;
  *n_segments = 1;
  *stride = 0;
  return RAFTQ_OK;
}
int raftq_collect_changed(raftq_t* h, raftq_advance_t* out, uint64_t cap, uint64_t* n) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!n) return fail(h, RAFTQ_EINVAL, "raftq_collect_changed: null count");
  if (!h->have_adv) return fail(h, RAFTQ_ESTATE, "raftq_collect_changed: last sweep did not set RAFTQ_SWEEP_CHANGED");
  if (cap && !out) return fail(h, RAFTQ_EINVAL, "raftq_collect_changed: null out with cap > 0");
  *n = h->adv.size();
  std::copy(h->adv.begin(), h->adv.begin() + (size_t)std::min<uint64_t>(cap, h->adv.size()), out);
  return RAFTQ_OK;
}

int raftq_tick(raftq_t* h, raftq_tick_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  uint64_t hup = 0, beat = 0;
  rq_oracle_tick(h->role.data(), h->elapsed.data(), h->G, h->election_tick, h->heartbeat_tick, h->seed, h->tick_no++, h->action.data(), &hup,
                 &beat);
  h->ticked = true;
  if (counts) {
    counts->n_hup = hup;
    counts->n_beat = beat;
  }
  return RAFTQ_OK;
}
int raftq_collect_hups(raftq_t* h, uint64_t* groups, uint64_t cap, uint64_t* n) { return tick_list(h, "raftq_collect_hups", 1, groups, cap, n); }
int raftq_collect_beats(raftq_t* h, uint64_t* groups, uint64_t cap, uint64_t* n) { return tick_list(h, "raftq_collect_beats", 2, groups, cap, n); }
int raftq_tick_collect(raftq_t* h, uint64_t* hups, uint64_t hup_cap, uint64_t* n_hup, uint64_t* beats, uint64_t beat_cap, uint64_t* n_beat) {
  if (int rc = raftq_tick(h, nullptr)) return rc;
  if (int rc = tick_list(h, "raftq_tick_collect", 1, hups, hup_cap, n_hup)) return rc;
  return tick_list(h, "raftq_tick_collect", 2, beats, beat_cap, n_beat);
}

int raftq_tick_collect_lists(raftq_t* h, unsigned flags, uint64_t hup_cap, uint64_t beat_cap, uint64_t* n_hup, uint64_t* n_beat) {
  if (!h || !n_hup || !n_beat) return fail(h, RAFTQ_EINVAL, "raftq_tick_collect_lists: null argument");
  if (flags & ~RAFTQ_TICK_BEAT_BITMAP) return fail(h, RAFTQ_EINVAL, "raftq_tick_collect_lists: unknown flag");
  if (int rc = raftq_tick(h, nullptr)) return rc;
  const bool bitmap = flags & RAFTQ_TICK_BEAT_BITMAP;
  h->tl_hups.clear();
  h->tl_beats.clear();
  h->tl_map.assign(bitmap ? (h->G + 63) / 64 : 0, 0);
  uint64_t nh = 0, nb = 0;
  for (uint64_t g = 0; g < h->G; ++g) {
    if (h->action[g] == 1) {
      if (nh++ < hup_cap) h->tl_hups.push_back((uint32_t)g);
    } else if (h->action[g] == 2) {
      if (bitmap) h->tl_map[g / 64] |= 1ull << (g % 64);
      else if (nb < beat_cap) h->tl_beats.push_back((uint32_t)g);
      ++nb;
    }
  }
  *n_hup = nh;
  *n_beat = nb;
  h->tl_bitmap = bitmap;
  h->tl_valid = true;
  return RAFTQ_OK;
}
int raftq_last_tick_lists(raftq_t* h, const uint32_t** hups, uint64_t* n_hups, const uint32_t** beats, uint64_t* n_beats,
                          const uint64_t** beat_bitmap, uint64_t* bitmap_words) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!h->tl_valid) return fail(h, RAFTQ_ESTATE, "raftq_last_tick_lists: no raftq_tick_collect_lists before it");
  if (hups) *hups = h->tl_hups.data();
  if (n_hups) *n_hups = h->tl_hups.size();
  if (beats) *beats = h->tl_bitmap ? nullptr : h->tl_beats.data();
  if (n_beats) *n_beats = h->tl_bitmap ? 0 : h->tl_beats.size();
  if (beat_bitmap) *beat_bitmap = h->tl_bitmap ? h->tl_map.data() : nullptr;
  if (bitmap_words) *bitmap_words = h->tl_bitmap ? h->tl_map.size() : 0;
  return RAFTQ_OK;
}

int raftq_step_stage(raftq_t* h, uint64_t n, raftq_msg_t** msgs) {
  if (!h || !msgs) return fail(h, RAFTQ_EINVAL, "raftq_step_stage: null argument");
  try {
    h->stage.resize((size_t)std::max<uint64_t>(n, 1));
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_step_stage: host allocation failed");
  }
  *msgs = h->stage.data();
  return RAFTQ_OK;
}
int raftq_step_batch(raftq_t* h, const raftq_msg_t* msgs, uint64_t n, raftq_step_out_t* out, raftq_step_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_step_counts_t{0, 0};
  h->n_out = 0;
  if (n == 0) return RAFTQ_OK;
  if (!msgs) return fail(h, RAFTQ_EINVAL, "raftq_step_batch: null argument");
  for (uint64_t i = 0; i < n; ++i) {
    const uint8_t t = msgs[i].type;
    const bool local = t == RAFTQ_MSG_HUP || t == RAFTQ_MSG_BEAT;
    const bool known = local || t == RAFTQ_MSG_APP || t == RAFTQ_MSG_APP_RESP || t == RAFTQ_MSG_VOTE || t == RAFTQ_MSG_VOTE_RESP ||
                       t == RAFTQ_MSG_HEARTBEAT || t == RAFTQ_MSG_HEARTBEAT_RESP;
    if (msgs[i].group >= h->G || !known || (!local && msgs[i].from >= h->N))
      return fail(h, RAFTQ_EINVAL, "a message is malformed (group / from out of range or unknown type); nothing applied");
  }
  try {
    h->outs.resize((size_t)n);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_step_batch: host allocation failed");
  }
  rq_node_state_t s = h->state();
  if (h->msg_flags) {
    rq_oracle_step_batch(&s, msgs, (size_t)n, h->outs.data());
  } else {  // the pad bytes are padding to a handle that has not opted in (raftq_step_set_msg_flags)
    std::vector<raftq_msg_t> plain(msgs, msgs + n);
    for (raftq_msg_t& m : plain) {
      m._pad[0] = m._pad[1] = 0;
      m._resv = 0;
    }
    rq_oracle_step_batch(&s, plain.data(), (size_t)n, h->outs.data());
  }
  h->n_out = n;
  if (out) std::copy(h->outs.begin(), h->outs.begin() + (size_t)n, out);
  if (counts) {
    std::vector<uint64_t> gs((size_t)n);
    for (uint64_t i = 0; i < n; ++i) gs[i] = msgs[i].group;
    std::sort(gs.begin(), gs.end());
    counts->n_msgs = n;
    counts->n_groups_touched = (uint64_t)(std::unique(gs.begin(), gs.end()) - gs.begin());
  }
  return RAFTQ_OK;
}
int raftq_step_results(raftq_t* h, const raftq_step_out_t** out, uint64_t* n) {
  if (!h || !out || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_results: null argument");
  *out = h->outs.data();
  *n = h->n_out;
  return RAFTQ_OK;
}
int raftq_step_set_compact(raftq_t* h, int on) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  h->compact = on != 0;
  return RAFTQ_OK;
}
// the 40-byte form of the last batch's results (include/raftq_step.h: group / addressee are the message's, log_term and
// last_index share a slot)
int raftq_step_results_c(raftq_t* h, const raftq_step_out_c_t** out, uint64_t* n) {
  if (!h || !out || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_results_c: null argument");
  try {
    h->outs_c.resize((size_t)h->n_out);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_step_results_c: host allocation failed");
  }
  for (uint64_t i = 0; i < h->n_out; ++i) {
    const raftq_step_out_t& o = h->outs[i];
    raftq_step_out_c_t c;
    memset(&c, 0, sizeof(c));
    c.term = o.term;
    c.index = o.index;
    c.commit = o.commit;
    c.aux = (o.type == RAFTQ_OUT_CAMPAIGN || o.type == RAFTQ_OUT_BECAME_LEADER) ? o.log_term : o.last_index;
    c.vote = (uint8_t)o.vote;
    c.lead = (uint8_t)o.lead;
    c.type = o.type;
    c.reject = o.reject;
    c.flags = o.flags;
    c.role = o.role;
    h->outs_c[i] = c;
  }
  *out = h->outs_c.data();
  *n = h->n_out;
  return RAFTQ_OK;
}
// ... and the 32-byte form (raftq_step_set_compact(h, 2)): no aux; a campaign's log_term rides in `commit`
int raftq_step_results_s(raftq_t* h, const raftq_step_out_s_t** out, uint64_t* n) {
  if (!h || !out || !n) return fail(h, RAFTQ_EINVAL, "raftq_step_results_s: null argument");
  try {
    h->outs_s.resize((size_t)h->n_out);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_step_results_s: host allocation failed");
  }
  for (uint64_t i = 0; i < h->n_out; ++i) {
    const raftq_step_out_t& o = h->outs[i];
    raftq_step_out_s_t c;
    memset(&c, 0, sizeof(c));
    c.term = o.term;
    c.index = o.index;
    c.commit = o.type == RAFTQ_OUT_CAMPAIGN ? o.log_term : o.commit;
    c.vote = (uint8_t)o.vote;
    c.lead = (uint8_t)o.lead;
    c.type = o.type;
    c.reject = o.reject;
    c.flags = o.flags;
    c.role = o.role;
    h->outs_s[i] = c;
  }
  *out = h->outs_s.data();
  *n = h->n_out;
  return RAFTQ_OK;
}
int raftq_apply_log_deltas(raftq_t* h, const raftq_log_delta_t* d, uint64_t n, uint64_t* committed_out) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (n == 0) return RAFTQ_OK;
  if (!d) return fail(h, RAFTQ_EINVAL, "raftq_apply_log_deltas: null argument");
  for (uint64_t i = 0; i < n; ++i)
    if (d[i].group >= h->G) return fail(h, RAFTQ_EINVAL, "a log delta is out of range; nothing applied");
  rq_node_state_t s = h->state();
  rq_oracle_apply_log_deltas(&s, d, (size_t)n, committed_out);
  return RAFTQ_OK;
}

int raftq_apply_log_deltas_nowait(raftq_t* h, const raftq_log_delta_t* d, uint64_t n) { return raftq_apply_log_deltas(h, d, n, nullptr); }

int raftq_wire_encode(raftq_t* h, const raftq_wire_msg_t* msgs, uint64_t n, const raftq_wire_ent_t* ents, uint64_t n_ents, const void* pool,
                      uint64_t pool_bytes, void* out, uint64_t cap, uint64_t* frame_off, raftq_wire_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (n == 0) {
    if (frame_off) frame_off[0] = 0;
    return RAFTQ_OK;
  }
  if (!msgs || (n_ents && !ents) || (pool_bytes && !pool) || (cap && !out)) return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: null argument");
  for (uint64_t i = 0; i < n; ++i) {
    const raftq_wire_msg_t& m = msgs[i];
    bool bad = m.to >= 255 || m.from >= 255 || (m.n_ents != 0 && (uint64_t)m.ent_first + m.n_ents > n_ents);
    for (uint32_t k = 0; !bad && k < m.n_ents; ++k) {
      const raftq_wire_ent_t& e = ents[m.ent_first + k];
      bad = e.data_len != 0 && (e.data_off > pool_bytes || e.data_len > pool_bytes - e.data_off);
    }
    if (bad) return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: a message has to / from >= 255, an entry range outside ents[], or a payload outside the pool");
  }
  const uint64_t need = rq_wire_encode(msgs, n, ents, (const uint8_t*)pool, nullptr, 0, nullptr);
  if (counts) {
    counts->n_msgs = n;
    counts->n_ents = n_ents;
    counts->bytes = need;
  }
  if (need > cap) return fail(h, RAFTQ_EINVAL, "raftq_wire_encode: out is too small (counts->bytes is the size needed)");
  rq_wire_encode(msgs, n, ents, (const uint8_t*)pool, (uint8_t*)out, cap, frame_off);
  return RAFTQ_OK;
}
// raftq_propose_frames from its parts, as include/raftq_wire.h states it: the records are validated (nothing applied on a
// refusal), then for every group appendEntry -- the oracle's raftq_apply_log_deltas with the new tail -- and the N - 1 MsgApps
// bcastAppend sends, built the way raftq_node.cpp's send_append builds them; then the oracle's encoder over msgs[] + those.
int raftq_propose_frames(raftq_t* h, const raftq_prop_t* props, uint64_t n_props, const raftq_prop_ent_t* prop_ents, uint64_t n_prop_ents,
                         const raftq_wire_msg_t* msgs, uint64_t n_msgs, const raftq_wire_ent_t* ents, uint64_t n_ents, const void* pool,
                         uint64_t pool_bytes, void* out, uint64_t cap, uint64_t* frame_off, raftq_wire_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (n_props == 0) return raftq_wire_encode(h, msgs, n_msgs, ents, n_ents, pool, pool_bytes, out, cap, frame_off, counts);
  if (!props || !prop_ents || n_prop_ents == 0 || (n_msgs && !msgs) || (n_ents && !ents) || (pool_bytes && !pool) || !out || cap == 0)
    return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: null argument");
  if (h->N < 2) return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: a single-peer group commits what it appends");
  std::vector<uint8_t> seen(h->G, 0);
  for (uint64_t i = 0; i < n_props; ++i) {
    const raftq_prop_t& p = props[i];
    bool bad = p.group >= h->G || p.n_ents == 0 || p.n_ents > 1024 || (uint64_t)p.ent_first + p.n_ents > n_prop_ents;
    if (!bad) bad = h->role[p.group] != RAFTQ_ROLE_LEADER || seen[p.group]++;
    if (bad) return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: a proposal names a group this node does not lead (or twice, or no entries) -- nothing was appended");
  }
  for (uint64_t k = 0; k < n_prop_ents; ++k) {  // every entry record, named or not
    const raftq_prop_ent_t& e = prop_ents[k];
    if (e.data_len != 0 && (e.data_off > pool_bytes || e.data_len > pool_bytes - e.data_off))
      return fail(h, RAFTQ_EINVAL, "raftq_propose_frames: a payload outside the pool -- nothing was appended");
  }
  const uint64_t n_dev = n_props * (h->N - 1);
  std::vector<raftq_wire_msg_t> all(n_msgs + n_dev);
  std::vector<raftq_wire_ent_t> all_e(n_ents + n_prop_ents);
  if (n_msgs) memcpy(all.data(), msgs, n_msgs * sizeof(raftq_wire_msg_t));
  if (n_ents) memcpy(all_e.data(), ents, n_ents * sizeof(raftq_wire_ent_t));
  for (uint64_t i = 0; i < n_props; ++i) {
    const raftq_prop_t& p = props[i];
    const uint64_t g = p.group, old_last = h->last_index[g], old_term = h->last_term[g];
    raftq_log_delta_t d{g, old_last + p.n_ents, h->term[g], 0};
    rq_node_state_t s = h->state();
    rq_oracle_apply_log_deltas(&s, &d, 1, nullptr);
    for (uint32_t k = 0; k < p.n_ents; ++k) {
      const raftq_prop_ent_t& e = prop_ents[p.ent_first + k];
      raftq_wire_ent_t& w = all_e[n_ents + p.ent_first + k];
      memset(&w, 0, sizeof(w));
      w.term = h->term[g];
      w.index = old_last + 1 + k;
      w.data_len = e.data_len;
      w.data_off = e.data_len ? e.data_off : 0;
      w.type = e.type;
    }
    uint32_t run = 0;
    for (uint32_t to = 0; to < h->N; ++to) {
      if (to == h->self) continue;
      raftq_wire_msg_t& m = all[n_msgs + (uint64_t)run * n_props + i];
      memset(&m, 0, sizeof(m));
      m.group = g;
      m.term = h->term[g];
      m.log_term = old_term;
      m.index = old_last;
      m.commit = h->committed[g];
      m.from = h->self;
      m.type = RAFTQ_MSG_APP;
      m.to = (uint8_t)to;
      m.ent_first = (uint32_t)(n_ents + p.ent_first);
      m.n_ents = p.n_ents;
      ++run;
    }
  }
  return raftq_wire_encode(h, all.data(), all.size(), all_e.data(), all_e.size(), pool, pool_bytes, out, cap, frame_off, counts);
}
int raftq_wire_decode(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, raftq_wire_msg_t* msgs,
                      raftq_wire_ent_t* ents, uint64_t ents_cap, raftq_wire_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  if (n == 0) return RAFTQ_OK;
  if ((!stream && nbytes) || !frame_off || !msgs) return fail(h, RAFTQ_EINVAL, "raftq_wire_decode: null argument");
  if (!ents) ents_cap = 0;
  uint64_t n_ents = 0, n_bad = 0;
  rq_wire_decode((const uint8_t*)stream, nbytes, frame_off, n, msgs, ents, ents_cap, &n_ents, &n_bad);
  if (counts) {
    counts->n_msgs = n;
    counts->n_ents = n_ents;
    counts->n_malformed = n_bad;
    counts->bytes = frame_off[n];
  }
  if (ents && n_ents > ents_cap) return fail(h, RAFTQ_EINVAL, "raftq_wire_decode: more entries than ents_cap (counts->n_ents is the number needed)");
  return RAFTQ_OK;
}
// raftq_step_frames from its parts: the oracle's decoder, the node's checks as include/raftq_wire.h states them, the oracle's Step
int raftq_step_frames(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, int tail_appends,
                      raftq_wire_msg_t* msgs, raftq_wire_ent_t* ents, uint64_t ents_cap, raftq_wire_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_wire_counts_t{0, 0, 0, 0};
  h->n_out = 0;
  if (n == 0) return RAFTQ_OK;
  if ((!stream && nbytes) || !frame_off || !msgs) return fail(h, RAFTQ_EINVAL, "raftq_step_frames: null argument");
  if (!h->msg_flags) return fail(h, RAFTQ_ESTATE, "raftq_step_frames: the handle has not opted in to RAFTQ_MSGF_*");
  if (!ents) ents_cap = 0;
  uint64_t n_ents = 0, n_bad = 0;
  rq_wire_decode((const uint8_t*)stream, nbytes, frame_off, n, msgs, ents, ents_cap, &n_ents, &n_bad);
  if (counts) *counts = raftq_wire_counts_t{n, n_ents, n_bad, frame_off[n]};
  std::vector<raftq_wire_ent_t> all;  // the last entry's term is needed even where the caller's array is short
  const raftq_wire_ent_t* ev = ents;
  try {
    if (n_ents > ents_cap) {
      all.resize((size_t)n_ents);
      std::vector<raftq_wire_msg_t> tmp((size_t)n);
      uint64_t a = 0, b = 0;
      rq_wire_decode((const uint8_t*)stream, nbytes, frame_off, n, tmp.data(), all.data(), n_ents, &a, &b);
      ev = all.data();
    }
    h->outs.resize((size_t)n);
    std::vector<raftq_msg_t> rec((size_t)n);
    for (uint64_t i = 0; i < n; ++i) {
      raftq_wire_msg_t& m = msgs[i];
      const uint8_t t = m.type;
      const bool kind_ok = t == RAFTQ_MSG_PROP || t == RAFTQ_MSG_APP || t == RAFTQ_MSG_APP_RESP || t == RAFTQ_MSG_VOTE ||
                           t == RAFTQ_MSG_VOTE_RESP || t == RAFTQ_MSG_HEARTBEAT || t == RAFTQ_MSG_HEARTBEAT_RESP;
      if ((m.flags & RAFTQ_WIRE_F_MALFORMED) || !kind_ok || m.group >= h->G || m.from >= h->N || m.to != h->self) {
        m.flags |= RAFTQ_MSGF_SKIP;
      } else if (t == RAFTQ_MSG_PROP) {
        m.flags |= RAFTQ_MSGF_HOLD;
      } else if (t == RAFTQ_MSG_APP) {
        m.flags |= RAFTQ_MSGF_BARRIER | (tail_appends ? RAFTQ_MSGF_ENTRIES : 0);
        m.reject_hint = m.n_ents ? ev[m.ent_first + m.n_ents - 1].term : 0;
      }
      memcpy(&rec[i], &m, sizeof(raftq_msg_t));
      rec[i]._resv = m.n_ents;  // (the oracle reads the count where a caller's record keeps it)
    }
    rq_node_state_t s = h->state();
    rq_oracle_step_batch(&s, rec.data(), (size_t)n, h->outs.data());
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_step_frames: host allocation failed");
  }
  h->n_out = n;
  return RAFTQ_OK;
}
int raftq_wire_scan_frames(const void* buf, uint64_t nbytes, int big_endian, uint64_t* off, uint64_t cap, uint64_t* n_frames,
                           uint64_t* consumed) {
  if ((!buf && nbytes) || !off || !n_frames || !consumed) return fail(nullptr, RAFTQ_EINVAL, "raftq_wire_scan_frames: null argument");
  return rq_wire_scan_frames((const uint8_t*)buf, nbytes, big_endian, off, cap, n_frames, consumed) == 0 ? RAFTQ_OK : RAFTQ_EINVAL;
}
int raftq_wal_encode(raftq_t* h, const raftq_wal_rec_t* recs, uint64_t n, const void* pool, uint64_t pool_bytes, uint32_t prev_crc, void* out,
                     uint64_t cap, uint64_t* frame_off, raftq_wal_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_wal_counts_t{0, 0, 0, prev_crc, 0};
  if (n == 0) {
    if (frame_off) frame_off[0] = 0;
    return RAFTQ_OK;
  }
  if (!recs || (pool_bytes && !pool) || (cap && !out)) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: null argument");
  for (uint64_t i = 0; i < n; ++i) {
    const raftq_wal_rec_t& r = recs[i];
    const bool payload = (r.kind == RAFTQ_WAL_ENTRY || r.kind == RAFTQ_WAL_METADATA) && r.data_len != 0;
    if (r.kind < 1 || r.kind > 5 || (payload && (r.data_off > pool_bytes || r.data_len > pool_bytes - r.data_off)))
      return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: a record has an unknown kind or a payload outside the pool; nothing was written");
  }
  uint32_t last = prev_crc;
  const uint64_t need = rq_wal_encode(recs, n, (const uint8_t*)pool, prev_crc, nullptr, 0, nullptr, &last);
  if (counts) {
    counts->n_recs = n;
    counts->n_valid = n;
    counts->bytes = need;
    counts->last_crc = last;
  }
  if (need > cap) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode: out is too small (counts->bytes is the size needed)");
  rq_wal_encode(recs, n, (const uint8_t*)pool, prev_crc, (uint8_t*)out, cap, frame_off, &last);
  return RAFTQ_OK;
}
// _begin .. _end: here the encode simply happens at _begin and _end hands over what it said
int raftq_wal_encode_begin(raftq_t* h, const raftq_wal_rec_t* recs, uint64_t n, const void* pool, uint64_t pool_bytes, uint32_t prev_crc, void* out,
                           uint64_t cap, uint64_t* frame_off) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (h->wal_pending) return fail(h, RAFTQ_ESTATE, "raftq_wal_encode_begin: the previous one has not been ended");
  if (n == 0 || !recs || !out || cap == 0) return fail(h, RAFTQ_EINVAL, "raftq_wal_encode_begin: null argument or empty batch");
  h->wal_rc = raftq_wal_encode(h, recs, n, pool, pool_bytes, prev_crc, out, cap, frame_off, &h->wal_counts);
  h->wal_err = h->err;
  h->wal_pending = true;
  return RAFTQ_OK;
}
int raftq_wal_encode_end(raftq_t* h, raftq_wal_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!h->wal_pending) return fail(h, RAFTQ_ESTATE, "raftq_wal_encode_end: nothing was begun");
  h->wal_pending = false;
  if (counts) *counts = h->wal_counts;
  if (h->wal_rc != RAFTQ_OK) return fail(h, h->wal_rc, h->wal_err);
  return RAFTQ_OK;
}
int raftq_wal_decode(raftq_t* h, const void* bytes, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, uint32_t prev_crc,
                     raftq_wal_rec_t* recs, raftq_wal_counts_t* counts) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (counts) *counts = raftq_wal_counts_t{0, 0, 0, prev_crc, 0};
  if (n == 0) return RAFTQ_OK;
  if ((!bytes && nbytes) || !frame_off || !recs) return fail(h, RAFTQ_EINVAL, "raftq_wal_decode: null argument");
  uint64_t n_valid = 0;
  uint32_t last = prev_crc;
  rq_wal_decode((const uint8_t*)bytes, nbytes, frame_off, n, prev_crc, recs, &n_valid, &last);
  if (counts) {
    counts->n_recs = n;
    counts->n_valid = n_valid;
    counts->bytes = frame_off[n];
    counts->last_crc = last;
  }
  return RAFTQ_OK;
}

}  // extern "C"
