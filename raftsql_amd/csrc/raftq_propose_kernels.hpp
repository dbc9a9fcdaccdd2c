// raftq_propose_kernels.hpp -- device code of raftq_propose_frames (include/raftq_wire.h): the leader's appendEntry and
// bcastAppend for a batch of proposing groups (raft.go:211-215 -> etcd raft.stepLeader MsgProp: `r.appendEntry(m.Entries...);
// r.bcastAppend()`), one lane per group, written straight into the input of the streaming encoder that runs behind it.
//
// Round 5 did this on the host, per proposal: handle_proposal + bcast_append (raftq_node.cpp) built N - 1 64-byte message
// records and an entry header per proposal, copied them lane by lane into a page-locked array, and the encoder's readers pulled
// them back over the link -- 2.9 ms of a 4.0 ms turn on one core for 32K groups (profiles/r05/one_node_phases.txt).  The records
// are a pure function of the group's device-resident state (Term, lastIndex, lastTerm, committed) and of which entries were
// proposed: nothing the host has to compute, and nothing that has to cross the link.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raftq_step_kernels.hpp"
#include "raftq_wire_parse.hpp"

namespace raftqk {

struct PropRec {  // == raftq_prop_t
  uint64_t group;
  uint32_t ent_first, n_ents;
};
struct PropEnt {  // == raftq_prop_ent_t
  uint64_t data_off;
  uint32_t data_len, type;
};
static_assert(sizeof(PropRec) == 16 && sizeof(PropEnt) == 16, "record layout");

constexpr uint32_t kPropPayload = 1, kPropGroup = 2, kPropCount = 3, kPropRange = 4, kPropNotLeader = 5, kPropTwice = 6;
constexpr uint32_t kPropMaxEnts = 1024;  // raft.Config.MaxSizePerMsg's share of entries (raft.go:157): more go out as the caller's own sends

// Nothing is applied unless every record is sound: a group of this handle, led by this node, named once (the group's list
// count word doubles as the "seen" mark: the apply kernel hands it back zero), 1 .. kPropMaxEnts entries inside prop_ents[],
// every entry's payload inside the pool.
// The records lie in page-locked HOST memory: this kernel reads them there ONCE, coalesced (16 bytes a lane: 1 KB a wave
// instruction over the link), and leaves a copy in device scratch for the kernel behind it -- round 6's first form had both
// kernels read the host arrays: 33 + 36 us for 32K groups, all of it link latency (profiles/r06/node_kernel_stats_first.csv).
static __global__ __launch_bounds__(kBlock) void propose_check_kernel(NodeArrays a, const PropRec* __restrict__ props, uint64_t n,
                                                                      const PropEnt* __restrict__ pe, uint64_t n_pe, uint64_t pool_bytes,
                                                                      unsigned int* bad, unsigned int stamp, PropRec* __restrict__ props_d,
                                                                      PropEnt* __restrict__ pe_d, unsigned long long* why) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  // the entry records, one per lane of the first ceil(n_pe / kBlock) workgroups ... (the grid covers max(n, n_pe) lanes)
  uint32_t reason = 0;  // kProp*: what is wrong with record i (the largest (reason, record) of the call goes into *why for the error text)
  if (i < n_pe) {
    const PropEnt e = pe[i];
    pe_d[i] = e;
    if (e.data_len != 0 && (e.data_off > pool_bytes || e.data_len > pool_bytes - e.data_off)) reason = kPropPayload;  // (an entry no record names is checked too: harmless)
  }
  if (i < n) {
    const PropRec p = props[i];
    props_d[i] = p;
    if (p.group >= a.n_groups) reason = kPropGroup;
    else if (p.n_ents == 0 || p.n_ents > kPropMaxEnts) reason = kPropCount;
    else if ((uint64_t)p.ent_first + p.n_ents > n_pe) reason = kPropRange;
    else if (a.role[p.group] != kLeader) reason = kPropNotLeader;
    else if (atomicAdd(&a.rec[p.group].lst_cnt, 1u) != 0) reason = kPropTwice;
  }
  if (reason) atomicMax(why, ((unsigned long long)(stamp & 0xffffffu) << 40) | ((unsigned long long)reason << 32) | (uint32_t)i);
  if (__ballot(reason != 0) != 0 && (threadIdx.x & 63) == 0) atomicExch(bad, stamp);  // (the word holds this call's stamp: refused)
}

// props / pe: the check kernel's copies in device memory.  msgs_out: the device part of the encoder's message array -- (N - 1) runs of n records, run r = the MsgApps for the r-th peer
// slot other than this node's; ents_out: the device part of its entry-header array, whose first element is entry `ent_base` of
// the whole array (a message's ent_first counts from the array's start).
static __global__ __launch_bounds__(kBlock) void propose_apply_kernel(NodeArrays a, const PropRec* __restrict__ props, uint64_t n,
                                                                      const PropEnt* __restrict__ pe, const unsigned int* __restrict__ bad,
                                                                      unsigned int stamp, WireMsg* __restrict__ msgs_out, WireEnt* __restrict__ ents_out, uint32_t ent_base) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const PropRec p = props[i];
  if (*bad == stamp) {  // refused: the marks of the check come off, nothing else happens
    if (p.group < a.n_groups) a.rec[p.group].lst_cnt = 0;
    return;
  }
  Node node(a, p.group);
  const uint64_t old_last = node.last_index, old_term = node.last_term;
  // appendEntry: `es[i].Term = r.Term; es[i].Index = li + 1 + i; r.raftLog.append(es...); r.prs[r.id].maybeUpdate(lastIndex)`.
  // (`r.maybeCommit()` cannot move anything with more than one peer: the leader's own Match is the largest, the quorum-th
  // largest is somebody else's and did not change -- the host wrapper refuses a single-peer handle.)
  node.last_index = old_last + p.n_ents;
  node.last_term = node.term;
  if (node.match(a.self) < node.last_index) node.set_match(a.self, node.last_index);
  node.lst_cnt = 0;
  node.store();  // (list words back to empty: the check's mark with them)
  for (uint32_t k = 0; k < p.n_ents; ++k) {
    const PropEnt e = pe[p.ent_first + k];
    WireEnt w;
    w.term = node.term;
    w.index = old_last + 1 + k;
    w.data_len = e.data_len;
    w.data_off = e.data_len ? e.data_off : 0;
    w.type = e.type;
    ents_out[p.ent_first + k] = w;
  }
  // bcastAppend -> sendAppend(to) with Progress.Next at the tail: MsgApp{Index: next - 1, LogTerm: term(next - 1), Entries, Commit}
  WireMsg m;
  m.group = p.group;
  m.term = node.term;
  m.log_term = old_term;
  m.index = old_last;
  m.commit = node.committed;
  m.reject_hint = 0;
  m.from = a.self;
  m.type = kMsgApp;
  m.reject = 0;
  m.flags = 0;
  m.ent_first = ent_base + p.ent_first;
  m.n_ents = p.n_ents;
  uint32_t run = 0;
  for (uint32_t to = 0; to < a.n_peers; ++to) {
    if (to == a.self) continue;
    m.to = (uint8_t)to;
    msgs_out[(uint64_t)run * n + i] = m;
    ++run;
  }
}

}  // namespace raftqk
