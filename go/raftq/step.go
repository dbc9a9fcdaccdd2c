// step.go -- cgo binding of include/raftq_step.h: the batched raft Step.
//
// SOURCE ONLY: never compiled or run (no Go toolchain in the build image; see README.md).
// It is the binding INTEGRATION.md section 1b describes.
package raftq

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raftsql_amd -lraftq -Wl,-rpath,${SRCDIR}/../../raftsql_amd
#include "raftq_step.h"
*/
import "C"

import "unsafe"

// raftpb.MessageType values Step accepts.
const (
	MsgHup           = C.RAFTQ_MSG_HUP
	MsgBeat          = C.RAFTQ_MSG_BEAT
	MsgApp           = C.RAFTQ_MSG_APP // header only: entries stay with the log's owner
	MsgAppResp       = C.RAFTQ_MSG_APP_RESP
	MsgVote          = C.RAFTQ_MSG_VOTE
	MsgVoteResp      = C.RAFTQ_MSG_VOTE_RESP
	MsgHeartbeat     = C.RAFTQ_MSG_HEARTBEAT
	MsgHeartbeatResp = C.RAFTQ_MSG_HEARTBEAT_RESP
)

// Msg is layout-identical to raftq_msg_t (64 bytes).  WireTo / Flags / Resv are padding -- ignored by the library --
// unless the engine opted in with SetMsgFlags(true); a caller that opts in sets them on EVERY record (the staging
// memory StepStage hands out is not zeroed).
type Msg struct {
	Group, Term, LogTerm, Index, Commit, RejectHint uint64
	From                                            uint32 // peer slot = raft ID - 1
	Type, Reject                                    uint8
	WireTo, Flags                                   uint8  // raftq_msg_t._pad: [1] = MsgfEntries | MsgfBarrier
	Resv                                            uint64 // under MsgfEntries: low 32 bits = number of entries
}

const (
	MsgfEntries = C.RAFTQ_MSGF_ENTRIES // on a MsgApp: Resv = entry count, RejectHint = the last entry's term
	MsgfBarrier = C.RAFTQ_MSGF_BARRIER // on a MsgApp: what follows it in its group waits if it is left to the log's owner
	MsgfHold    = C.RAFTQ_MSGF_HOLD    // not stepped (a MsgProp: the log owner's), answered OutHeld; the rest of its group is deferred
	MsgfSkip    = C.RAFTQ_MSGF_SKIP    // nobody's: no field is looked at, answered OutSkipped
)

// SetMsgFlags opts the engine in to (or out of) Msg.Flags / Msg.Resv; no batch may be in flight.
func (e *Engine) SetMsgFlags(on bool) error {
	v := C.int(0)
	if on {
		v = 1
	}
	return e.err(C.raftq_step_set_msg_flags(e.h, v))
}

// StepOut is layout-identical to raftq_step_out_t (64 bytes).
type StepOut struct {
	Group, Term, Index, LogTerm, Commit, LastIndex uint64
	To, Vote, Lead                                 uint32
	Type, Reject, Flags, Role                      uint8
}

const (
	OutNone           = C.RAFTQ_OUT_NONE
	OutVoteResp       = C.RAFTQ_OUT_VOTE_RESP
	OutHeartbeatResp  = C.RAFTQ_OUT_HEARTBEAT_RESP
	OutCampaign       = C.RAFTQ_OUT_CAMPAIGN
	OutBecameLeader   = C.RAFTQ_OUT_BECAME_LEADER
	OutProgress       = C.RAFTQ_OUT_PROGRESS
	OutBcastHeartbeat = C.RAFTQ_OUT_BCAST_HEARTBEAT
	OutAppend         = C.RAFTQ_OUT_APPEND
	OutDeferred       = C.RAFTQ_OUT_DEFERRED // not applied: behind a MsgApp (MsgfBarrier) that was left to the log's owner
	OutAppended       = C.RAFTQ_OUT_APPENDED // MsgApp flagged MsgfEntries that appended at the tail: store the entries, ack Index
	OutSkipped        = C.RAFTQ_OUT_SKIPPED  // MsgfSkip
	OutHeld           = C.RAFTQ_OUT_HELD     // MsgfHold

	FlagHardState   = C.RAFTQ_OUTF_HARDSTATE // persist {Term, Vote, Commit} before sending (raft.go:228-230)
	FlagCommitted   = C.RAFTQ_OUTF_COMMITTED
	FlagUpdated     = C.RAFTQ_OUTF_UPDATED
	FlagSteppedDown = C.RAFTQ_OUTF_STEPPED_DOWN
)

// StepOutC is layout-identical to raftq_step_out_c_t (40 bytes): the result record without what the caller's
// own batch already says (group, addressee); Aux = LogTerm for OutCampaign / OutBecameLeader, LastIndex otherwise.
type StepOutC struct {
	Term, Index, Commit, Aux            uint64
	Vote, Lead, Type, Reject, Flags, Role uint8
	_                                   [2]uint8
}

// SetCompact switches the result format (no batch in flight); StepResultsC reads the records of the batch
// collected last in place (valid until the next submit).
func (e *Engine) SetCompact(on bool) error {
	v := C.int(0)
	if on {
		v = 1
	}
	return e.err(C.raftq_step_set_compact(e.h, v))
}

func (e *Engine) StepResultsC() ([]StepOutC, error) {
	var p *C.raftq_step_out_c_t
	var n C.uint64_t
	if rc := C.raftq_step_results_c(e.h, &p, &n); rc != C.RAFTQ_OK {
		return nil, e.err(rc)
	}
	return unsafe.Slice((*StepOutC)(unsafe.Pointer(p)), int(n)), nil
}

// LogDelta is layout-identical to raftq_log_delta_t.
type LogDelta struct{ Group, LastIndex, LastTerm, CommitTo uint64 }

// SetSelf: which peer slot this process is in every group (raft.Config.ID - 1, raft.go:153).
// StepOutS is layout-identical to raftq_step_out_s_t (32 bytes, round 6): StepOutC without Aux.  raftLog.lastIndex() after a
// message is the log owner's own knowledge, a new leader's LogTerm is its Term, and a campaign -- which moves no commit index --
// carries its LogTerm in Commit (include/raftq_step.h).  SetCompactFormat(2) selects it.  SOURCE ONLY.
type StepOutS struct {
	Term, Index, Commit                   uint64
	Vote, Lead, Type, Reject, Flags, Role uint8
	_                                     [2]uint8
}

// SetCompactFormat picks the result record: 0 = 64 bytes, 1 = 40 (StepOutC), 2 = 32 (StepOutS); no batch may be in flight.
func (e *Engine) SetCompactFormat(f int) error { return e.err(C.raftq_step_set_compact(e.h, C.int(f))) }

// StepResultsS is the last collected batch's results in the 32-byte format, in place (valid until the next submit).
func (e *Engine) StepResultsS() ([]StepOutS, error) {
	var p *C.raftq_step_out_s_t
	var n C.uint64_t
	if rc := C.raftq_step_results_s(e.h, &p, &n); rc != C.RAFTQ_OK {
		return nil, e.err(rc)
	}
	return unsafe.Slice((*StepOutS)(unsafe.Pointer(p)), int(n)), nil
}

func (e *Engine) SetSelf(slot uint32) error { return e.err(C.raftq_set_self(e.h, C.uint32_t(slot))) }

// StepBatch is rc.node.Step (raft.go:268-270) for every message; out[i] answers msgs[i].
// Messages of one group are applied in slice order.
func (e *Engine) StepBatch(msgs []Msg, out []StepOut) (groupsTouched uint64, err error) {
	if len(msgs) == 0 {
		return 0, nil
	}
	var c C.raftq_step_counts_t
	var po *C.raftq_step_out_t
	if len(out) >= len(msgs) {
		po = (*C.raftq_step_out_t)(unsafe.Pointer(&out[0]))
	}
	rc := C.raftq_step_batch(e.h, (*C.raftq_msg_t)(unsafe.Pointer(&msgs[0])), C.uint64_t(len(msgs)), po, &c)
	return uint64(c.n_groups_touched), e.err(rc)
}

// StepStage returns the pinned staging slice for n messages: fill it in place (e.g. straight
// from the rafthttp receive path) and hand the same slice to StepBatch -- no host copy is made.
func (e *Engine) StepStage(n int) ([]Msg, error) {
	var p *C.raftq_msg_t
	if rc := C.raftq_step_stage(e.h, C.uint64_t(n), &p); rc != C.RAFTQ_OK {
		return nil, e.err(rc)
	}
	return unsafe.Slice((*Msg)(unsafe.Pointer(p)), n), nil
}

// ApplyLogDeltas reports new log tails (appendEntry on a leader, maybeAppend on a follower).
// committed (len(d) or nil) receives raftLog.committed after each record.
func (e *Engine) ApplyLogDeltas(d []LogDelta, committed []uint64) error {
	if len(d) == 0 {
		return nil
	}
	var pc *C.uint64_t
	if len(committed) >= len(d) {
		pc = (*C.uint64_t)(unsafe.Pointer(&committed[0]))
	}
	return e.err(C.raftq_apply_log_deltas(e.h, (*C.raftq_log_delta_t)(unsafe.Pointer(&d[0])), C.uint64_t(len(d)), pc))
}

// ApplyLogDeltasNowait enqueues the same reports and returns (raftq_apply_log_deltas_nowait): nothing comes back -- for a
// LEADER's appendEntry with more than one peer, which cannot move raftLog.committed.
func (e *Engine) ApplyLogDeltasNowait(d []LogDelta) error {
	if len(d) == 0 {
		return nil
	}
	return e.err(C.raftq_apply_log_deltas_nowait(e.h, (*C.raftq_log_delta_t)(unsafe.Pointer(&d[0])), C.uint64_t(len(d))))
}

// LoadRoles sets role[g] (0 follower, 1 candidate, 2 leader) and, optionally, the election clocks.
func (e *Engine) LoadRoles(role []uint8, elapsed []uint32) error {
	var pe *C.uint32_t
	if elapsed != nil {
		pe = (*C.uint32_t)(unsafe.Pointer(&elapsed[0]))
	}
	return e.err(C.raftq_load_roles(e.h, (*C.uint8_t)(unsafe.Pointer(&role[0])), pe))
}

// LoadNode bulk-loads the per-group node scalars Step works on (raftq_load_node); lead may be nil.
func (e *Engine) LoadNode(term []uint64, vote, lead []uint32, lastIndex, lastTerm []uint64) error {
	var pl *C.uint32_t
	if lead != nil {
		pl = (*C.uint32_t)(unsafe.Pointer(&lead[0]))
	}
	return e.err(C.raftq_load_node(e.h, (*C.uint64_t)(unsafe.Pointer(&term[0])), (*C.uint32_t)(unsafe.Pointer(&vote[0])), pl,
		(*C.uint64_t)(unsafe.Pointer(&lastIndex[0])), (*C.uint64_t)(unsafe.Pointer(&lastTerm[0]))))
}

// NodeState is a host copy of everything Step keeps per group (tests, snapshots).
type NodeState struct {
	Term, LastIndex, LastTerm, FirstIdx, Committed []uint64
	Vote, Lead, Elapsed                            []uint32
	Role                                           []uint8
	Match                                          []uint64 // [p*G + g]
}

// ReadNode copies the node state back (raftq_read_node + raftq_read_tick + raftq_read_committed + raftq_read_match).
func (e *Engine) ReadNode() (*NodeState, error) {
	g := int(e.Groups)
	s := &NodeState{Term: make([]uint64, g), LastIndex: make([]uint64, g), LastTerm: make([]uint64, g), FirstIdx: make([]uint64, g),
		Committed: make([]uint64, g), Vote: make([]uint32, g), Lead: make([]uint32, g), Elapsed: make([]uint32, g),
		Role: make([]uint8, g), Match: make([]uint64, g*int(e.Peers))}
	if err := e.err(C.raftq_read_node(e.h, (*C.uint64_t)(unsafe.Pointer(&s.Term[0])), (*C.uint32_t)(unsafe.Pointer(&s.Vote[0])),
		(*C.uint32_t)(unsafe.Pointer(&s.Lead[0])), (*C.uint64_t)(unsafe.Pointer(&s.LastIndex[0])),
		(*C.uint64_t)(unsafe.Pointer(&s.LastTerm[0])), (*C.uint64_t)(unsafe.Pointer(&s.FirstIdx[0])))); err != nil {
		return nil, err
	}
	if err := e.err(C.raftq_read_tick(e.h, nil, (*C.uint32_t)(unsafe.Pointer(&s.Elapsed[0])), (*C.uint8_t)(unsafe.Pointer(&s.Role[0])))); err != nil {
		return nil, err
	}
	if err := e.ReadCommitted(s.Committed); err != nil {
		return nil, err
	}
	return s, e.err(C.raftq_read_match(e.h, (*C.uint64_t)(unsafe.Pointer(&s.Match[0]))))
}

// Msg40 is layout-identical to raftq_msg40_t: the 40-byte inbound record.  Aux is m.RejectHint on MsgAppResp and
// m.LogTerm on every other kind (no kind Step takes carries both).
type Msg40 struct {
	Group  uint32
	From   uint8
	Type   uint8
	Reject uint8
	_      uint8
	Term   uint64
	Index  uint64
	Aux    uint64
	Commit uint64
}

// StepStagePacked returns the staging slice the next SubmitPacked will use, with room for n packed messages.
func (e *Engine) StepStagePacked(n int) ([]Msg40, error) {
	var p *C.raftq_msg40_t
	if rc := C.raftq_step_stage_packed(e.h, C.uint64_t(n), &p); rc != C.RAFTQ_OK {
		return nil, e.err(rc)
	}
	return unsafe.Slice((*Msg40)(unsafe.Pointer(p)), n), nil
}

// StepSubmitPacked enqueues a batch of packed records (at most three batches in flight); collect it like any other.
func (e *Engine) StepSubmitPacked(msgs []Msg40) error {
	if len(msgs) == 0 {
		return nil
	}
	return e.err(C.raftq_step_submit_packed(e.h, (*C.raftq_msg40_t)(unsafe.Pointer(&msgs[0])), C.uint64_t(len(msgs))))
}
