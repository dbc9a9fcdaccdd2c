// engine_more.go -- the rest of include/raftq.h and the pipelined half of include/raftq_step.h: every engine-level
// export has a binding (tests/test_go_binding.py checks names, arities and argument kinds against the headers).
//
// SOURCE ONLY: never compiled or run (no Go toolchain in the build image; see README.md).
package raftq

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raftsql_amd -lraftq -Wl,-rpath,${SRCDIR}/../../raftsql_amd
#include <stdlib.h>
#include "raftq_step.h"
*/
import "C"

import "unsafe"

// ABIVersion is RAFTQ_ABI_VERSION of the loaded library.
func ABIVersion() int { return int(C.raftq_abi_version()) }

// Quorum is etcd's raft.q(): floor(n/2)+1.
func Quorum(peers uint32) uint32 { return uint32(C.raftq_quorum(C.uint32_t(peers))) }

// NGroups / NPeers read the handle's shape back.
func (e *Engine) NGroups() uint64 { return uint64(C.raftq_groups(e.h)) }
func (e *Engine) NPeers() uint32  { return uint32(C.raftq_peers(e.h)) }

// SetStream installs a caller-owned hipStream_t; Stream returns the one in use.
func (e *Engine) SetStream(stream unsafe.Pointer) error { return e.err(C.raftq_set_stream(e.h, stream)) }
func (e *Engine) Stream() unsafe.Pointer                { return C.raftq_get_stream(e.h) }

// TermDelta is layout-identical to raftq_term_delta_t.
type TermDelta struct{ Group, CurTerm, FirstIdxCurTerm uint64 }

// ApplyTermDeltas updates the current-term gate of the groups named (raftLog.maybeCommit's term test).
func (e *Engine) ApplyTermDeltas(d []TermDelta) error {
	if len(d) == 0 {
		return nil
	}
	return e.err(C.raftq_apply_term_deltas(e.h, (*C.raftq_term_delta_t)(unsafe.Pointer(&d[0])), C.uint64_t(len(d))))
}

// CommitAdvance is the synchronous commit sweep of SURVEY 8b; committedOut may be nil.
func (e *Engine) CommitAdvance(gated bool, committedOut []uint64) (changed uint64, err error) {
	var p *C.uint64_t
	if committedOut != nil {
		p = (*C.uint64_t)(unsafe.Pointer(&committedOut[0]))
	}
	g := C.int(0)
	if gated {
		g = 1
	}
	var n C.uint64_t
	err = e.err(C.raftq_commit_advance(e.h, g, p, &n))
	return uint64(n), err
}

// VoteTally is the synchronous RequestVote tally; outcomeOut may be nil.
func (e *Engine) VoteTally(outcomeOut []uint8) (Counts, error) {
	var p *C.uint8_t
	if outcomeOut != nil {
		p = (*C.uint8_t)(unsafe.Pointer(&outcomeOut[0]))
	}
	var c C.raftq_counts_t
	if err := e.err(C.raftq_vote_tally(e.h, p, &c)); err != nil {
		return Counts{}, err
	}
	return Counts{uint64(c.n_changed), uint64(c.n_won), uint64(c.n_lost)}, nil
}

// ReadVotes copies votes[p*G+g] back.
func (e *Engine) ReadVotes(out []uint8) error {
	return e.err(C.raftq_read_votes(e.h, (*C.uint8_t)(unsafe.Pointer(&out[0]))))
}

// SetTimers: ElectionTick / HeartbeatTick of raft.Config (raft.go:154-155) and the timeout jitter's seed.
func (e *Engine) SetTimers(electionTick, heartbeatTick uint32, seed uint64) error {
	return e.err(C.raftq_set_timers(e.h, C.uint32_t(electionTick), C.uint32_t(heartbeatTick), C.uint64_t(seed)))
}

// TickCounts is layout-identical to raftq_tick_counts_t.
type TickCounts struct{ Hup, Beat uint64 }

// Tick is rc.node.Tick() (raft.go:223-224) for every group; wantCounts makes it synchronous.
func (e *Engine) Tick(wantCounts bool) (TickCounts, error) {
	var c C.raftq_tick_counts_t
	if !wantCounts {
		return TickCounts{}, e.err(C.raftq_tick(e.h, nil))
	}
	err := e.err(C.raftq_tick(e.h, &c))
	return TickCounts{uint64(c.n_hup), uint64(c.n_beat)}, err
}

// CollectHups lists the groups the last Tick sent MsgHup to, ascending.
func (e *Engine) CollectHups(out []uint64) (uint64, error) {
	var n C.uint64_t
	var p *C.uint64_t
	if len(out) > 0 {
		p = (*C.uint64_t)(unsafe.Pointer(&out[0]))
	}
	rc := C.raftq_collect_hups(e.h, p, C.uint64_t(len(out)), &n)
	return uint64(n), e.err(rc)
}

// TickCollect is one Tick and both of its lists in one call (raftq_tick_collect: two launches, one wait).
func (e *Engine) TickCollect(hups, beats []uint64) (nHup, nBeat uint64, err error) {
	var hp, bp *C.uint64_t
	if len(hups) > 0 {
		hp = (*C.uint64_t)(unsafe.Pointer(&hups[0]))
	}
	if len(beats) > 0 {
		bp = (*C.uint64_t)(unsafe.Pointer(&beats[0]))
	}
	var nh, nb C.uint64_t
	err = e.err(C.raftq_tick_collect(e.h, hp, C.uint64_t(len(hups)), &nh, bp, C.uint64_t(len(beats)), &nb))
	return uint64(nh), uint64(nb), err
}

// TickBeatBitmap asks TickCollectLists for the MsgBeat groups as a bitmap in group order instead of a list.
const TickBeatBitmap = uint(C.RAFTQ_TICK_BEAT_BITMAP)

// TickLists is what TickCollectLists left in the library's page-locked memory: valid until the next call on the engine.
type TickLists struct {
	Hups       []uint32 // ascending
	Beats      []uint32 // ascending; nil with TickBeatBitmap
	BeatBitmap []uint64 // bit g%64 of word g/64; nil without TickBeatBitmap
	NHup       uint64   // totals (the lists hold at most hupCap / beatCap of them)
	NBeat      uint64
}

// TickCollectLists is rc.node.Tick() (raft.go:223-224) of every group and its two lists, left in place: 4-byte group ids in
// page-locked memory, nothing copied (raftq_tick_collect_lists + raftq_last_tick_lists: three launches, one wait).
func (e *Engine) TickCollectLists(flags uint, hupCap, beatCap uint64) (TickLists, error) {
	var nh, nb C.uint64_t
	if err := e.err(C.raftq_tick_collect_lists(e.h, C.uint(flags), C.uint64_t(hupCap), C.uint64_t(beatCap), &nh, &nb)); err != nil {
		return TickLists{}, err
	}
	var ph, pb *C.uint32_t
	var pm *C.uint64_t
	var lh, lb, lm C.uint64_t
	if err := e.err(C.raftq_last_tick_lists(e.h, &ph, &lh, &pb, &lb, &pm, &lm)); err != nil {
		return TickLists{}, err
	}
	t := TickLists{NHup: uint64(nh), NBeat: uint64(nb)}
	if lh > 0 {
		t.Hups = unsafe.Slice((*uint32)(unsafe.Pointer(ph)), int(lh))
	}
	if lb > 0 {
		t.Beats = unsafe.Slice((*uint32)(unsafe.Pointer(pb)), int(lb))
	}
	if lm > 0 {
		t.BeatBitmap = unsafe.Slice((*uint64)(unsafe.Pointer(pm)), int(lm))
	}
	return t, nil
}

// Campaign is becomeCandidate for the groups named (distinct).
func (e *Engine) Campaign(groups []uint64, selfPeer uint32) error {
	if len(groups) == 0 {
		return nil
	}
	return e.err(C.raftq_campaign(e.h, (*C.uint64_t)(unsafe.Pointer(&groups[0])), C.uint64_t(len(groups)), C.uint32_t(selfPeer)))
}

// Cycle is one batching turn (raft.go:227-235 for every group at once): acks in, sweep, advance list out, one wait.
func (e *Engine) Cycle(deltas []Delta, votes []VoteDelta, flags uint, out []Advance, wantCounts bool) (advanced uint64, c Counts, err error) {
	var dp *C.raftq_delta_t
	var vp *C.raftq_vote_delta_t
	var op *C.raftq_advance_t
	if len(deltas) > 0 {
		dp = (*C.raftq_delta_t)(unsafe.Pointer(&deltas[0]))
	}
	if len(votes) > 0 {
		vp = (*C.raftq_vote_delta_t)(unsafe.Pointer(&votes[0]))
	}
	if len(out) > 0 {
		op = (*C.raftq_advance_t)(unsafe.Pointer(&out[0]))
	}
	var n C.uint64_t
	var cc C.raftq_counts_t
	var cp *C.raftq_counts_t
	if wantCounts {
		cp = &cc
	}
	err = e.err(C.raftq_cycle(e.h, dp, C.uint64_t(len(deltas)), vp, C.uint64_t(len(votes)), C.uint(flags), op, C.uint64_t(len(out)), &n, cp))
	return uint64(n), Counts{uint64(cc.n_changed), uint64(cc.n_won), uint64(cc.n_lost)}, err
}

// Stage returns the handle's ack buffer (device memory behind a large BAR, pinned host memory otherwise) as slices the
// message handlers fill in place; pass the same slices to Cycle.  The memory belongs to the library: WRITE-ONLY.
func (e *Engine) Stage(nDeltas, nVotes int) ([]Delta, []VoteDelta, error) {
	var dp *C.raftq_delta_t
	var vp *C.raftq_vote_delta_t
	if err := e.err(C.raftq_stage(e.h, C.uint64_t(nDeltas), C.uint64_t(nVotes), &dp, &vp)); err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*Delta)(unsafe.Pointer(dp)), nDeltas), unsafe.Slice((*VoteDelta)(unsafe.Pointer(vp)), nVotes), nil
}

// LastAdvances reads the advance list of the last Cycle in place (pinned memory, valid until the next turn).
func (e *Engine) LastAdvances() ([]Advance, error) {
	var p *C.raftq_advance_t
	var n C.uint64_t
	if err := e.err(C.raftq_last_advances(e.h, &p, &n)); err != nil {
		return nil, err
	}
	return unsafe.Slice((*Advance)(unsafe.Pointer(p)), int(n)), nil
}

// Delta16 / Advance16 are layout-identical to raftq_delta16_t / raftq_advance16_t: the 16-byte records of the packed turn.
type Delta16 struct {
	Match       uint64
	Group, Peer uint32
}
type Advance16 struct {
	NewCommit         uint64
	Group, AdvancedBy uint32
}

// CyclePacked is Cycle with the 16-byte records (handles of at most 2^32 groups).
func (e *Engine) CyclePacked(deltas []Delta16, votes []VoteDelta, flags uint, out []Advance16, wantCounts bool) (advanced uint64, c Counts, err error) {
	var dp *C.raftq_delta16_t
	var vp *C.raftq_vote_delta_t
	var op *C.raftq_advance16_t
	if len(deltas) > 0 {
		dp = (*C.raftq_delta16_t)(unsafe.Pointer(&deltas[0]))
	}
	if len(votes) > 0 {
		vp = (*C.raftq_vote_delta_t)(unsafe.Pointer(&votes[0]))
	}
	if len(out) > 0 {
		op = (*C.raftq_advance16_t)(unsafe.Pointer(&out[0]))
	}
	var n C.uint64_t
	var cc C.raftq_counts_t
	var cp *C.raftq_counts_t
	if wantCounts {
		cp = &cc
	}
	err = e.err(C.raftq_cycle_packed(e.h, dp, C.uint64_t(len(deltas)), vp, C.uint64_t(len(votes)), C.uint(flags), op, C.uint64_t(len(out)), &n, cp))
	return uint64(n), Counts{uint64(cc.n_changed), uint64(cc.n_won), uint64(cc.n_lost)}, err
}

func (e *Engine) StagePacked(nDeltas, nVotes int) ([]Delta16, []VoteDelta, error) {
	var dp *C.raftq_delta16_t
	var vp *C.raftq_vote_delta_t
	if err := e.err(C.raftq_stage_packed(e.h, C.uint64_t(nDeltas), C.uint64_t(nVotes), &dp, &vp)); err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*Delta16)(unsafe.Pointer(dp)), nDeltas), unsafe.Slice((*VoteDelta)(unsafe.Pointer(vp)), nVotes), nil
}

func (e *Engine) LastAdvancesPacked() ([]Advance16, error) {
	var p *C.raftq_advance16_t
	var n C.uint64_t
	if err := e.err(C.raftq_last_advances_packed(e.h, &p, &n)); err != nil {
		return nil, err
	}
	return unsafe.Slice((*Advance16)(unsafe.Pointer(p)), int(n)), nil
}

// CycleSegmented: a CyclePacked turn passed this flag (list read in place, no counts) may leave its advance list in segments --
// the sweep writes a tile's records itself and the compaction pass does not run (raftq.h RAFTQ_CYCLE_SEGMENTED).
const CycleSegmented = C.RAFTQ_CYCLE_SEGMENTED

// LastAdvanceSegments returns the last packed turn's advance list as segments: segment s holds counts[s] records at
// recs[s*stride:]; walked in order they are the ascending list.  A turn that produced the contiguous list is one segment.
// Views of the library's pinned memory, valid until the next turn.
func (e *Engine) LastAdvanceSegments() (recs []Advance16, counts []uint32, stride uint64, err error) {
	var p *C.raftq_advance16_t
	var pc *C.uint32_t
	var ns C.uint32_t
	var st C.uint64_t
	if err = e.err(C.raftq_last_advance_segments(e.h, &p, &pc, &ns, &st)); err != nil || ns == 0 {
		return nil, nil, 0, err
	}
	counts = unsafe.Slice((*uint32)(unsafe.Pointer(pc)), int(ns))
	n := uint64(st)*uint64(ns-1) + uint64(counts[ns-1])
	return unsafe.Slice((*Advance16)(unsafe.Pointer(p)), int(n)), counts, uint64(st), nil
}

// TimerBegin / TimerEnd bracket work on the handle's stream with HIP events (milliseconds).
func (e *Engine) TimerBegin() error { return e.err(C.raftq_timer_begin(e.h)) }
func (e *Engine) TimerEnd() (float32, error) {
	var ms C.float
	err := e.err(C.raftq_timer_end(e.h, &ms))
	return float32(ms), err
}

// Size / Stream / TimerBegin / TimerEnd of a sweep set.
func (s *Set) Size() uint32           { return uint32(C.raftq_set_size(s.s)) }
func (s *Set) Stream() unsafe.Pointer { return C.raftq_set_get_stream(s.s) }
func (s *Set) TimerBegin() error      { return s.err(C.raftq_set_timer_begin(s.s)) }

// Tick is one Tick of every member as ONE dispatch (raftq_set_tick); Wait before reading a member's lists.
func (s *Set) Tick() error { return s.err(C.raftq_set_tick(s.s)) }

func (s *Set) TimerEnd() (float32, error) {
	var ms C.float
	err := s.err(C.raftq_set_timer_end(s.s, &ms))
	return float32(ms), err
}

// SweepManyAsync is K launches from ONE cgo call (raftq_sweep_many_async); engines that share a stream -- members of a
// Set -- are alternated over two streams inside the library.
func SweepManyAsync(engines []*Engine, flags uint) error {
	if len(engines) == 0 {
		return nil
	}
	arr := (*[1 << 28]*C.raftq_t)(C.malloc(C.size_t(len(engines)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	for i, e := range engines {
		arr[i] = e.h
	}
	return engines[0].err(C.raftq_sweep_many_async((**C.raftq_t)(unsafe.Pointer(arr)), C.uint32_t(len(engines)), C.uint(flags)))
}

// StepSubmit / StepCollect: the pipelined Step (three batches in flight); StepResults reads the collected batch's
// records in place.
func (e *Engine) StepSubmit(msgs []Msg) error {
	if len(msgs) == 0 {
		return nil
	}
	return e.err(C.raftq_step_submit(e.h, (*C.raftq_msg_t)(unsafe.Pointer(&msgs[0])), C.uint64_t(len(msgs))))
}

func (e *Engine) StepCollect(out []StepOut) (groupsTouched uint64, err error) {
	var p *C.raftq_step_out_t
	if len(out) > 0 {
		p = (*C.raftq_step_out_t)(unsafe.Pointer(&out[0]))
	}
	var c C.raftq_step_counts_t
	err = e.err(C.raftq_step_collect(e.h, p, &c))
	return uint64(c.n_groups_touched), err
}

func (e *Engine) StepResults() ([]StepOut, error) {
	var p *C.raftq_step_out_t
	var n C.uint64_t
	if err := e.err(C.raftq_step_results(e.h, &p, &n)); err != nil {
		return nil, err
	}
	return unsafe.Slice((*StepOut)(unsafe.Pointer(p)), int(n)), nil
}
