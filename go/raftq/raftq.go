// Package raftq is the cgo binding of libraftq.so (include/raftq.h): the
// MI355X batched quorum engine that stands in for the per-group
// maybeCommit / poll arithmetic raftsql reaches through raft.Node
// (reference raft.go:214, 224, 269).
//
// STATUS: SOURCE ONLY.  The build image has no Go toolchain, so this file has
// never been compiled or run; it documents the binding a maintainer would add
// (see INTEGRATION.md).  The tested host mirror is raftsql_amd/engine.py.
package raftq

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raftsql_amd -lraftq -Wl,-rpath,${SRCDIR}/../../raftsql_amd
#include <stdlib.h>
#include <string.h>
#include "raftq.h"

// The handle-less error text (raftq_last_error(NULL), raftq_set_last_error(NULL)) is thread-local in the
// library, and a goroutine may change OS thread between two cgo calls.  These helpers make the failing call
// and the fetch of its text ONE cgo call, so the text always belongs to the call (ADVICE r01).
static int raftq_create_msg(int device, uint64_t groups, uint32_t peers, raftq_t** out, char* msg, size_t cap) {
  int rc = raftq_create(device, groups, peers, out);
  if (rc != RAFTQ_OK && cap) { strncpy(msg, raftq_last_error(NULL), cap - 1); msg[cap - 1] = 0; }
  return rc;
}
static int raftq_device_count_msg(int* n, char* msg, size_t cap) {
  int rc = raftq_device_count(n);
  if (rc != RAFTQ_OK && cap) { strncpy(msg, raftq_last_error(NULL), cap - 1); msg[cap - 1] = 0; }
  return rc;
}
static int raftq_set_create_msg(raftq_t* const* hs, uint32_t n, raftq_set_t** out, char* msg, size_t cap) {
  int rc = raftq_set_create(hs, n, out);
  if (rc != RAFTQ_OK && cap) { strncpy(msg, raftq_set_last_error(NULL), cap - 1); msg[cap - 1] = 0; }
  return rc;
}
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"
)

// Sweep flags (mirror RAFTQ_SWEEP_*).
const (
	SweepCommit  = 0x01
	SweepGated   = 0x02
	SweepVotes   = 0x04
	SweepNoAdopt = 0x08
	SweepLDS     = 0x10
	SweepChanged = 0x20
	SweepStream  = 0x40
	SweepCached  = 0x80
)

// Vote slot / outcome encodings.
const (
	VoteNone, VoteGranted, VoteRejected     = 0, 1, 2
	OutcomePending, OutcomeWon, OutcomeLost = 0, 1, 2
)

// Counts are the tallies of one sweep.
type Counts struct{ Changed, Won, Lost uint64 }

// Delta is one MsgAppResp-shaped update; layout-identical to raftq_delta_t.
type Delta struct {
	Group, Match uint64
	Peer, _      uint32
}

// VoteDelta is one MsgVoteResp-shaped update; layout-identical to raftq_vote_delta_t.
type VoteDelta struct {
	Group uint64
	Peer  uint32
	Vote  uint8
	_     [3]uint8
}

// Advance is one group whose commit index moved; layout-identical to raftq_advance_t.
type Advance struct{ Group, Old, New uint64 }

// Engine owns the resident quorum state of G groups x N peers on one GPU.
// Like raft.Node it must be driven from ONE goroutine.
type Engine struct {
	h      *C.raftq_t
	Groups uint64
	Peers  uint32
	// walBegun: a WalSaveBegin that enqueued something is waiting for its WalSaveEnd; walPrevCrc: what an End answers when the
	// Begin before it had no records (nothing was enqueued, the chain's CRC is where it was)
	walBegun   bool
	walPrevCrc uint32
}

func (e *Engine) err(rc C.int) error {
	if rc == C.RAFTQ_OK {
		return nil
	}
	return fmt.Errorf("raftq: %d: %s", int(rc), C.GoString(C.raftq_last_error(e.h)))
}

// DeviceCount reports the GPUs libraftq can see.
func DeviceCount() (int, error) {
	var n C.int
	var msg [256]C.char
	if rc := C.raftq_device_count_msg(&n, &msg[0], 256); rc != C.RAFTQ_OK {
		return 0, errors.New(C.GoString(&msg[0]))
	}
	return int(n), nil
}

// New allocates zeroed state for groups x peers on `device`.
func New(device int, groups uint64, peers uint32) (*Engine, error) {
	var h *C.raftq_t
	var msg [256]C.char
	if rc := C.raftq_create_msg(C.int(device), C.uint64_t(groups), C.uint32_t(peers), &h, &msg[0], 256); rc != C.RAFTQ_OK {
		return nil, fmt.Errorf("raftq: %d: %s", int(rc), C.GoString(&msg[0]))
	}
	return &Engine{h: h, Groups: groups, Peers: peers}, nil
}

// Close frees the device state.
func (e *Engine) Close() { C.raftq_destroy(e.h); e.h = nil }

// LoadMatch bulk-loads match[p*G+g] and committed[g]; either may be nil.
// The slices are only read during the call (cgo pointer rule).
func (e *Engine) LoadMatch(match, committed []uint64) error {
	var m, c *C.uint64_t
	if match != nil {
		m = (*C.uint64_t)(unsafe.Pointer(&match[0]))
	}
	if committed != nil {
		c = (*C.uint64_t)(unsafe.Pointer(&committed[0]))
	}
	return e.err(C.raftq_load_match(e.h, m, c))
}

// LoadTerms loads cur_term[g] and the first log index of that term (0 = none).
func (e *Engine) LoadTerms(curTerm, firstIdxCurTerm []uint64) error {
	return e.err(C.raftq_load_terms(e.h, (*C.uint64_t)(unsafe.Pointer(&curTerm[0])),
		(*C.uint64_t)(unsafe.Pointer(&firstIdxCurTerm[0]))))
}

// LoadVotes loads votes[p*G+g].
func (e *Engine) LoadVotes(votes []uint8) error {
	return e.err(C.raftq_load_votes(e.h, (*C.uint8_t)(unsafe.Pointer(&votes[0]))))
}

// ApplyDeltas scatters MsgAppResp updates (Progress.maybeUpdate: max).
func (e *Engine) ApplyDeltas(d []Delta) error {
	if len(d) == 0 {
		return nil
	}
	return e.err(C.raftq_apply_deltas(e.h, (*C.raftq_delta_t)(unsafe.Pointer(&d[0])), C.uint64_t(len(d))))
}

// ApplyVoteDeltas records MsgVoteResp answers (poll: first response wins).
func (e *Engine) ApplyVoteDeltas(d []VoteDelta) error {
	if len(d) == 0 {
		return nil
	}
	return e.err(C.raftq_apply_vote_deltas(e.h, (*C.raftq_vote_delta_t)(unsafe.Pointer(&d[0])), C.uint64_t(len(d))))
}

// StepAsync enqueues one sweep over all groups.
func (e *Engine) StepAsync(flags uint) error { return e.err(C.raftq_step_async(e.h, C.uint(flags))) }

// Wait blocks until the sweep is done and returns its tallies.
func (e *Engine) Wait() (Counts, error) {
	var c C.raftq_counts_t
	if err := e.err(C.raftq_wait(e.h, &c)); err != nil {
		return Counts{}, err
	}
	return Counts{uint64(c.n_changed), uint64(c.n_won), uint64(c.n_lost)}, nil
}

// CollectChanged returns the groups the last SweepChanged sweep advanced,
// ascending by group: the batched equivalent of Ready.HardState.Commit.
func (e *Engine) CollectChanged(buf []Advance) ([]Advance, uint64, error) {
	var n C.uint64_t
	var p *C.raftq_advance_t
	if len(buf) > 0 {
		p = (*C.raftq_advance_t)(unsafe.Pointer(&buf[0]))
	}
	if err := e.err(C.raftq_collect_changed(e.h, p, C.uint64_t(len(buf)), &n)); err != nil {
		return nil, 0, err
	}
	k := uint64(n)
	if k > uint64(len(buf)) {
		k = uint64(len(buf))
	}
	return buf[:k], uint64(n), nil
}

// ReadCommitted copies the current commit index of every group.
func (e *Engine) ReadCommitted(out []uint64) error {
	return e.err(C.raftq_read_committed(e.h, (*C.uint64_t)(unsafe.Pointer(&out[0]))))
}

// ReadOutcome copies the vote outcome of every group.
func (e *Engine) ReadOutcome(out []uint8) error {
	return e.err(C.raftq_read_outcome(e.h, (*C.uint8_t)(unsafe.Pointer(&out[0]))))
}

// ---- sweep sets (include/raftq.h "sweep sets"): K engines of one shape, ONE dispatch per sweep -------------

// Set launch shapes (mirror RAFTQ_SET_*).
const (
	SetGrid       = 0
	SetPersistent = 1
)

// Set sweeps its member engines with a single kernel launch.  Close the Set before its members.
type Set struct {
	s       *C.raftq_set_t
	Members []*Engine
}

// NewSet re-homes every member onto the set's stream (raftq_set_create).
func NewSet(members []*Engine) (*Set, error) {
	if len(members) == 0 {
		return nil, errors.New("raftq: empty set")
	}
	// the handle array lives in C memory: Go pointers to Go pointers must not cross cgo
	arr := (*[1 << 28]*C.raftq_t)(C.malloc(C.size_t(len(members)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(arr))
	for i, e := range members {
		arr[i] = e.h
	}
	var s *C.raftq_set_t
	var msg [256]C.char
	if rc := C.raftq_set_create_msg((**C.raftq_t)(unsafe.Pointer(arr)), C.uint32_t(len(members)), &s, &msg[0], 256); rc != C.RAFTQ_OK {
		return nil, fmt.Errorf("raftq: %d: %s", int(rc), C.GoString(&msg[0]))
	}
	return &Set{s: s, Members: members}, nil
}

func (s *Set) err(rc C.int) error {
	if rc == C.RAFTQ_OK {
		return nil
	}
	return fmt.Errorf("raftq: %d: %s", int(rc), C.GoString(C.raftq_set_last_error(s.s)))
}

// Close gives every member a stream of its own back (raftq_set_destroy).
func (s *Set) Close() { C.raftq_set_destroy(s.s); s.s = nil }

// Mode selects the launch shape; workgroups 0 keeps the default residency of the persistent walk.
func (s *Set) Mode(mode int, workgroups uint32) error {
	return s.err(C.raftq_set_mode(s.s, C.int(mode), C.uint32_t(workgroups)))
}

// SweepAsync enqueues one pass over every member: the G-fold Ready loop of raft.go:220-245, K-fold.
func (s *Set) SweepAsync(flags uint) error { return s.err(C.raftq_set_sweep_async(s.s, C.uint(flags))) }

// Wait blocks until the set's stream is idle; perMember (len == members, or nil) and total receive the tallies.
func (s *Set) Wait(perMember []Counts, total *Counts) error {
	var pm *C.raftq_counts_t
	if perMember != nil {
		pm = (*C.raftq_counts_t)(unsafe.Pointer(&perMember[0]))
	}
	return s.err(C.raftq_set_wait(s.s, pm, (*C.raftq_counts_t)(unsafe.Pointer(total))))
}

// SweepMany is the same K sweeps as K launches (raftq_sweep_many_async): for callers without a Set.
func SweepMany(engines []*Engine, flags uint) error {
	for _, e := range engines {
		if err := e.StepAsync(flags); err != nil {
			return err
		}
	}
	return nil
}

// CloneStateFrom copies src's quorum state device-to-device (raftq_clone_state).
func (e *Engine) CloneStateFrom(src *Engine) error { return e.err(C.raftq_clone_state(e.h, src.h)) }

// CollectBeats lists the leader groups the last Tick sent MsgBeat to (raftq_collect_beats).
func (e *Engine) CollectBeats(out []uint64) (uint64, error) {
	var n C.uint64_t
	var p *C.uint64_t
	if len(out) > 0 {
		p = (*C.uint64_t)(unsafe.Pointer(&out[0]))
	}
	rc := C.raftq_collect_beats(e.h, p, C.uint64_t(len(out)), &n)
	return uint64(n), e.err(rc)
}
