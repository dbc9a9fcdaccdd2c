// wire.go -- cgo binding of include/raftq_wire.h: batched raftpb.Message stream frames and
// walpb.Record WAL frames, and Step fed straight from received frames.
//
// SOURCE ONLY: never compiled or run (no Go toolchain in the build image; see README.md).
// It is the binding INTEGRATION.md section 4 describes: where a G-group raftsql would call
// it instead of G x rc.transport.Send / rc.wal.Save / w.ReadAll (raft.go:230, :228, :124).
package raftq

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raftsql_amd -lraftq -Wl,-rpath,${SRCDIR}/../../raftsql_amd
#include "raftq_wire.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// WireMsg is layout-identical to raftq_wire_msg_t (64 bytes); its first fields are Msg's, so a
// decoded slice can be reinterpreted as []Msg for StepBatch.
type WireMsg struct {
	Group, Term, LogTerm, Index, Commit, RejectHint uint64
	From                                            uint32 // peer slot = raft ID - 1
	Type, Reject, To, Flags                         uint8
	EntFirst, NEnts                                 uint32
}

// WireEnt is layout-identical to raftq_wire_ent_t (32 bytes).
type WireEnt struct {
	Term, Index, DataOff uint64
	DataLen, Type        uint32
}

// WalRec is layout-identical to raftq_wal_rec_t (48 bytes).
type WalRec struct {
	Group, Term, Index, DataOff uint64
	DataLen, Vote, Crc          uint32
	Kind, EntryType, Flags, _   uint8
}

const (
	WireMalformed = C.RAFTQ_WIRE_F_MALFORMED
	WireSnapshot  = C.RAFTQ_WIRE_F_SNAPSHOT
	WireGroup     = C.RAFTQ_WIRE_F_GROUP

	WalMetadata = C.RAFTQ_WAL_METADATA
	WalEntry    = C.RAFTQ_WAL_ENTRY
	WalState    = C.RAFTQ_WAL_STATE
	WalCrc      = C.RAFTQ_WAL_CRC
	WalSnapshot = C.RAFTQ_WAL_SNAPSHOT

	WalFMalformed = C.RAFTQ_WAL_F_MALFORMED
	WalFBadCRC    = C.RAFTQ_WAL_F_BADCRC // wal.ErrCRCMismatch
)

func bytesPtr(b []byte) unsafe.Pointer {
	if len(b) == 0 {
		return nil
	}
	return unsafe.Pointer(&b[0])
}

// HostAlloc returns page-locked memory (raftq_host_alloc): buffers handed to the calls below
// from it move by direct DMA.  Not Go-managed memory: free it with HostFree.
func HostAlloc(n int) ([]byte, error) {
	var p unsafe.Pointer
	if rc := C.raftq_host_alloc(&p, C.uint64_t(n)); rc != C.RAFTQ_OK {
		return nil, fmt.Errorf("raftq: %d: %s", int(rc), C.GoString(C.raftq_last_error(nil)))
	}
	return unsafe.Slice((*byte)(p), n), nil
}

func HostFree(b []byte) {
	if len(b) != 0 {
		C.raftq_host_free(unsafe.Pointer(&b[0]))
	}
}

// EncodeMessages marshals msgs (entries in ents, payloads in pool) into rafthttp stream frames in
// out -- what messageEncoder.encode does per message behind rc.transport.Send(rd.Messages)
// (raft.go:230).  frameOff (len(msgs)+1, or nil) receives the frame boundaries.  If out is too
// small the error is returned and `need` says how many bytes are required.
func (e *Engine) EncodeMessages(msgs []WireMsg, ents []WireEnt, pool, out []byte, frameOff []uint64) (need uint64, err error) {
	if len(msgs) == 0 {
		return 0, nil
	}
	var c C.raftq_wire_counts_t
	var pe *C.raftq_wire_ent_t
	if len(ents) > 0 {
		pe = (*C.raftq_wire_ent_t)(unsafe.Pointer(&ents[0]))
	}
	var po *C.uint64_t
	if len(frameOff) > len(msgs) {
		po = (*C.uint64_t)(unsafe.Pointer(&frameOff[0]))
	}
	rc := C.raftq_wire_encode(e.h, (*C.raftq_wire_msg_t)(unsafe.Pointer(&msgs[0])), C.uint64_t(len(msgs)), pe,
		C.uint64_t(len(ents)), bytesPtr(pool), C.uint64_t(len(pool)), bytesPtr(out), C.uint64_t(len(out)), po, &c)
	return uint64(c.bytes), e.err(rc)
}

// ScanFrames walks the length words of buf (bigEndian: rafthttp streams; little: WAL files) and
// returns the boundaries of the whole frames and the bytes they cover; a torn tail stays with the caller.
func ScanFrames(buf []byte, bigEndian bool, off []uint64) (frames int, consumed uint64) {
	if len(off) < 2 {
		return 0, 0
	}
	be := C.int(0)
	if bigEndian {
		be = 1
	}
	var n, used C.uint64_t
	C.raftq_wire_scan_frames(bytesPtr(buf), C.uint64_t(len(buf)), be, (*C.uint64_t)(unsafe.Pointer(&off[0])),
		C.uint64_t(len(off)-1), &n, &used)
	return int(n), uint64(used)
}

// DecodeMessages unmarshals the frames of stream delimited by frameOff into msgs (len(frameOff)-1)
// and their entry headers into ents; Entry.Data stays in stream (WireEnt.DataOff points into it).
func (e *Engine) DecodeMessages(stream []byte, frameOff []uint64, msgs []WireMsg, ents []WireEnt) (nEnts, nMalformed uint64, err error) {
	n := len(frameOff) - 1
	if n <= 0 {
		return 0, 0, nil
	}
	var c C.raftq_wire_counts_t
	var pe *C.raftq_wire_ent_t
	if len(ents) > 0 {
		pe = (*C.raftq_wire_ent_t)(unsafe.Pointer(&ents[0]))
	}
	rc := C.raftq_wire_decode(e.h, bytesPtr(stream), C.uint64_t(len(stream)), (*C.uint64_t)(unsafe.Pointer(&frameOff[0])),
		C.uint64_t(n), (*C.raftq_wire_msg_t)(unsafe.Pointer(&msgs[0])), pe, C.uint64_t(len(ents)), &c)
	return uint64(c.n_ents), uint64(c.n_malformed), e.err(rc)
}

// StepSubmitWire is raftNode.Process (raft.go:268-270) for a whole receive buffer: the frames are
// unmarshalled on the GPU and stepped there; collect with StepCollect as for StepSubmit.
func (e *Engine) StepSubmitWire(stream []byte, frameOff []uint64) error {
	n := len(frameOff) - 1
	if n <= 0 {
		return nil
	}
	return e.err(C.raftq_step_submit_wire(e.h, bytesPtr(stream), C.uint64_t(len(stream)),
		(*C.uint64_t)(unsafe.Pointer(&frameOff[0])), C.uint64_t(n)))
}

// StepWireMsgs / StepWireEntries: the decoded records of the batch just collected (MsgApp entry
// ranges for the log's owner); valid until the next submit into that slot.
func (e *Engine) StepWireMsgs() ([]WireMsg, error) {
	var p *C.raftq_wire_msg_t
	var n C.uint64_t
	if rc := C.raftq_step_wire_msgs(e.h, &p, &n); rc != C.RAFTQ_OK {
		return nil, e.err(rc)
	}
	return unsafe.Slice((*WireMsg)(unsafe.Pointer(p)), int(n)), nil
}

func (e *Engine) StepWireEntries() ([]WireEnt, error) {
	var p *C.raftq_wire_ent_t
	var n C.uint64_t
	if rc := C.raftq_step_wire_entries(e.h, &p, &n); rc != C.RAFTQ_OK {
		return nil, e.err(rc)
	}
	return unsafe.Slice((*WireEnt)(unsafe.Pointer(p)), int(n)), nil
}

// WalSave is rc.wal.Save(rd.HardState, rd.Entries) (raft.go:228) for every group's Ready at once:
// recs in order (a STATE record per dirty HardState, an ENTRY record per entry), appended to a
// segment whose running CRC is prevCrc.  Returns the bytes to write and the CRC to carry on.
func (e *Engine) WalSave(recs []WalRec, pool []byte, prevCrc uint32, out []byte) (n uint64, lastCrc uint32, err error) {
	if len(recs) == 0 {
		return 0, prevCrc, nil
	}
	var c C.raftq_wal_counts_t
	rc := C.raftq_wal_encode(e.h, (*C.raftq_wal_rec_t)(unsafe.Pointer(&recs[0])), C.uint64_t(len(recs)), bytesPtr(pool),
		C.uint64_t(len(pool)), C.uint32_t(prevCrc), bytesPtr(out), C.uint64_t(len(out)), nil, &c)
	return uint64(c.bytes), uint32(c.last_crc), e.err(rc)
}

// WalReadAll is w.ReadAll() (raft.go:124) for the frames of a segment: records parsed, the CRC
// chain recomputed and compared.  valid < len(recs) is where the reference would log.Fatalf
// (raft.go:126): recs[valid].Flags says whether it did not parse or its CRC mismatched.
func (e *Engine) WalReadAll(data []byte, frameOff []uint64, prevCrc uint32, recs []WalRec) (valid uint64, lastCrc uint32, err error) {
	n := len(frameOff) - 1
	if n <= 0 {
		return 0, prevCrc, nil
	}
	var c C.raftq_wal_counts_t
	rc := C.raftq_wal_decode(e.h, bytesPtr(data), C.uint64_t(len(data)), (*C.uint64_t)(unsafe.Pointer(&frameOff[0])),
		C.uint64_t(n), C.uint32_t(prevCrc), (*C.raftq_wal_rec_t)(unsafe.Pointer(&recs[0])), &c)
	return uint64(c.n_valid), uint32(c.last_crc), e.err(rc)
}

// StepStageWire returns the arrays the next StepSubmitWire will take, with room for nCap frames and nbytesCap stream
// bytes (raftq_step_stage_wire): device memory behind a large BAR -- fill them in place and submit slices of them.
func (e *Engine) StepStageWire(nCap, nbytesCap int) (frameOff []uint64, stream []byte, err error) {
	var po *C.uint64_t
	var ps unsafe.Pointer
	if rc := C.raftq_step_stage_wire(e.h, C.uint64_t(nCap), C.uint64_t(nbytesCap), &po, &ps); rc != C.RAFTQ_OK {
		return nil, nil, e.err(rc)
	}
	return unsafe.Slice((*uint64)(unsafe.Pointer(po)), nCap+1), unsafe.Slice((*byte)(ps), nbytesCap), nil
}

// StepFrames is a node's inbound half of a turn as ONE submission and one wait (raftq_step_frames): rafthttp's decoder, the
// checks a node makes on what it received and rc.node.Step (raft.go:268-270) for every frame, in arrival order.  The decoder
// says in every record's flag byte what Step made of it: a frame that is nobody's is skipped (OutSkipped), a MsgProp is
// held for the log's owner (OutHeld, the rest of its group OutDeferred), a MsgApp is a barrier that says what it carries.
// msgs / ents / stream / frameOff must be page-locked (HostAlloc).  Results: StepResults(), one per frame.  nEnts may exceed
// len(ents): the frames have been stepped all the same, WireDecode fetches the remaining headers.
func (e *Engine) StepFrames(stream []byte, frameOff []uint64, tailAppends bool, msgs []WireMsg, ents []WireEnt) (nEnts, nMalformed uint64, err error) {
	n := len(frameOff) - 1
	if n <= 0 {
		return 0, 0, nil
	}
	if len(msgs) < n {
		return 0, 0, fmt.Errorf("raftq: StepFrames: %d frames, room for %d records", n, len(msgs))
	}
	var pe *C.raftq_wire_ent_t
	if len(ents) > 0 {
		pe = (*C.raftq_wire_ent_t)(unsafe.Pointer(&ents[0]))
	}
	ta := C.int(0)
	if tailAppends {
		ta = 1
	}
	var c C.raftq_wire_counts_t
	rc := C.raftq_step_frames(e.h, bytesPtr(stream), C.uint64_t(len(stream)), (*C.uint64_t)(unsafe.Pointer(&frameOff[0])), C.uint64_t(n), ta,
		(*C.raftq_wire_msg_t)(unsafe.Pointer(&msgs[0])), pe, C.uint64_t(len(ents)), &c)
	return uint64(c.n_ents), uint64(c.n_malformed), e.err(rc)
}

// WalSaveBegin / WalSaveEnd: rc.wal.Save (raft.go:228) in two halves, so that a turn's WAL encode and its outbound marshal
// are one submission: Begin enqueues (raftq_wal_encode_begin; page-locked buffers), the wait of the WireEncode called next
// covers it, End reports what WalSave would have.  out is not to be read, nor recs / pool reused, before End.
func (e *Engine) WalSaveBegin(recs []WalRec, pool []byte, prevCrc uint32, out []byte) error {
	e.walPrevCrc = prevCrc
	if len(recs) == 0 { // a turn with nothing to persist: the End paired with it answers (0, prevCrc, nil)
		e.walBegun = false
		return nil
	}
	err := e.err(C.raftq_wal_encode_begin(e.h, (*C.raftq_wal_rec_t)(unsafe.Pointer(&recs[0])), C.uint64_t(len(recs)), bytesPtr(pool),
		C.uint64_t(len(pool)), C.uint32_t(prevCrc), bytesPtr(out), C.uint64_t(len(out)), nil))
	e.walBegun = err == nil
	return err
}

func (e *Engine) WalSaveEnd() (n uint64, lastCrc uint32, err error) {
	if !e.walBegun {
		return 0, e.walPrevCrc, nil
	}
	e.walBegun = false
	var c C.raftq_wal_counts_t
	rc := C.raftq_wal_encode_end(e.h, &c)
	return uint64(c.bytes), uint32(c.last_crc), e.err(rc)
}

// Prop is layout-identical to raftq_prop_t (16 bytes): a group this node leads, and which of propEnts are its new entries.
type Prop struct {
	Group    uint64
	EntFirst uint32 // into propEnts
	NEnts    uint32 // >= 1
}

// PropEnt is layout-identical to raftq_prop_ent_t (16 bytes): where one proposed Entry's Data lies in the pool.
type PropEnt struct {
	DataOff uint64
	DataLen uint32
	Type    uint32 // raftpb.EntryType
}

// ProposeFrames is a node's outbound half of a turn for what it was asked to propose (raftq_propose_frames; raft.go:211-215 ->
// :227-230) as ONE submission and one wait: for every Prop the leader's appendEntry on the device-resident state and the N - 1
// MsgApps of bcastAppend, written into the encoder's input in HBM -- they never exist in host memory --, then the marshal of
// msgs (what the caller queued itself this turn) followed by those MsgApps, one run per peer slot != self, ascending.  The
// caller still owns the log (it appends the entries itself) and Progress.Next (it names only groups whose followers are all at
// the tail, and moves Next past the new entries).  Every slice must be page-locked (HostAlloc).  SOURCE ONLY (round 6).
func (e *Engine) ProposeFrames(props []Prop, propEnts []PropEnt, msgs []WireMsg, ents []WireEnt, pool, out []byte, frameOff []uint64) (need uint64, err error) {
	var pp *C.raftq_prop_t
	var ppe *C.raftq_prop_ent_t
	var pm *C.raftq_wire_msg_t
	var pe *C.raftq_wire_ent_t
	var po *C.uint64_t
	if len(props) > 0 {
		pp = (*C.raftq_prop_t)(unsafe.Pointer(&props[0]))
	}
	if len(propEnts) > 0 {
		ppe = (*C.raftq_prop_ent_t)(unsafe.Pointer(&propEnts[0]))
	}
	if len(msgs) > 0 {
		pm = (*C.raftq_wire_msg_t)(unsafe.Pointer(&msgs[0]))
	}
	if len(ents) > 0 {
		pe = (*C.raftq_wire_ent_t)(unsafe.Pointer(&ents[0]))
	}
	if len(frameOff) > 0 {
		po = (*C.uint64_t)(unsafe.Pointer(&frameOff[0]))
	}
	var c C.raftq_wire_counts_t
	rc := C.raftq_propose_frames(e.h, pp, C.uint64_t(len(props)), ppe, C.uint64_t(len(propEnts)), pm, C.uint64_t(len(msgs)), pe, C.uint64_t(len(ents)),
		bytesPtr(pool), C.uint64_t(len(pool)), bytesPtr(out), C.uint64_t(len(out)), po, &c)
	return uint64(c.bytes), e.err(rc)
}
