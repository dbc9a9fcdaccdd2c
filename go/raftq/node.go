// node.go -- cgo binding of include/raftq_node.h plus the per-group raftPipe surface.
//
// SOURCE ONLY: never compiled or run (no Go toolchain in the build image; see README.md).
//
// NewMultiRaftPipe is the G-group form of the reference's
//     func NewRaftPipe(id int, peers []string, proposeC chan string) *raftPipe   (raftpipe.go:9-12)
// Every group keeps the exact surface db.go consumes (raftpipe.go:3-7):
//     ProposeC chan<- string, CommitC <-chan *string, ErrorC <-chan error, Close() error
// so db.go / httpapi.go run unchanged on top of one element of MultiRaftPipe.Groups.
// What replaces raft.go's per-group goroutines (raft.go:186-187, 211) is ONE crank goroutine
// per GPU (Run) that ticks, delivers what the transport received, advances every group at
// once on the GPU and fans the commit channels out.
package raftq

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../raftsql_amd -lraftq -Wl,-rpath,${SRCDIR}/../../raftsql_amd
#include <stdlib.h>
#include "raftq_node.h"
*/
import "C"

import (
	"fmt"
	"time"
	"unsafe"
)

// RaftPipe is field-for-field the reference's raftPipe (raftpipe.go:3-7).
type RaftPipe struct {
	ProposeC chan<- string
	CommitC  <-chan *string
	ErrorC   <-chan error
	close    func() error
}

// Close is raftPipe.Close (raftpipe.go:14-17): close the proposal side, return the error.
func (rp *RaftPipe) Close() error { return rp.close() }

// Transport moves the node's frames between processes (the reference uses rafthttp,
// raft.go:170-186).  Send must not retain buf.
type Transport interface {
	Send(toPeer uint32, buf []byte)
	Recv() <-chan []byte // frames other peers polled for this node
}

// MultiRaftPipe is one raft node for G groups on one GPU.
type MultiRaftPipe struct {
	n      *C.raftq_node_t
	Groups []*RaftPipe
	peers  uint32
	stopc  chan struct{}
	donec  chan error
	Wal    WalFile // nil: no WAL is written (what the node would have saved is only counted)
}

// WalFile is where the node's WAL bytes go: the open segment file (*os.File satisfies it).
type WalFile interface {
	Write(p []byte) (int, error)
	Sync() error
}

// NewMultiRaftPipeFromWAL is NewMultiRaftPipe for a restart: walBytes is the content of the WAL
// segment (replayWAL, raft.go:122-134 -- w.ReadAll with every CRC checked on the GPU); the node keeps
// appending to `segment` on the same CRC chain.  restoreHardState=false is the reference's behaviour.
func NewMultiRaftPipeFromWAL(device, id, nPeers int, nGroups uint64, walBytes []byte, restoreHardState bool, segment WalFile, tr Transport) (*MultiRaftPipe, error) {
	var n *C.raftq_node_t
	if rc := C.raftq_node_create(C.int(device), C.uint64_t(nGroups), C.uint32_t(nPeers), C.uint32_t(id-1), &n); rc != C.RAFTQ_OK {
		return nil, fmt.Errorf("raftq_node_create: %d: %s", int(rc), C.GoString(C.raftq_last_error(nil)))
	}
	m := &MultiRaftPipe{n: n, peers: uint32(nPeers), stopc: make(chan struct{}), donec: make(chan error, 1), Wal: segment}
	restore := C.int(0)
	if restoreHardState {
		restore = 1
	}
	if len(walBytes) > 0 {
		if rc := C.raftq_node_replay_wal(n, unsafe.Pointer(&walBytes[0]), C.uint64_t(len(walBytes)), restore, nil); rc != C.RAFTQ_OK {
			return nil, nodeErr(n, rc) // the reference: log.Fatalf("raftsql: failed to read WAL (%v)", err) (raft.go:126)
		}
	}
	if segment != nil {
		C.raftq_node_wal_enable(n)
	}
	if rc := C.raftq_node_start(n, 10, 1, C.uint64_t(0x1000+id)); rc != C.RAFTQ_OK {
		return nil, nodeErr(n, rc)
	}
	m.Groups = make([]*RaftPipe, nGroups)
	for g := range m.Groups {
		m.Groups[g] = m.pipeFor(uint64(g))
	}
	go m.run(tr)
	return m, nil
}

func nodeErr(n *C.raftq_node_t, rc C.int) error {
	if rc == C.RAFTQ_OK {
		return nil
	}
	return fmt.Errorf("raftq_node: %d: %s", int(rc), C.GoString(C.raftq_node_last_error(n)))
}

// NewMultiRaftPipe: id is 1-based like the reference's (raft.go:150); wal[g] holds group g's
// logged entries (term, payload) for replayWAL (raft.go:122-134), nil for a fresh node.
func NewMultiRaftPipe(device int, id int, nPeers int, nGroups uint64, wal [][]WalEntry, tr Transport) (*MultiRaftPipe, error) {
	var n *C.raftq_node_t
	if rc := C.raftq_node_create(C.int(device), C.uint64_t(nGroups), C.uint32_t(nPeers), C.uint32_t(id-1), &n); rc != C.RAFTQ_OK {
		return nil, fmt.Errorf("raftq_node_create: %d: %s", int(rc), C.GoString(C.raftq_last_error(nil)))
	}
	m := &MultiRaftPipe{n: n, peers: uint32(nPeers), stopc: make(chan struct{}), donec: make(chan error, 1)}
	for g, ents := range wal {
		if len(ents) == 0 {
			continue
		}
		terms := make([]C.uint64_t, len(ents))
		lens := make([]C.uint32_t, len(ents))
		ptrs := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(len(ents)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
		for i, e := range ents {
			terms[i], lens[i] = C.uint64_t(e.Term), C.uint32_t(len(e.Data))
			ptrs[i] = C.CBytes(e.Data) // C memory: no Go pointer crosses the boundary inside an array
		}
		rc := C.raftq_node_replay(n, C.uint64_t(g), &terms[0], (*unsafe.Pointer)(unsafe.Pointer(ptrs)), &lens[0], C.uint64_t(len(ents)))
		for i := range ents {
			C.free(ptrs[i])
		}
		C.free(unsafe.Pointer(ptrs))
		if rc != C.RAFTQ_OK {
			return nil, nodeErr(n, rc)
		}
	}
	// raft.Config{ElectionTick: 10, HeartbeatTick: 1} (raft.go:154-155)
	if rc := C.raftq_node_start(n, 10, 1, C.uint64_t(0x1000+id)); rc != C.RAFTQ_OK {
		return nil, nodeErr(n, rc)
	}
	m.Groups = make([]*RaftPipe, nGroups)
	for g := range m.Groups {
		m.Groups[g] = m.pipeFor(uint64(g))
	}
	go m.run(tr)
	return m, nil
}

// WalEntry is one logged entry of a group.
type WalEntry struct {
	Term uint64
	Data []byte
}

// pipeFor builds group g's channels: a forwarder goroutine per direction, as raft.go:211-218
// (proposals) and publishEntries' blocking send on the unbuffered commitC (raft.go:82-96).
func (m *MultiRaftPipe) pipeFor(g uint64) *RaftPipe {
	proposeC := make(chan string)
	commitC := make(chan *string)
	errorC := make(chan error, 1)
	go func() { // proposeC -> raftq_node_propose
		for prop := range proposeC {
			b := []byte(prop)
			var p *C.char
			if len(b) > 0 {
				p = (*C.char)(unsafe.Pointer(&b[0]))
			}
			C.raftq_node_propose(m.n, C.uint64_t(g), unsafe.Pointer(p), C.uint32_t(len(b)))
		}
	}()
	go func() { // raftq_node_recv -> commitC: replayed entries, nil sentinel, live entries
		buf := make([]byte, 1<<20)
		for {
			var ln C.uint32_t
			var kind C.int
			rc := C.raftq_node_recv(m.n, C.uint64_t(g), -1, unsafe.Pointer(&buf[0]), C.uint32_t(len(buf)), &ln, &kind)
			if rc != C.RAFTQ_OK || kind == C.RAFTQ_NODE_CLOSED {
				close(commitC)
				if e := C.raftq_node_error(m.n); e != 0 {
					errorC <- nodeErr(m.n, e)
				}
				close(errorC)
				return
			}
			switch kind {
			case C.RAFTQ_NODE_SENTINEL:
				commitC <- nil // raft.go:132
			case C.RAFTQ_NODE_ENTRY:
				s := string(buf[:ln]) // a fresh copy per entry, as raft.go:88
				commitC <- &s
			}
		}
	}()
	return &RaftPipe{ProposeC: proposeC, CommitC: commitC, ErrorC: errorC,
		close: func() error { close(proposeC); return <-errorC }}
}

// run is serveChannels (raft.go:204-246) for every group at once.
func (m *MultiRaftPipe) run(tr Transport) {
	ticker := time.NewTicker(100 * time.Millisecond) // raft.go:207
	defer ticker.Stop()
	crank := time.NewTicker(200 * time.Microsecond) // batching window of one Ready iteration
	defer crank.Stop()
	wire := make([]byte, 4<<20)
	for {
		select {
		case <-ticker.C:
			C.raftq_node_tick(m.n) // rc.node.Tick() for all groups (raft.go:223-224)
		case frames := <-tr.Recv():
			if len(frames) > 0 { // rc.Process for every message in the buffer (raft.go:268-270)
				C.raftq_node_deliver(m.n, unsafe.Pointer(&frames[0]), C.uint64_t(len(frames)))
			}
			continue
		case <-crank.C:
		case <-m.stopc:
			m.donec <- nodeErr(m.n, C.raftq_node_close(m.n))
			return
		}
		var published C.uint64_t
		if rc := C.raftq_node_advance(m.n, &published); rc != C.RAFTQ_OK {
			m.donec <- nodeErr(m.n, rc) // writeError (raft.go:136-142): channels are closed by the node
			return
		}
		// rc.wal.Save(rd.HardState, rd.Entries) (raft.go:228) for every group at once: the node hands out
		// walpb.Record frames (raftq_node_wal_enable); they are on disk before anything is sent
		if m.Wal != nil {
			for {
				var ln C.uint64_t
				if rc := C.raftq_node_wal_poll(m.n, unsafe.Pointer(&wire[0]), C.uint64_t(len(wire)), &ln); rc != C.RAFTQ_OK || ln == 0 {
					break
				}
				if _, err := m.Wal.Write(wire[:ln]); err != nil {
					m.donec <- err
					return
				}
			}
			if err := m.Wal.Sync(); err != nil {
				m.donec <- err
				return
			}
		}
		// rafthttp message-stream frames (u64 big-endian length | raftpb.Message): tr.Send writes them to
		// the peer's stream as they are
		for p := uint32(0); p < m.peers; p++ { // rc.transport.Send(rd.Messages) (raft.go:230)
			for {
				var ln C.uint64_t
				if rc := C.raftq_node_poll(m.n, C.uint32_t(p), unsafe.Pointer(&wire[0]), C.uint64_t(len(wire)), &ln); rc != C.RAFTQ_OK || ln == 0 {
					break
				}
				tr.Send(p, wire[:ln])
			}
		}
	}
}

// Close stops the crank and returns the node's error (nil when it shut down cleanly).
func (m *MultiRaftPipe) Close() error {
	close(m.stopc)
	err := <-m.donec
	C.raftq_node_destroy(m.n)
	return err
}

// NodeStatus is raft.Node.Status() for one group, as include/raftq_node.h lays it out (raftq_node_status_t).
type NodeStatus struct {
	Term, Commit, LastIndex, Applied uint64
	Lead, Vote                       uint32
	Role                             uint8
}

// Statuses returns the status of groups [first, first+count) in one cgo call (raftq_node_status_batch): what a
// G-group host polls for leader discovery instead of G calls.
func (m *MultiRaftPipe) Statuses(first, count uint64) ([]NodeStatus, error) {
	if count == 0 {
		return nil, nil
	}
	raw := make([]C.raftq_node_status_t, count)
	if rc := C.raftq_node_status_batch(m.n, C.uint64_t(first), C.uint64_t(count), &raw[0]); rc != C.RAFTQ_OK {
		return nil, nodeErr(m.n, rc)
	}
	out := make([]NodeStatus, count)
	for i := range raw {
		out[i] = NodeStatus{Term: uint64(raw[i].term), Commit: uint64(raw[i].commit), LastIndex: uint64(raw[i].last_index),
			Applied: uint64(raw[i].applied), Lead: uint32(raw[i].lead), Vote: uint32(raw[i].vote), Role: uint8(raw[i].role)}
	}
	return out, nil
}

// Campaign is raft.Node.Campaign for the given groups: each gets a local MsgHup at the next turn of the crank.
func (m *MultiRaftPipe) Campaign(groups []uint64) error {
	if len(groups) == 0 {
		return nil
	}
	if rc := C.raftq_node_campaign(m.n, (*C.uint64_t)(unsafe.Pointer(&groups[0])), C.uint64_t(len(groups))); rc != C.RAFTQ_OK {
		return nodeErr(m.n, rc)
	}
	return nil
}
