//go:build etcd

// etcd_pin_test.go -- the check that would PIN parity (SURVEY.md F1/F2, 8c): drive the real
// github.com/coreos/etcd/raft (the module raft.go:27-34 imports; v2.2-v2.3 line) and this repo's
// restatement with the same acknowledgements and compare the commit index after every message.
// Build with `go test -tags etcd ./go/raftq` in a GOPATH that holds that etcd checkout.
//
// SOURCE ONLY: the module is not on the build machine and there is no Go toolchain.
package raftq

import (
	"math/rand"
	"testing"

	"github.com/coreos/etcd/raft"
	pb "github.com/coreos/etcd/raft/raftpb"
)

func TestCommitIndexAgainstEtcd(t *testing.T) {
	for _, n := range []int{1, 2, 3, 4, 5, 7, 9} {
		rng := rand.New(rand.NewSource(int64(0xC0FFEE00 + n)))
		peers := make([]raft.Peer, n)
		for i := range peers {
			peers[i] = raft.Peer{ID: uint64(i + 1)}
		}
		st := raft.NewMemoryStorage()
		c := &raft.Config{ID: 1, ElectionTick: 10, HeartbeatTick: 1, Storage: st, MaxSizePerMsg: 1 << 20, MaxInflightMsgs: 256}
		rn, err := raft.NewRawNode(c, peers)
		if err != nil {
			t.Fatal(err)
		}
		drain := func() pb.HardState { // apply a Ready the way raft.go:227-235 does
			var hs pb.HardState
			for rn.HasReady() {
				rd := rn.Ready()
				st.Append(rd.Entries)
				if !raft.IsEmptyHardState(rd.HardState) {
					hs = rd.HardState
					st.SetHardState(hs)
				}
				rn.Advance(rd)
			}
			return hs
		}
		drain()
		rn.Campaign()
		drain()
		for p := 2; p <= n; p++ { // win the election
			rn.Step(pb.Message{Type: pb.MsgVoteResp, From: uint64(p), To: 1, Term: 1})
		}
		drain()
		for i := 0; i < 40; i++ {
			rn.Propose([]byte{byte(i)})
		}
		drain()
		last, _ := st.LastIndex()
		first, _ := st.FirstIndex()
		_ = first
		match := make([]uint64, n)
		match[0] = last
		committed := rn.Status().Commit
		firstIdxCurTerm := uint64(0)
		for i := uint64(1); i <= last; i++ { // the compact gate: first index of the leader's term
			if tm, _ := st.Term(i); tm == 1 {
				firstIdxCurTerm = i
				break
			}
		}
		for step := 0; step < 400 && n > 1; step++ {
			p := 1 + rng.Intn(n-1)
			idx := uint64(rng.Int63n(int64(last) + 1))
			rn.Step(pb.Message{Type: pb.MsgAppResp, From: uint64(p + 1), To: 1, Term: 1, Index: idx})
			drain()
			if idx > match[p] { // Progress.maybeUpdate
				match[p] = idx
			}
			committed = commitLikeEtcd(match, committed, firstIdxCurTerm, true)
			if got := rn.Status().Commit; got != committed {
				t.Fatalf("N=%d step %d: etcd commit %d, restatement %d (match %v)", n, step, got, committed, match)
			}
		}
	}
}
