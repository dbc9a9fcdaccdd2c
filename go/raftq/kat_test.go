// kat_test.go -- the repo's known-answer vectors (tests/golden/kat.json) run three ways wherever Go
// exists: (1) a Go loop in the SHAPE of the dependency's own code (etcd raft.maybeCommit: gather
// Match into a uint64Slice, sort.Sort(sort.Reverse(...)), index q-1; raft.poll: count granted /
// rejected in the votes map), (2) the GPU engine through this cgo binding, and -- build tag
// `etcd` -- (3) the real github.com/coreos/etcd/raft driven message by message, which is the one
// check that would PIN this repo's oracle (SURVEY.md 8c: parity is otherwise unpinned).
//
// SOURCE ONLY: never compiled or run here (no Go toolchain; the etcd module is absent).
package raftq

import (
	"encoding/json"
	"os"
	"sort"
	"testing"
)

type katFile struct {
	Mci    []struct{ Match []uint64; Mci uint64 }                                       `json:"mci"`
	Commit []struct{ Match []uint64; Committed, Ungated, FirstIdx, Gated uint64 }        `json:"commit"`
	Poll   []struct{ Votes []uint8; Outcome uint8 }                                      `json:"poll"`
}

func loadKAT(t *testing.T) katFile {
	b, err := os.ReadFile("../../tests/golden/kat.json")
	if err != nil {
		t.Fatal(err)
	}
	var k katFile
	if err := json.Unmarshal(b, &k); err != nil {
		t.Fatal(err)
	}
	return k
}

type uint64Slice []uint64

func (p uint64Slice) Len() int           { return len(p) }
func (p uint64Slice) Less(i, j int) bool { return p[i] < p[j] }
func (p uint64Slice) Swap(i, j int)      { p[i], p[j] = p[j], p[i] }

// the dependency's maybeCommit, up to the raftLog call: `mis := make(uint64Slice, 0, len(r.prs))`,
// append every Match, sort descending, `mci := mis[r.q()-1]`
func mciLikeEtcd(match []uint64) uint64 {
	mis := make(uint64Slice, 0, len(match))
	for _, m := range match {
		mis = append(mis, m)
	}
	sort.Sort(sort.Reverse(mis))
	return mis[len(match)/2]
}

// raftLog.maybeCommit with the compact gate of DESIGN.md section 2
func commitLikeEtcd(match []uint64, committed, firstIdxCurTerm uint64, gated bool) uint64 {
	mci := mciLikeEtcd(match)
	if mci > committed && (!gated || (firstIdxCurTerm != 0 && mci >= firstIdxCurTerm)) {
		return mci
	}
	return committed
}

// poll + the candidate's switch on the tally (2015-era rule: lost when rejections reach q)
func pollLikeEtcd(votes []uint8) uint8 {
	q, granted, rejected := len(votes)/2+1, 0, 0
	for _, v := range votes {
		if v == VoteGranted {
			granted++
		} else if v == VoteRejected {
			rejected++
		}
	}
	switch {
	case granted >= q:
		return OutcomeWon
	case rejected >= q:
		return OutcomeLost
	}
	return OutcomePending
}

func TestKATGoRestatement(t *testing.T) {
	k := loadKAT(t)
	for i, c := range k.Mci {
		if got := mciLikeEtcd(c.Match); got != c.Mci {
			t.Errorf("mci[%d]: %d, want %d", i, got, c.Mci)
		}
	}
	for i, c := range k.Commit {
		if got := commitLikeEtcd(c.Match, c.Committed, 0, false); got != c.Ungated {
			t.Errorf("commit[%d] ungated: %d, want %d", i, got, c.Ungated)
		}
		if c.FirstIdx != 0 || c.Gated != 0 {
			if got := commitLikeEtcd(c.Match, c.Committed, c.FirstIdx, true); got != c.Gated {
				t.Errorf("commit[%d] gated: %d, want %d", i, got, c.Gated)
			}
		}
	}
	for i, c := range k.Poll {
		if got := pollLikeEtcd(c.Votes); got != c.Outcome {
			t.Errorf("poll[%d]: %d, want %d", i, got, c.Outcome)
		}
	}
}

// the same vectors through the GPU engine: one group per vector, grouped by peer count
func TestKATEngine(t *testing.T) {
	if n, err := DeviceCount(); err != nil || n == 0 {
		t.Skip("no GPU visible to libraftq")
	}
	k := loadKAT(t)
	byN := map[int][]int{}
	for i, c := range k.Commit {
		byN[len(c.Match)] = append(byN[len(c.Match)], i)
	}
	for n, idx := range byN {
		g := uint64(len(idx))
		e, err := New(0, g, uint32(n))
		if err != nil {
			t.Fatal(err)
		}
		match := make([]uint64, uint64(n)*g) // [N][G] peer-major
		committed, first, term := make([]uint64, g), make([]uint64, g), make([]uint64, g)
		for j, i := range idx {
			c := k.Commit[i]
			for p := 0; p < n; p++ {
				match[uint64(p)*g+uint64(j)] = c.Match[p]
			}
			committed[j], first[j], term[j] = c.Committed, c.FirstIdx, 1
		}
		if err := e.LoadMatch(match, committed); err != nil {
			t.Fatal(err)
		}
		if err := e.LoadTerms(term, first); err != nil {
			t.Fatal(err)
		}
		for _, gated := range []bool{false, true} {
			flags := uint(SweepCommit | SweepNoAdopt)
			if gated {
				flags |= SweepGated
			}
			if err := e.StepAsync(flags); err != nil {
				t.Fatal(err)
			}
			if _, err := e.Wait(); err != nil {
				t.Fatal(err)
			}
			out := make([]uint64, g)
			if err := e.ReadCommitted(out); err != nil {
				t.Fatal(err)
			}
			for j, i := range idx {
				c := k.Commit[i]
				want := c.Ungated
				if gated {
					if c.FirstIdx == 0 && c.Gated == 0 {
						continue // vector has no gated answer
					}
					want = c.Gated
				}
				if out[j] != want {
					t.Errorf("engine commit[%d] gated=%v: %d, want %d", i, gated, out[j], want)
				}
			}
		}
		e.Close()
	}
}
