// step_tables_test.go -- etcd's own Step tables AS RECALLED (tests/golden/kat.json, "upstream_step_tables_recalled":
// TestRecvMsgVote, TestAllServerStepdown, TestStepIgnoreOldTermMsg, TestHandleHeartbeat, TestLeaderAppResp) through
// the engine's batched Step, from Go.  The Python twins are tests/test_step_oracle.py (oracle) and
// tests/test_step_gpu.py (GPU).  Where the etcd module is present these tables are upstream's own tests: running
// `go test ./raft` there and this file here compares the two on exactly the inputs upstream chose.
//
// SOURCE ONLY: there is no Go toolchain on the build machine; never compiled.
package raftq

import (
	"encoding/json"
	"os"
	"testing"
)

type recalledMsg struct {
	Type       uint8  `json:"type"`
	Term       uint64 `json:"term"`
	From       uint32 `json:"from"`
	Index      uint64 `json:"index"`
	LogTerm    uint64 `json:"log_term"`
	Commit     uint64 `json:"commit"`
	Reject     uint8  `json:"reject"`
	RejectHint uint64 `json:"reject_hint"`
}

type recalledCase struct {
	Table     string            `json:"table"`
	Row       int               `json:"row"`
	N         uint32            `json:"n"`
	Self      uint32            `json:"self"`
	Init      map[string]any    `json:"init"`
	Setup     []recalledMsg     `json:"setup"`
	Msgs      []recalledMsg     `json:"msgs"`
	WantOut   []map[string]int  `json:"want_out"`
	WantState map[string]uint64 `json:"want_state"`
}

func u64(m map[string]any, k string) uint64 {
	if v, ok := m[k].(float64); ok {
		return uint64(v)
	}
	return 0
}

func TestUpstreamStepTablesAsRecalled(t *testing.T) {
	b, err := os.ReadFile("../../tests/golden/kat.json")
	if err != nil {
		t.Skip(err)
	}
	var kat struct {
		Tables struct {
			Cases []recalledCase `json:"cases"`
		} `json:"upstream_step_tables_recalled"`
	}
	if err := json.Unmarshal(b, &kat); err != nil {
		t.Fatal(err)
	}
	for _, c := range kat.Tables.Cases {
		e, err := New(0, 1, c.N)
		if err != nil {
			t.Skip(err) // no GPU: the engine has no CPU path
		}
		if err := e.SetSelf(c.Self); err != nil {
			t.Fatal(err)
		}
		// start state: one group, fields of the table's `init` (absent = zero)
		match := make([]uint64, c.N)
		if m, ok := c.Init["match"].([]any); ok {
			for p := range m {
				match[p] = uint64(m[p].(float64))
			}
		}
		if err := e.LoadMatch(match, []uint64{u64(c.Init, "committed")}); err != nil {
			t.Fatal(err)
		}
		term, first := u64(c.Init, "term"), u64(c.Init, "first_idx")
		gateTerm := term
		if first != 0 && gateTerm == 0 {
			gateTerm = 1
		}
		if first == 0 {
			gateTerm = 0
		}
		if err := e.LoadTerms([]uint64{gateTerm}, []uint64{first}); err != nil {
			t.Fatal(err)
		}
		if err := e.LoadRoles([]uint8{uint8(u64(c.Init, "role"))}, nil); err != nil {
			t.Fatal(err)
		}
		if err := e.LoadNode([]uint64{term}, []uint32{uint32(u64(c.Init, "vote"))}, []uint32{uint32(u64(c.Init, "lead"))},
			[]uint64{u64(c.Init, "last_index")}, []uint64{u64(c.Init, "last_term")}); err != nil {
			t.Fatal(err)
		}
		all := append(append([]recalledMsg{}, c.Setup...), c.Msgs...)
		outs := make([]StepOut, len(all))
		for i, m := range all {
			msg := []Msg{{Term: m.Term, LogTerm: m.LogTerm, Index: m.Index, Commit: m.Commit, RejectHint: m.RejectHint,
				From: m.From, Type: m.Type, Reject: m.Reject}}
			if _, err := e.StepBatch(msg, outs[i:i+1]); err != nil {
				t.Fatalf("%s row %d: %v", c.Table, c.Row, err)
			}
		}
		outs = outs[len(c.Setup):]
		for i, w := range c.WantOut {
			if v, ok := w["type"]; ok && int(outs[i].Type) != v {
				t.Errorf("%s row %d: out[%d].type = %d, table says %d", c.Table, c.Row, i, outs[i].Type, v)
			}
			if v, ok := w["reject"]; ok && int(outs[i].Reject) != v {
				t.Errorf("%s row %d: out[%d].reject = %d, table says %d", c.Table, c.Row, i, outs[i].Reject, v)
			}
		}
		st, err := e.ReadNode()
		if err != nil {
			t.Fatal(err)
		}
		got := map[string]uint64{"role": uint64(st.Role[0]), "term": st.Term[0], "lead": uint64(st.Lead[0]),
			"last_index": st.LastIndex[0], "committed": st.Committed[0], "match1": 0}
		if c.N > 1 {
			got["match1"] = st.Match[1]
		}
		for k, v := range c.WantState {
			if got[k] != v {
				t.Errorf("%s row %d: %s = %d, table says %d", c.Table, c.Row, k, got[k], v)
			}
		}
		e.Close()
	}
}
