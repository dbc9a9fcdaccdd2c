package raftq

// Batcher is the single goroutine that replaces the per-message commit check
// raft.Node would run inside Step (reference raft.go:268-270): peers'
// MsgAppResp / MsgVoteResp are turned into deltas, applied in one scatter,
// evaluated in one sweep over every group, and the advanced groups come back
// as one compacted list -- the batched form of the Ready loop at
// raft.go:220-245.  SOURCE ONLY (no Go toolchain here); see raftq.go.

// AppResp is what Process(ctx, m) extracts from a MsgAppResp.
type AppResp struct {
	Group uint64
	Peer  uint32
	Match uint64
}

// VoteResp is what Process(ctx, m) extracts from a MsgVoteResp.
type VoteResp struct {
	Group   uint64
	Peer    uint32
	Granted bool
}

// Batcher owns one Engine.  AppC / VoteC are fed by the rafthttp handlers of
// all groups; AdvanceC carries commit-index movements to the per-group
// publishers that keep each group's commitC contract (raft.go:82-96).
type Batcher struct {
	E        *Engine
	AppC     chan AppResp
	VoteC    chan VoteResp
	AdvanceC chan []Advance
	ErrorC   chan error
	Gated    bool
	MaxBatch int
}

// Run drains whatever has arrived (up to MaxBatch), applies it, sweeps, and
// publishes the advances; it returns when AppC is closed.
func (b *Batcher) Run() {
	deltas := make([]Delta, 0, b.MaxBatch)
	votes := make([]VoteDelta, 0, b.MaxBatch)
	adv := make([]Advance, b.MaxBatch)
	flags := uint(SweepCommit | SweepVotes | SweepChanged)
	if b.Gated {
		flags |= SweepGated
	}
	for {
		deltas, votes = deltas[:0], votes[:0]
		first, ok := <-b.AppC // block for the first message, then drain
		if !ok {
			close(b.AdvanceC)
			return
		}
		deltas = append(deltas, Delta{Group: first.Group, Match: first.Match, Peer: first.Peer})
	drain:
		for len(deltas)+len(votes) < b.MaxBatch {
			select {
			case m, ok := <-b.AppC:
				if !ok {
					break drain
				}
				deltas = append(deltas, Delta{Group: m.Group, Match: m.Match, Peer: m.Peer})
			case v := <-b.VoteC:
				vd := VoteDelta{Group: v.Group, Peer: v.Peer, Vote: VoteRejected}
				if v.Granted {
					vd.Vote = VoteGranted
				}
				votes = append(votes, vd)
			default:
				break drain
			}
		}
		if err := b.E.ApplyDeltas(deltas); err != nil {
			b.ErrorC <- err
			return
		}
		if err := b.E.ApplyVoteDeltas(votes); err != nil {
			b.ErrorC <- err
			return
		}
		if err := b.E.StepAsync(flags); err != nil {
			b.ErrorC <- err
			return
		}
		if _, err := b.E.Wait(); err != nil {
			b.ErrorC <- err
			return
		}
		got, total, err := b.E.CollectChanged(adv)
		if err != nil {
			b.ErrorC <- err
			return
		}
		if total > uint64(len(adv)) { // rare: more groups advanced than the buffer
			adv = make([]Advance, total)
			got, _, err = b.E.CollectChanged(adv)
			if err != nil {
				b.ErrorC <- err
				return
			}
		}
		out := make([]Advance, len(got))
		copy(out, got)
		b.AdvanceC <- out
	}
}
