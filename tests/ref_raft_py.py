"""A third, independent statement of etcd raft's Step for payload-free messages -- pure
Python, object-per-group, shaped like the 2015-era Go original (a `raft` struct with a
`prs` map of Progress and a `votes` map, `step` as a function pointer swapped by the
become* methods).  TEST INFRASTRUCTURE: it exists so that three implementations written in
three different shapes (this one, the sequential C oracle oracle/raftq_step_oracle.c, the
HIP kernel) have to agree on every output byte; nothing in the product imports it.

Source of the semantics: github.com/coreos/etcd/raft (un-vendored dependency of the
reference, raft.go:27-34), reached through rc.node.Step (raft.go:268-270) and rc.node.Tick
(raft.go:223-224); Raft paper sections 5.1-5.4.  PARITY UNPINNED like the C oracle.
"""
from __future__ import annotations

from typing import Optional
from dataclasses import dataclass, field

NONE = 0  # raft.None
MsgHup, MsgBeat, MsgApp, MsgAppResp, MsgVote, MsgVoteResp, MsgHeartbeat, MsgHeartbeatResp = 0, 1, 3, 4, 5, 6, 8, 9
StateFollower, StateCandidate, StateLeader = 0, 1, 2
(OutNone, OutVoteResp, OutHeartbeatResp, OutCampaign, OutBecameLeader, OutProgress, OutBcastHeartbeat,
 OutAppend) = range(8)
OutAppended = 8
OutDeferred = 9
FlagHardState, FlagCommitted, FlagUpdated, FlagSteppedDown = 1, 2, 4, 8


@dataclass
class Message:
    type: int
    frm: int = 0  # raft ID (1-based); 0 for local messages
    term: int = 0
    log_term: int = 0
    index: int = 0
    commit: int = 0
    reject: bool = False
    entries: Optional[tuple] = None  # MsgApp: the Terms of the entries it carries; None = not said (header only)
    barrier: bool = False  # MsgApp: if it is left to the log's owner, the group's later messages of the batch wait


@dataclass
class Progress:
    match: int = 0

    def maybe_update(self, n: int) -> bool:
        if self.match < n:
            self.match = n
            return True
        return False


@dataclass
class Result:
    type: int = OutNone
    index: int = 0
    log_term: int = 0
    reject: int = 0
    flags: int = 0


@dataclass
class Raft:
    id: int            # 1-based raft ID
    peers: list        # all IDs, including id
    term: int = 0
    vote: int = NONE
    lead: int = NONE
    state: int = StateFollower
    elapsed: int = 0
    committed: int = 0
    last_index: int = 0
    last_term: int = 0
    first_index_of_term: int = 0  # compact form of raftLog.term(i) == r.Term (DESIGN.md "term gate")
    prs: dict = field(default_factory=dict)
    votes: dict = field(default_factory=dict)

    def __post_init__(self):
        if not self.prs:
            self.prs = {p: Progress() for p in self.peers}
            self.prs[self.id].match = self.last_index

    def q(self) -> int:
        return len(self.prs) // 2 + 1

    # -- raftLog -----------------------------------------------------------
    def is_up_to_date(self, lasti: int, term: int) -> bool:
        return term > self.last_term or (term == self.last_term and lasti >= self.last_index)

    def commit_to(self, tocommit: int) -> None:
        tocommit = min(tocommit, self.last_index)  # upstream panics beyond the tail; the engine clamps
        if self.committed < tocommit:
            self.committed = tocommit

    def maybe_commit(self) -> bool:
        mis = sorted((pr.match for pr in self.prs.values()), reverse=True)
        mci = mis[self.q() - 1]
        term_ok = self.first_index_of_term != 0 and mci >= self.first_index_of_term
        if mci > self.committed and term_ok:
            self.committed = mci
            return True
        return False

    # -- state transitions ---------------------------------------------------
    def reset(self, term: int) -> None:
        if self.term != term:
            self.term = term
            self.vote = NONE
        self.lead = NONE
        self.elapsed = 0
        self.votes = {}
        for p in self.prs:
            self.prs[p] = Progress(match=self.last_index if p == self.id else 0)

    def become_follower(self, term: int, lead: int) -> None:
        self.reset(term)
        self.lead = lead
        self.state = StateFollower
        self.first_index_of_term = 0

    def become_candidate(self) -> None:
        self.reset(self.term + 1)
        self.vote = self.id
        self.state = StateCandidate
        self.first_index_of_term = 0

    def become_leader(self) -> None:
        self.reset(self.term)
        self.lead = self.id
        self.state = StateLeader
        # appendEntry(pb.Entry{Data: nil})
        self.last_index += 1
        self.last_term = self.term
        self.first_index_of_term = self.last_index
        self.prs[self.id].maybe_update(self.last_index)
        self.maybe_commit()

    def poll(self, frm: int, v: bool) -> int:
        if frm not in self.votes:
            self.votes[frm] = v
        return sum(1 for g in self.votes.values() if g)

    # -- Step --------------------------------------------------------------
    def step(self, m: Message) -> Result:
        """one message of a BATCH: `self.held` (cleared by the caller between batches) is the barrier's state"""
        if getattr(self, "held", False):
            return Result(OutDeferred)
        before = (self.term, self.vote, self.committed, self.state)
        res = self._step(m)
        if m.type == MsgApp and m.barrier and res.type == OutAppend:
            self.held = True
        if (self.term, self.vote, self.committed) != before[:3]:
            res.flags |= FlagHardState
        if self.committed != before[2]:
            res.flags |= FlagCommitted
        if before[3] != StateFollower and self.state == StateFollower:
            res.flags |= FlagSteppedDown
        return res

    def _step(self, m: Message) -> Result:
        if m.type == MsgHup:
            if self.state == StateLeader:
                return Result()
            self.become_candidate()
            if self.q() == self.poll(self.id, True):
                self.become_leader()
                return Result(OutBecameLeader, self.last_index, self.last_term)
            return Result(OutCampaign, self.last_index, self.last_term)
        if m.term == 0:
            pass
        elif m.term > self.term:
            self.become_follower(m.term, NONE if m.type == MsgVote else m.frm)
        elif m.term < self.term:
            return Result()
        return {StateLeader: self.step_leader, StateCandidate: self.step_candidate,
                StateFollower: self.step_follower}[self.state](m)

    def step_leader(self, m: Message) -> Result:
        if m.type == MsgBeat:
            return Result(OutBcastHeartbeat)
        if m.type == MsgVote:
            return Result(OutVoteResp, reject=1)
        if m.type == MsgAppResp:
            pr = self.prs[m.frm]
            res = Result(OutProgress, reject=int(m.reject))
            if not m.reject and pr.maybe_update(min(m.index, self.last_index)):
                res.flags |= FlagUpdated
                self.maybe_commit()
            res.index = pr.match
            return res
        if m.type == MsgHeartbeatResp:
            return Result(OutProgress, index=self.prs[m.frm].match)
        return Result()

    def handle_append_entries(self, m: Message) -> Result:
        """handleAppendEntries once the header is accepted.  Only a message that says what it carries AND lands on the
        tail is finished here: matchTerm holds, nothing lies behind the tail for findConflict, the entries go on,
        commitTo(min(m.Commit, lastnewi)).  Everything else is the log owner's (OutAppend)."""
        if m.entries is not None and m.index == self.last_index and m.log_term == self.last_term:
            if m.entries:
                self.last_index = m.index + len(m.entries)
                self.last_term = m.entries[-1]
            self.commit_to(min(m.commit, self.last_index))
            return Result(OutAppended, index=self.last_index)
        return Result(OutAppend)

    def step_candidate(self, m: Message) -> Result:
        if m.type == MsgApp:
            self.become_follower(self.term, m.frm)
            return self.handle_append_entries(m)
        if m.type == MsgHeartbeat:
            self.become_follower(self.term, m.frm)
            self.commit_to(m.commit)
            return Result(OutHeartbeatResp)
        if m.type == MsgVote:
            return Result(OutVoteResp, reject=1)
        if m.type == MsgVoteResp:
            gr = self.poll(m.frm, not m.reject)
            if self.q() == gr:
                self.become_leader()
                return Result(OutBecameLeader, self.last_index, self.last_term)
            if self.q() == len(self.votes) - gr:
                self.become_follower(self.term, NONE)
        return Result()

    def step_follower(self, m: Message) -> Result:
        if m.type == MsgApp:
            self.elapsed = 0
            self.lead = m.frm
            return self.handle_append_entries(m)
        if m.type == MsgHeartbeat:
            self.elapsed = 0
            self.lead = m.frm
            self.commit_to(m.commit)
            return Result(OutHeartbeatResp)
        if m.type == MsgVote:
            if (self.vote == NONE or self.vote == m.frm) and self.is_up_to_date(m.index, m.log_term):
                self.elapsed = 0
                self.vote = m.frm
                return Result(OutVoteResp)
            return Result(OutVoteResp, reject=1)
        return Result()


def from_node_state(s, g: int) -> Raft:
    """Build the object form of group g of an oracle.pyoracle.NodeState."""
    n = s.N
    r = Raft(id=s.self_peer + 1, peers=list(range(1, n + 1)), term=int(s.term[g]), vote=int(s.vote[g]),
             lead=int(s.lead[g]), state=int(s.role[g]), elapsed=int(s.elapsed[g]), committed=int(s.committed[g]),
             last_index=int(s.last_index[g]), last_term=int(s.last_term[g]), first_index_of_term=int(s.first_idx[g]))
    r.prs = {p + 1: Progress(int(s.match[p, g])) for p in range(n)}
    r.votes = {p + 1: (int(s.votes[p, g]) == 1) for p in range(n) if int(s.votes[p, g]) in (1, 2)}
    return r


def matches_node_state(r: Raft, s, g: int) -> bool:
    n = s.N
    scal = (r.term, r.vote, r.lead, r.state, r.elapsed, r.committed, r.last_index, r.last_term, r.first_index_of_term)
    want = tuple(int(x[g]) for x in (s.term, s.vote, s.lead, s.role, s.elapsed, s.committed, s.last_index,
                                     s.last_term, s.first_idx))
    if scal != want:
        return False
    if [r.prs[p + 1].match for p in range(n)] != [int(s.match[p, g]) for p in range(n)]:
        return False
    votes = [1 if r.votes.get(p + 1) is True else 2 if r.votes.get(p + 1) is False else 0 for p in range(n)]
    return votes == [int(s.votes[p, g]) for p in range(n)]
