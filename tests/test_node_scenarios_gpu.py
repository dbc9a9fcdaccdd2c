"""GPU: etcd's own multi-node scenarios (raft_test.go's `network` tests), AS RECALLED, through raftq_node clusters.

The module the reference imports (github.com/coreos/etcd/raft, raft.go:27-34) is absent here and so are its tests; the
scenarios below are written from memory of the 2015-era raft_test.go -- alignment evidence, not a pin (oracle
headers: "parity unpinned").  They matter for the hot path because what they assert is the commit index: when the q-th
largest match may advance it, and when the term gate (raft paper 5.4.2) must hold it back.

Upstream's `network` is message-driven: `send(m)` delivers m and everything it causes until nothing is in flight; no
timer ever fires; `cut` / `isolate` / `ignore(type)` / `recover` shape the fabric; a `nopStepper` peer swallows
everything.  `Net` below is that fabric over raftq_node: every scenario runs on G groups at once (each group an
independent copy of the scenario), so each `send` is one batched Step / encode / decode per node on the GPU.  Peer ids
are upstream's (1-based); node slot = id - 1."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HUP, BEAT, PROP, APP, APP_RESP, VOTE, VOTE_RESP, SNAP, HEARTBEAT, HEARTBEAT_RESP = range(10)
FOLLOWER, CANDIDATE, LEADER = 0, 1, 2
G = 24


class Net:
    def __init__(self, n, nop=()):
        from raftsql_amd.node import Cluster

        self.c = Cluster(G, n, seed=3)
        self.c.start()
        self.n, self.nop = n, set(nop)
        self.cuts, self.ignored = set(), set()
        self.groups = np.arange(G, dtype=np.uint64)

    def close(self):
        self.c.close()

    def node(self, i):
        return self.c.nodes[i - 1]

    def live(self):
        return [i for i in range(1, self.n + 1) if i not in self.nop]

    # -- the fabric ---------------------------------------------------------------------------------------------
    def cut(self, a, b):
        self.cuts |= {(a, b), (b, a)}

    def isolate(self, a):
        for b in range(1, self.n + 1):
            if b != a:
                self.cut(a, b)

    def ignore(self, t):
        self.ignored.add(t)

    def recover(self):
        self.cuts.clear()
        self.ignored.clear()

    def _filter(self, blob):
        from oracle import pywire as W

        off, used = W.scan_frames(blob, big_endian=True)
        assert used == len(blob)
        mm, _, bad = W.wire_decode(blob, off)
        assert bad == 0
        keep = [blob[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1) if int(mm[i]["type"]) not in self.ignored]
        return b"".join(keep)

    def pump(self):
        """deliver until nothing is in flight (upstream: network.send's loop)"""
        for _ in range(400):
            moved = False
            for i in self.live():
                self.node(i).advance()
            for i in self.live():
                for j in range(1, self.n + 1):
                    if j == i:
                        continue
                    blob = self.node(i).poll(j - 1)
                    if not blob:
                        continue
                    moved = True
                    if j in self.nop or (i, j) in self.cuts:
                        continue
                    if self.ignored:
                        blob = self._filter(blob)
                    self.node(j).deliver(blob)
            if not moved:
                return
        raise AssertionError("the network did not quiesce")

    # -- upstream's send(...) of the local message kinds ------------------------------------------------------------
    def hup(self, i):
        self.node(i).campaign(self.groups)
        self.pump()

    def beat(self, i):
        self.node(i).tick()  # heartbeatTick = 1: a tick on a leader is MsgBeat
        self.pump()

    def prop(self, i, data):
        self.node(i).propose_batch(self.groups, [data] * G)
        self.pump()

    def inject(self, to, msgs, ents=None):
        from oracle import pywire as W

        stream, _ = W.wire_encode(msgs, ents)
        self.node(to).deliver(stream.tobytes())
        self.pump()

    # -- what the assertions read -----------------------------------------------------------------------------------
    def status(self, i):
        return self.node(i).statuses()

    def expect(self, i, role=None, term=None, commit=None, log=None):
        st = self.status(i)
        if role is not None:
            assert np.all(st["role"] == role), (i, "role", np.unique(st["role"]))
        if term is not None:
            assert np.all(st["term"] == term), (i, "term", np.unique(st["term"]))
        if commit is not None:
            assert np.all(st["commit"] == commit), (i, "commit", np.unique(st["commit"]))
        if log is not None:
            for g in (0, G // 2, G - 1):
                assert self.node(i).log(g) == log, (i, g, self.node(i).log(g))
            assert np.all(st["last_index"] == len(log))


@pytest.fixture()
def net(gpu_engine_cls):
    made = []

    def make(n, nop=()):
        made.append(Net(n, nop))
        return made[-1]

    yield make
    for m in made:
        m.close()


def test_single_node_candidate_and_commit(net):
    """TestSingleNodeCandidate / TestSingleNodeCommit: one node elects itself; two proposals -> committed = 3."""
    tt = net(1)
    tt.hup(1)
    tt.expect(1, role=LEADER, term=1, commit=1)
    tt.prop(1, b"some data")
    tt.prop(1, b"some data")
    tt.expect(1, commit=3, log=[(1, b""), (1, b"some data"), (1, b"some data")])


def test_log_replication(net):
    """TestLogReplication: (a) one proposal -> committed 2 everywhere; (b) proposal, a second node takes over, another
    proposal -> committed 4 everywhere, and the committed non-empty entries are the proposals in order."""
    tt = net(3)
    tt.hup(1)
    tt.prop(1, b"somedata")
    for i in (1, 2, 3):
        tt.expect(i, commit=2, log=[(1, b""), (1, b"somedata")])
    tt2 = net(3)
    tt2.hup(1)
    tt2.prop(1, b"somedata")
    tt2.hup(2)
    tt2.prop(2, b"somedata2")
    want = [(1, b""), (1, b"somedata"), (2, b""), (2, b"somedata2")]
    for i in (1, 2, 3):
        tt2.expect(i, commit=4, log=want)
        assert [d for d in tt2.node(i).drain(0) if d is not None] == [b"somedata", b"somedata2"]
    tt2.expect(2, role=LEADER, term=2)
    tt2.expect(1, role=FOLLOWER, term=2)


def test_commit_without_new_term_entry(net):
    """TestCommitWithoutNewTermEntry: entries of term 1 that reached only 2 of 5 peers commit once the next leader's own
    entry (becomeLeader's empty one, term 2) is on a quorum -- by counting that entry, they ride along."""
    tt = net(5)
    tt.hup(1)
    tt.cut(1, 3)
    tt.cut(1, 4)
    tt.cut(1, 5)
    tt.prop(1, b"some data")
    tt.prop(1, b"some data")
    tt.expect(1, role=LEADER, commit=1)
    tt.recover()
    tt.hup(2)  # term 2; its empty entry is index 4
    tt.expect(1, commit=4)
    for i in range(1, 6):
        tt.expect(i, commit=4, log=[(1, b""), (1, b"some data"), (1, b"some data"), (2, b"")])


def test_cannot_commit_without_new_term_entry(net):
    """TestCannotCommitWithoutNewTermEntry: the same start, but the new leader's MsgApp are lost: holding a quorum of
    VOTES it must not commit the old term's entries (raft 5.4.2: the term gate of the commit-index advance); after the
    fabric heals, a heartbeat round and one proposal in its own term carry everything: committed = 5."""
    tt = net(5)
    tt.hup(1)
    tt.cut(1, 3)
    tt.cut(1, 4)
    tt.cut(1, 5)
    tt.prop(1, b"some data")
    tt.prop(1, b"some data")
    tt.expect(1, commit=1)
    tt.recover()
    tt.ignore(APP)  # "avoid committing ChangeTerm proposal"
    tt.hup(2)
    tt.expect(2, role=LEADER, term=2, commit=1)  # no log entries from the previous term are committed
    tt.recover()
    tt.beat(2)  # "send heartbeat; reset wait"
    tt.prop(2, b"some data")
    tt.expect(2, commit=5)
    for i in range(1, 6):
        tt.expect(i, log=[(1, b""), (1, b"some data"), (1, b"some data"), (2, b""), (2, b"some data")])


def test_dueling_candidates(net):
    """TestDuelingCandidates: 1 and 3 cannot see each other; 1 wins term 1 with 2's vote, 3 stays a candidate; healed,
    3 campaigns again with an empty log: term 2 everywhere, nobody votes for it -- 1 steps down, 3 gives up on a quorum
    of rejections.  Logs: 1 and 2 keep the term-1 entry (committed), 3's stays empty."""
    tt = net(3)
    tt.cut(1, 3)
    tt.hup(1)
    tt.hup(3)
    tt.expect(1, role=LEADER, term=1)
    tt.expect(3, role=CANDIDATE, term=1)
    tt.recover()
    tt.hup(3)
    tt.expect(1, role=FOLLOWER, term=2, commit=1, log=[(1, b"")])
    tt.expect(2, role=FOLLOWER, term=2, commit=1, log=[(1, b"")])
    tt.expect(3, role=FOLLOWER, term=2, commit=0, log=[])


def test_candidate_concede(net):
    """TestCandidateConcede: an isolated candidate (term 1) meets the leader the others elected in the same term: the
    heartbeat makes it a follower, the next MsgApp brings its log level."""
    tt = net(3)
    tt.isolate(1)
    tt.hup(1)
    tt.hup(3)
    tt.expect(1, role=CANDIDATE, term=1)
    tt.expect(3, role=LEADER, term=1)
    tt.recover()
    tt.beat(3)  # "send heartbeat; reset wait"
    tt.prop(3, b"force follower")  # flushes a MsgApp out to 1
    tt.beat(3)  # "flush out commit"
    tt.expect(1, role=FOLLOWER, term=1)
    for i in (1, 2, 3):
        tt.expect(i, commit=2, log=[(1, b""), (1, b"force follower")])


def test_old_messages(net):
    """TestOldMessages: leadership goes 1 -> 2 -> 1 (terms 1, 2, 3); a MsgApp of term 2 arriving at the term-3 leader
    is ignored; a proposal then commits at index 4."""
    from oracle import pywire as W

    tt = net(3)
    tt.hup(1)
    tt.hup(2)
    tt.hup(1)
    tt.expect(1, role=LEADER, term=3)
    msgs = np.zeros(G, W.WIRE_MSG_DT)
    ents = np.zeros(G, W.WIRE_ENT_DT)
    msgs["group"], msgs["term"], msgs["type"], msgs["from"], msgs["to"] = np.arange(G), 2, APP, 1, 0
    msgs["ent_first"], msgs["n_ents"] = np.arange(G), 1
    ents["term"], ents["index"] = 2, 3
    tt.inject(1, msgs, ents)  # "pretend we're an old leader trying to make progress; this entry is expected to be ignored"
    tt.prop(1, b"somedata")
    for i in (1, 2, 3):
        tt.expect(i, commit=4, log=[(1, b""), (2, b""), (3, b""), (3, b"somedata")])


@pytest.mark.parametrize("n,nop,success", [(3, (), True), (3, (3,), True), (3, (2, 3), False), (4, (2, 3), False),
                                           (5, (2, 3), True)])
def test_proposal(net, n, nop, success):
    """TestProposal: a proposal on node 1 commits exactly when a quorum of real peers exists; without one node 1 stays a
    candidate and the proposal is dropped ("no leader").  Term 1 everywhere."""
    tt = net(n, nop)
    tt.hup(1)
    tt.prop(1, b"somedata")
    want = [(1, b""), (1, b"somedata")] if success else []
    for i in tt.live():
        tt.expect(i, term=1, log=want, commit=2 if success else 0)
    if not success:
        tt.expect(1, role=CANDIDATE)
        assert tt.node(1).stats()["proposals_dropped"] == G


@pytest.mark.parametrize("nop", [(), (3,)])
def test_proposal_by_proxy(net, nop):
    """TestProposalByProxy: a proposal handed to a follower is forwarded to the leader and commits."""
    tt = net(3, nop)
    tt.hup(1)
    tt.prop(2, b"somedata")
    for i in tt.live():
        tt.expect(i, term=1, commit=2, log=[(1, b""), (1, b"somedata")])
    assert tt.node(2).stats()["proposals_dropped"] == 0


def test_campaign_argument_checks(net):
    from raftsql_amd.engine import RaftqError

    tt = net(3)
    with pytest.raises(RaftqError):
        tt.node(1).campaign([0, G])  # refused as a whole
    tt.pump()
    tt.expect(1, role=FOLLOWER, term=0)
    tt.node(1).campaign(np.zeros(0, np.uint64))  # nothing to do
    tt.node(1).campaign([5])
    tt.pump()
    st = tt.status(1)
    assert st["role"][5] == LEADER and np.count_nonzero(st["role"] == LEADER) == 1
    tt.node(1).campaign([5])  # already leader: the MsgHup is ignored
    tt.pump()
    assert tt.status(1)["term"][5] == 1


def test_handle_msg_app_table(gpu_engine_cls):
    """TestHandleMsgApp as recalled: a follower at term 2 holding entries (index 1, term 1), (2, 2) receives one MsgApp;
    the table fixes lastIndex, commit index and the Reject flag of the single MsgAppResp it answers with.  The eleven
    rows run as eleven groups of one node, one batch.  (Upstream calls handleAppendEntries directly, so three rows carry
    Term 1 on a term-2 node; through Step such a message would be dropped as stale -- they are sent with Term 2 here,
    which handleAppendEntries never looks at.)"""
    from oracle import pywire as W
    from raftsql_amd.node import RaftNode

    #        logTerm index commit entries[(index, term)]      wIndex wCommit wReject
    rows = [(3, 2, 3, [], 2, 0, True),   # previous log mismatch
            (3, 3, 3, [], 2, 0, True),   # previous log non-exist
            (1, 1, 1, [], 2, 1, False),
            (0, 0, 1, [(1, 2)], 1, 1, False),
            (2, 2, 3, [(3, 2), (4, 2)], 4, 3, False),
            (2, 2, 4, [(3, 2)], 3, 3, False),
            (1, 1, 4, [(2, 2)], 2, 2, False),
            (1, 1, 3, [], 2, 1, False),          # match entry 1, commit up to last new entry 1
            (1, 1, 3, [(2, 2)], 2, 2, False),    # match entry 1, commit up to last new entry 2
            (2, 2, 3, [], 2, 2, False),          # match entry 2, commit up to last new entry 2
            (2, 2, 4, [], 2, 2, False)]          # commit up to log.last()
    k = len(rows)
    nd = RaftNode(k, 3, 0)
    try:
        for g in range(k):
            nd.replay(g, [(1, b"one"), (2, b"two")])
            nd.set_hard_state(g, 2, 0, 0)  # becomeFollower(2, None)
        nd.start(10, 1, seed=1)
        n_ents = sum(len(r[3]) for r in rows)
        msgs, ents = np.zeros(k, W.WIRE_MSG_DT), np.zeros(n_ents, W.WIRE_ENT_DT)
        pool, at = b"", 0
        for g, (lt, idx, commit, es, *_rest) in enumerate(rows):
            msgs[g] = (g, 2, lt, idx, commit, 0, 1, APP, 0, 0, 0, at, len(es))
            for (ei, et) in es:
                data = b"new-%d-%d" % (g, ei)
                ents[at] = (et, ei, len(pool), len(data), 0)
                pool += data
                at += 1
        stream, _ = W.wire_encode(msgs, ents, pool)
        nd.deliver(stream.tobytes())
        nd.advance()
        st = nd.statuses()
        out = nd.poll(1)
        off, used = W.scan_frames(out, big_endian=True)
        assert used == len(out) and len(off) - 1 == k  # exactly one answer per row
        mm, _, bad = W.wire_decode(out, off)
        assert bad == 0
        by_group = {int(m["group"]): m for m in mm}
        for g, (lt, idx, commit, es, w_index, w_commit, w_reject) in enumerate(rows):
            assert int(st["last_index"][g]) == w_index, (g, "lastIndex", int(st["last_index"][g]))
            assert int(st["commit"][g]) == w_commit, (g, "commit", int(st["commit"][g]))
            m = by_group[g]
            assert int(m["type"]) == APP_RESP and bool(m["reject"]) == w_reject, (g, m)
            if w_reject:
                assert int(m["index"]) == idx and int(m["reject_hint"]) == 2  # Index: m.Index, RejectHint: lastIndex
            else:
                assert int(m["index"]) == idx + len(es)                       # Index: lastnewi
        # the entries themselves: row 3 replaced a conflicting entry 1, rows 4-5 extended the log
        assert nd.log(3) == [(2, b"new-3-1")]
        assert nd.log(4) == [(1, b"one"), (2, b"two"), (2, b"new-4-3"), (2, b"new-4-4")]
        assert nd.log(6) == [(1, b"one"), (2, b"two")]  # entry 2 matched: the log keeps its own copy
    finally:
        nd.destroy()


def test_leader_cycle(net):
    """TestLeaderCycle: each node in turn campaigns and is elected; the others follow."""
    tt = net(3)
    for term, campaigner in enumerate((1, 2, 3), start=1):
        tt.hup(campaigner)
        for i in (1, 2, 3):
            tt.expect(i, role=LEADER if i == campaigner else FOLLOWER, term=term)


def test_leader_sync_follower_log_figure_7(gpu_engine_cls):
    """TestLeaderSyncFollowerLog (raft_paper_test.go; the six follower logs of Figure 7 of the raft paper): a leader whose
    log is (1,1,1,4,4,5,5,6,6,6) by term is elected at term 9 over a follower whose log is missing entries, has extra
    uncommitted entries, or both; after one proposal the follower's log is the leader's.  The six cases are six groups
    of one three-node cluster (the third node swallows everything, as upstream's nopStepper does; its vote is injected).
    What is exercised: the reject / RejectHint back-off of the leader's replication cursor and the follower's
    findConflict-and-truncate -- host logic of raftq_node -- around the batched Step."""
    from oracle import pywire as W
    from raftsql_amd.node import Cluster

    lead_terms = [1, 1, 1, 4, 4, 5, 5, 6, 6, 6]
    followers = [[1, 1, 1, 4, 4, 5, 5, 6, 6],
                 [1, 1, 1, 4],
                 [1, 1, 1, 4, 4, 5, 5, 6, 6, 6, 6],
                 [1, 1, 1, 4, 4, 5, 5, 6, 6, 6, 7, 7],
                 [1, 1, 1, 4, 4, 4, 4],
                 [1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3]]
    k, term = len(followers), 8
    c = Cluster(k, 3, seed=2)
    try:
        for g in range(k):
            # an entry is identified by (index, term): the same payload wherever the two logs agree on both
            c.nodes[0].replay(g, [(t, b"e%d.%d" % (i, t)) for i, t in enumerate(lead_terms, 1)])
            c.nodes[0].set_hard_state(g, term, 0, len(lead_terms))
            c.nodes[1].replay(g, [(t, b"e%d.%d" % (i, t)) for i, t in enumerate(followers[g], 1)])
            c.nodes[1].set_hard_state(g, term - 1, 0, 0)
        c.start()
        live = (0, 1)

        def pump():
            for _ in range(400):
                moved = False
                for p in live:
                    c.nodes[p].advance()
                for p in live:
                    for q in range(3):
                        if q == p:
                            continue
                        blob = c.nodes[p].poll(q)
                        if blob:
                            moved = True
                            if q in live:
                                c.nodes[q].deliver(blob)
                if not moved:
                    return
            raise AssertionError("did not quiesce")

        groups = np.arange(k, dtype=np.uint64)
        c.nodes[0].campaign(groups)
        pump()
        # "The election occurs in the term after the one we loaded": the third member's vote
        v = np.zeros(k, W.WIRE_MSG_DT)
        v["group"], v["term"], v["type"], v["from"], v["to"] = groups, term + 1, VOTE_RESP, 2, 0
        stream, _ = W.wire_encode(v)
        c.nodes[0].deliver(stream.tobytes())
        pump()
        st = c.nodes[0].statuses()
        assert np.all(st["role"] == LEADER) and np.all(st["term"] == term + 1)
        c.nodes[0].propose_batch(groups, [b""] * k)  # upstream proposes an empty entry
        pump()
        want_terms = lead_terms + [term + 1, term + 1]  # + becomeLeader's empty entry + the proposal
        for g in range(k):
            a, b = c.nodes[0].log(g), c.nodes[1].log(g)
            assert [t for t, _ in a] == want_terms, (g, a)
            assert b == a, (g, [t for t, _ in b])
        assert np.all(c.nodes[1].statuses()["commit"] == len(want_terms))
    finally:
        c.close()


def test_follower_tables_of_the_paper_tests(gpu_engine_cls):
    """raft_paper_test.go's follower-side tables as recalled, one group per row, one batch: TestFollowerAppendEntries (4 rows:
    the log after the MsgApp), TestFollowerCheckMsgApp (5 rows: Index / Reject / RejectHint of the answer) and
    TestFollowerCommitEntry (4 rows: the commit index and what reaches the commit channel)."""
    from oracle import pywire as W
    from raftsql_amd.node import RaftNode

    base = [(1, b"e1.1"), (2, b"e2.2")]
    rows = []  # (initial log, hard state (term, commit), msg (term, log_term, index, commit), entries [(index, term, data)], check)
    for index, term, ents, wents in [(2, 2, [(3, 3)], [1, 2, 3]), (1, 1, [(2, 3), (3, 4)], [1, 3, 4]), (0, 0, [(1, 1)], [1, 2]),
                                     (0, 0, [(1, 3)], [3])]:
        rows.append((base, (2, 0), (2, term, index, 0), [(i, t, b"e%d.%d" % (i, t)) for i, t in ents], ("log_terms", wents)))
    for term, index, windex, wreject, whint in [(0, 0, 1, False, 0), (1, 1, 1, False, 0), (2, 2, 2, False, 0), (1, 2, 2, True, 2),
                                                (3, 3, 3, True, 2)]:
        rows.append((base, (2, 1), (2, term, index, 0), [], ("resp", (windex, wreject, whint))))
    for ents, commit in [([b"some data"], 1), ([b"some data", b"some data2"], 2), ([b"some data2", b"some data"], 2),
                         ([b"some data", b"some data2"], 1)]:
        rows.append(([], (1, 0), (1, 0, 0, commit), [(i, 1, d) for i, d in enumerate(ents, 1)], ("commit", (commit, ents[:commit]))))
    k = len(rows)
    nd = RaftNode(k, 3, 0)
    try:
        for g, (log, (hs_term, hs_commit), *_rest) in enumerate(rows):
            if log:
                nd.replay(g, log)
            nd.set_hard_state(g, hs_term, 0, hs_commit)
        nd.start(10, 1, seed=1)
        replayed = [nd.drain(g) for g in range(k)]  # replayWAL's entries and the nil sentinel
        assert all(r[-1] is None for r in replayed)
        msgs = np.zeros(k, W.WIRE_MSG_DT)
        ents = np.zeros(sum(len(r[3]) for r in rows), W.WIRE_ENT_DT)
        pool, at = b"", 0
        for g, (_log, _hs, (term, log_term, index, commit), es, _chk) in enumerate(rows):
            msgs[g] = (g, term, log_term, index, commit, 0, 1, APP, 0, 0, 0, at, len(es))
            for (ei, et, data) in es:
                ents[at] = (et, ei, len(pool), len(data), 0)
                pool += data
                at += 1
        stream, _ = W.wire_encode(msgs, ents, pool)
        nd.deliver(stream.tobytes())
        nd.advance()
        st = nd.statuses()
        out = nd.poll(1)
        off, used = W.scan_frames(out, big_endian=True)
        mm, _, bad = W.wire_decode(out, off)
        assert used == len(out) and bad == 0 and len(mm) == k
        by_group = {int(m["group"]): m for m in mm}
        for g, (*_x, (kind, want)) in enumerate(rows):
            m = by_group[g]
            assert int(m["type"]) == APP_RESP and int(m["term"]) == rows[g][2][0]
            if kind == "log_terms":
                assert [t for t, _ in nd.log(g)] == want, (g, nd.log(g))
                assert not m["reject"]
            elif kind == "resp":
                windex, wreject, whint = want
                assert (int(m["index"]), bool(m["reject"])) == (windex, wreject), (g, m)
                if wreject:
                    assert int(m["reject_hint"]) == whint
            else:
                wcommit, wents = want
                assert int(st["commit"][g]) == wcommit, (g, int(st["commit"][g]))
                assert nd.drain(g) == wents, g  # nextEnts: exactly the committed prefix, in order
    finally:
        nd.destroy()


def test_log_maybe_append_table(gpu_engine_cls):
    """log_test.go's TestLogMaybeAppend as recalled, through handleAppendEntries on a follower holding (1,1), (2,2), (3,3) with
    commit index 1: one group per row, one batch.  wlasti is the Index of the MsgAppResp (lastnewi), wappend = not rejected,
    wcommit the commit index afterwards -- never above lastnewi, never decreasing.  (Upstream's row that conflicts with a
    committed entry panics there and is left out; its (0, 0) row meets handleAppendEntries' `m.Index < committed` answer
    first: Index = committed.)"""
    from oracle import pywire as W
    from raftsql_amd.node import RaftNode

    li, lt, commit = 3, 3, 1
    #        logTerm index committed  ents [(index, term)]          wlasti  wappend wcommit
    rows = [(lt - 1, li, li, [(li + 1, 4)], 0, False, commit),        # not match: term is different
            (lt, li + 1, li, [(li + 2, 4)], 0, False, commit),        # not match: index out of bound
            (lt, li, li, [], li, True, li),                           # match with the last existing entry
            (lt, li, li + 1, [], li, True, li),                       # do not increase commit higher than lastnewi
            (lt, li, li - 1, [], li, True, li - 1),                   # commit up to the commit in the message
            (lt, li, 0, [], li, True, commit),                        # commit do not decrease
            (0, 0, li, [], commit, True, commit),                     # (see the docstring)
            (lt, li, li, [(li + 1, 4)], li + 1, True, li),
            (lt, li, li + 1, [(li + 1, 4)], li + 1, True, li + 1),
            (lt, li, li + 2, [(li + 1, 4)], li + 1, True, li + 1),    # do not increase commit higher than lastnewi
            (lt, li, li + 2, [(li + 1, 4), (li + 2, 4)], li + 2, True, li + 2),
            (lt - 1, li - 1, li, [(li, 4)], li, True, li),            # match with the entry in the middle
            (lt - 2, li - 2, li, [(li - 1, 4)], li - 1, True, li - 1),
            (lt - 2, li - 2, li, [(li - 1, 4), (li, 4)], li, True, li)]
    k = len(rows)
    nd = RaftNode(k, 3, 0)
    try:
        for g in range(k):
            nd.replay(g, [(1, b"e1.1"), (2, b"e2.2"), (3, b"e3.3")])
            nd.set_hard_state(g, 4, 0, commit)
        nd.start(10, 1, seed=1)
        msgs = np.zeros(k, W.WIRE_MSG_DT)
        ents = np.zeros(sum(len(r[3]) for r in rows), W.WIRE_ENT_DT)
        pool, at = b"", 0
        for g, (log_term, index, committed, es, *_w) in enumerate(rows):
            msgs[g] = (g, 4, log_term, index, committed, 0, 1, APP, 0, 0, 0, at, len(es))
            for (ei, et) in es:
                data = b"e%d.%d" % (ei, et)
                ents[at] = (et, ei, len(pool), len(data), 0)
                pool += data
                at += 1
        stream, _ = W.wire_encode(msgs, ents, pool)
        nd.deliver(stream.tobytes())
        nd.advance()
        st = nd.statuses()
        out = nd.poll(1)
        off, used = W.scan_frames(out, big_endian=True)
        mm, _, bad = W.wire_decode(out, off)
        assert used == len(out) and bad == 0 and len(mm) == k
        by_group = {int(m["group"]): m for m in mm}
        for g, (log_term, index, committed, es, wlasti, wappend, wcommit) in enumerate(rows):
            m = by_group[g]
            assert bool(m["reject"]) == (not wappend), (g, m)
            if wappend:
                assert int(m["index"]) == wlasti, (g, int(m["index"]))
                want_log = [1, 2, 3][:min(3, es[0][0] - 1) if es else 3] + [t for _, t in es]
                assert [t for t, _ in nd.log(g)] == want_log, (g, nd.log(g))
            else:
                assert int(m["index"]) == index and int(m["reject_hint"]) == li
                assert [t for t, _ in nd.log(g)] == [1, 2, 3]
            assert int(st["commit"][g]) == wcommit, (g, int(st["commit"][g]))
    finally:
        nd.destroy()


def test_leader_commit_preceding_entries(gpu_engine_cls):
    """TestLeaderCommitPrecedingEntries (raft_paper_test.go): when a leader commits an entry of its own term, all preceding
    entries -- including those of earlier leaders -- are committed with it.  Four starting logs (empty; (2,1); (1,1),(2,2);
    (1,1)) as four groups of a three-node cluster whose members all hold the same log at term 2; node 1 is elected (term 3)
    and proposes once: every node ends with log + (3, empty) + (3, "some data"), all of it committed."""
    from raftsql_amd.node import Cluster

    logs = [[], [2], [1, 2], [1]]
    k = len(logs)
    c = Cluster(k, 3, seed=6)
    try:
        for g, terms in enumerate(logs):
            for nd in c.nodes:
                if terms:
                    nd.replay(g, [(t, b"e%d.%d" % (i, t)) for i, t in enumerate(terms, 1)])
                nd.set_hard_state(g, 2, 0, 0)
        c.start()
        groups = np.arange(k, dtype=np.uint64)
        c.nodes[0].campaign(groups)
        c.settle()
        assert np.all(c.nodes[0].statuses()["role"] == LEADER) and np.all(c.nodes[0].statuses()["term"] == 3)
        c.nodes[0].propose_batch(groups, [b"some data"] * k)
        c.settle()
        c.nodes[0].tick()  # a heartbeat carries the commit index to the followers
        c.settle()
        for g, terms in enumerate(logs):
            want = [(t, b"e%d.%d" % (i, t)) for i, t in enumerate(terms, 1)] + [(3, b""), (3, b"some data")]
            for nd in c.nodes:
                assert nd.log(g) == want, (g, nd.log(g))
                assert int(nd.statuses()["commit"][g]) == len(want)
    finally:
        c.close()


def test_follower_start_election_and_leader_bcast_beat(gpu_engine_cls):
    """TestFollowerStartElection: after 2 x electionTimeout - 1 ticks without hearing from a leader a follower has
    campaigned exactly once -- term + 1, candidate, voted for itself, one MsgVote to each peer carrying that term.
    TestLeaderBcastBeat: a leader sends one MsgHeartbeat to each follower at every heartbeat tick, entries or not."""
    from oracle import pywire as W
    from raftsql_amd.node import RaftNode

    k, et = 64, 10
    nd = RaftNode(k, 3, 0)
    try:
        nd.start(et, 1, seed=99)
        out = {1: b"", 2: b""}
        for _ in range(2 * et - 1):
            nd.tick()
            nd.advance()
            for q in (1, 2):
                out[q] += nd.poll(q)
        st = nd.statuses()
        assert np.all(st["role"] == CANDIDATE) and np.all(st["term"] == 1) and np.all(st["vote"] == 1)
        for q in (1, 2):
            off, used = W.scan_frames(out[q], big_endian=True)
            mm, _, bad = W.wire_decode(out[q], off)
            assert used == len(out[q]) and bad == 0
            assert len(mm) == k and np.all(mm["type"] == VOTE) and np.all(mm["term"] == 1) and np.all(mm["to"] == q)
            assert sorted(mm["group"].tolist()) == list(range(k))
        # both peers grant: leader of every group; then every tick is a heartbeat round
        v = np.zeros(2 * k, W.WIRE_MSG_DT)
        v["group"], v["term"], v["type"], v["to"] = np.tile(np.arange(k), 2), 1, VOTE_RESP, 0
        v["from"] = np.repeat([1, 2], k)
        stream, _ = W.wire_encode(v)
        nd.deliver(stream.tobytes())
        nd.advance()
        assert np.all(nd.statuses()["role"] == LEADER)
        for q in (1, 2):
            nd.poll(q)  # becomeLeader's MsgApp
        for _ in range(3):
            nd.tick()
            nd.advance()
            for q in (1, 2):
                blob = nd.poll(q)
                off, used = W.scan_frames(blob, big_endian=True)
                mm, _, bad = W.wire_decode(blob, off)
                assert bad == 0 and len(mm) == k and np.all(mm["type"] == HEARTBEAT) and np.all(mm["n_ents"] == 0)
                assert np.all(mm["term"] == 1) and np.all(mm["commit"] == 0)  # min(match[q] = 0, committed)
    finally:
        nd.destroy()
