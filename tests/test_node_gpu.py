"""GPU: multi-node raft clusters of G groups, after the reference's own integration tests
(raftsql_test.go:11-171).  There: 3 real nodes in one process over loopback TCP, one group.
Here: 3 (or 5) real nodes in one process over an in-memory transport, G groups each; every
election, vote, heartbeat and commit decision is the batched GPU Step / Tick.  What is
asserted is what the reference asserts -- statements become visible on every node, in order;
a stopped node does not block a quorum; a restarted node replays exactly its log and then
the nil sentinel -- plus the raft safety properties the reference takes on faith."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def Cluster(gpu_engine_cls):
    from raftsql_amd.node import Cluster as C

    return C


def elect(c, max_ticks=400):
    """Tick until every group has a leader.  The bound is liveness, not speed: with etcd's rule (a fresh draw on every tick
    past the base timeout) two of three timers fire on the same tick in roughly a third of the rounds, a round is ~12 ticks,
    and the suite elects thousands of groups -- a 60-tick bound passes or fails with the stream's luck (round 6 redefined
    the stream and ten tests that had only ever seen the old one ran out of ticks)."""
    max_ticks = max(max_ticks, 400)
    for _ in range(max_ticks):
        c.step(tick=True)
        if np.all(c.leaders() >= 0):
            c.settle()
            return
    raise AssertionError(f"no leader for groups {np.nonzero(c.leaders() < 0)[0][:10]} after {max_ticks} ticks")


def check_safety(c):
    """Election safety + log matching + leader completeness over the live nodes."""
    live = [nd for p, nd in enumerate(c.nodes) if p not in c.down]
    for g in range(c.G):
        sts = [nd.status(g) for nd in live]
        by_term = {}
        for st in sts:
            if st.role == 2:
                assert by_term.setdefault(st.term, 0) == 0, f"two leaders in term {st.term} of group {g}"
                by_term[st.term] += 1
        logs = [nd.log(g) for nd in live]
        commit = min(st.commit for st in sts)
        for lg in logs[1:]:
            assert lg[:commit] == logs[0][:commit], f"committed prefixes differ in group {g}"
        for a in logs:
            assert all(a[i][0] <= a[i + 1][0] for i in range(len(a) - 1)), "log terms must not decrease"


def test_new_db_analog_three_nodes(Cluster):
    """TestNewDB (raftsql_test.go:92-115): CREATE on node 0, one INSERT per node; every node
    sees all four statements, in the same order."""
    c = Cluster(6, 3)
    try:
        c.start()
        for nd in c.nodes:
            for g in range(c.G):
                assert nd.drain(g) == [None]  # fresh WAL: only the nil sentinel (raft.go:131-132)
        elect(c)
        lead = c.leaders()
        assert np.all(lead >= 0)
        g = 2
        c.nodes[0].propose(g, b"CREATE TABLE t (id int primary key, v int)")
        c.settle()
        for i, nd in enumerate(c.nodes):  # followers forward to the leader (MsgProp)
            nd.propose(g, b"INSERT INTO t (v) VALUES (%d)" % i)
            c.settle()
        want = [b"CREATE TABLE t (id int primary key, v int)"] + [b"INSERT INTO t (v) VALUES (%d)" % i for i in range(3)]
        for nd in c.nodes:
            assert nd.drain(g) == want
            for other in range(c.G):
                if other != g:
                    assert nd.drain(other) == []  # "main.x does not exist" on every node
        check_safety(c)
    finally:
        c.close()


def test_restart_db_analog(Cluster):
    """TestRestartDB (raftsql_test.go:117-171): stop a node; 2 of 3 still commit; the restarted
    node replays exactly its 4 logged statements before the nil sentinel, does not have `foo`
    right after replay, and gets it once the leader has caught it up."""
    c = Cluster(3, 3)
    try:
        c.start()
        elect(c)
        g = 1
        stmts = [b"CREATE TABLE t (id int primary key, v int)"] + [b"INSERT INTO t (v) VALUES (%d)" % i for i in range(3)]
        for s in stmts:
            c.nodes[int(c.leaders()[g])].propose(g, s)
            c.settle()
        for nd in c.nodes:
            assert [d for d in nd.drain(g) if d is not None] == stmts
        victim = (int(c.leaders()[g]) + 1) % 3  # a follower of group g
        logs = c.stop(victim)
        assert [d for _, d in logs[g] if d] == stmts
        elect(c)  # groups the victim led need a new leader; 2 of 3 is a quorum
        c.nodes[int(c.leaders()[g])].propose(g, b"INSERT INTO t (v) VALUES ('foo')")
        c.settle()
        for p, nd in enumerate(c.nodes):
            if p != victim:
                assert nd.drain(g) == [b"INSERT INTO t (v) VALUES ('foo')"]
        nd = c.restart(victim, logs)
        replayed = nd.drain(g)
        assert replayed[-1] is None and replayed[:-1] == stmts and len(replayed) == 5  # exactly 4, then nil
        assert nd.status(g).term == 0  # HardState is not restored by replay (raft.go:124)
        c.run(5)
        c.settle()
        assert nd.drain(g) == [b"INSERT INTO t (v) VALUES ('foo')"]
        check_safety(c)
    finally:
        c.close()


def test_many_groups_elect_and_commit(Cluster):
    """2000 groups x 3 nodes: every group elects exactly one leader, leadership spreads over the
    nodes (randomised timeouts), and a proposal per group commits on every node."""
    G = 2000
    c = Cluster(G, 3, seed=11)
    try:
        c.start()
        elect(c, max_ticks=80)
        lead = c.leaders()
        counts = np.bincount(lead, minlength=3)
        assert counts.sum() == G and counts.min() > G // 10, counts
        for g in range(G):
            c.nodes[int(lead[g])].propose(g, b"v%d" % g)
        c.settle()
        c.run(2)  # a heartbeat round carries the commit index to the followers
        c.settle()
        for p, nd in enumerate(c.nodes):
            got = [nd.drain(g) for g in range(G)]
            assert all(got[g] == [None, b"v%d" % g] for g in range(G)), p
        st = c.nodes[0].stats()
        assert st["msgs_stepped"] > 3 * G and st["proposals_dropped"] == 0
        check_safety(c)
    finally:
        c.close()


def test_partitioned_leader_cannot_commit_and_rejoins(Cluster):
    """5 nodes: cut the leader of a group off.  It keeps accepting proposals but commits nothing;
    the majority side elects a new leader and commits; after healing, the old leader's
    uncommitted entry is overwritten and every node delivers the same sequence."""
    c = Cluster(4, 5, seed=3)
    try:
        c.start()
        elect(c)
        g = 0
        old = int(c.leaders()[g])
        c.nodes[old].propose(g, b"a")
        c.settle()
        for q in range(5):
            if q != old:
                c.cut.add((old, q))
        c.nodes[old].propose(g, b"lost")  # reaches nobody
        c.run(3)
        assert c.nodes[old].status(g).commit == c.nodes[old].status(g).last_index - 1
        for _ in range(400):
            c.step()
            l2 = [p for p in range(5) if p != old and c.nodes[p].status(g).role == 2]
            if l2:
                break
        assert l2, "majority side elected no leader"
        new = l2[0]
        assert c.nodes[new].status(g).term > c.nodes[old].status(g).term
        c.nodes[new].propose(g, b"b")
        c.settle()
        c.run(2)
        c.settle()
        c.cut.clear()
        c.run(6)
        c.settle()
        assert c.nodes[old].status(g).role == 0
        seqs = [[d for d in nd.drain(g) if d is not None] for nd in c.nodes]
        assert all(s == [b"a", b"b"] for s in seqs), seqs
        assert b"lost" not in [d for _, d in c.nodes[old].log(g)]
        check_safety(c)
    finally:
        c.close()


def test_restart_with_an_uncommitted_tail_that_gets_overwritten(Cluster):
    """A node restarts with logged-but-uncommitted entries: replayWAL publishes them all (raft.go:122-134).  When
    the new leader overwrites that tail, the REPLACEMENT entries at the same indices must still reach the commit
    channel once they commit (`applied` stayed beyond the truncation point and they were lost)."""
    c = Cluster(2, 5, seed=11)
    try:
        c.start()
        elect(c)
        g = 1
        old = int(c.leaders()[g])
        c.nodes[old].propose(g, b"a")
        c.settle()
        for q in range(5):
            if q != old:
                c.cut.add((old, q))
        c.nodes[old].propose(g, b"lost1")  # two entries that reach nobody
        c.nodes[old].propose(g, b"lost2")
        c.run(2)
        st = c.nodes[old].status(g)
        assert st.commit == st.last_index - 2
        logs = c.stop(old)
        assert [d for _, d in logs[g] if d] == [b"a", b"lost1", b"lost2"]
        for _ in range(400):
            c.step()
            l2 = [p for p in range(5) if p != old and c.nodes[p].status(g).role == 2]
            if l2:
                break
        assert l2, "majority side elected no leader"
        new = l2[0]
        c.nodes[new].propose(g, b"b")  # lands on an index the old leader filled with lost1 / lost2
        c.settle()
        c.nodes[new].propose(g, b"c")
        c.settle()
        assert c.nodes[new].status(g).last_index == st.last_index + 1  # noop(t2), b, c over lost1, lost2
        c.cut.clear()
        nd = c.restart(old, logs, restore_hard_state=True)
        assert nd.drain(g) == [b"a", b"lost1", b"lost2", None]  # the replay, exactly as the reference does it
        c.run(8)
        c.settle()
        assert nd.drain(g) == [b"b", b"c"], "replacement entries below the replay cursor were not published"
        assert [d for _, d in nd.log(g) if d] == [b"a", b"b", b"c"]
        check_safety(c)
    finally:
        c.close()


@pytest.mark.parametrize("seed,from_wal,crank", [(1, False, False), (2, False, False), (3, False, False), (4, True, False),
                                                 (5, True, False), (6, False, True), (7, False, True)])
def test_chaos_safety_and_convergence(Cluster, seed, from_wal, crank):
    """Random message loss, partitions, stops and restarts (WAL + HardState restored; from_wal: from the
    node's own WAL bytes -- raftq_node_replay_wal -- and nothing else) while clients
    keep proposing on whatever node they reach.  Throughout: at most one leader per term, committed
    prefixes agree, and what the never-restarted nodes delivered are prefixes of one sequence per
    group.  After healing: all nodes hold the same committed sequence, no payload twice, nothing
    invented, and every live stream is exactly that sequence.  crank: the nodes' turns and the transport on the library's
    own threads (raftq_crank_step), stopped nodes and lost transfers included."""
    rng = np.random.default_rng(seed)
    G, N = 24, 5
    c = Cluster(G, N, seed=seed, wal=from_wal, threads=crank, native_transport=crank)

    def restart(p):
        logs = stopped.pop(p)
        if from_wal:
            c.restart_from_wal(p, restore_hard_state=True)
        else:
            c.restart(p, logs, restore_hard_state=True)

    # delivered[p][g]: the live commit stream of node p; None once p was restarted (a replay
    # re-publishes the whole WAL, uncommitted tail included -- the reference's contract, raft.go:129-132)
    delivered = [[[] for _ in range(G)] for _ in range(N)]
    proposed = set()

    def collect():
        for p, nd in enumerate(c.nodes):
            if p in c.down or delivered[p] is None:
                continue
            for g in range(G):
                delivered[p][g] += [d for d in nd.drain(g) if d is not None]
        for g in range(G):
            streams = [delivered[p][g] for p in range(N) if delivered[p] is not None]
            if not streams:  # every node has been restarted at least once: no live stream left to compare
                continue
            longest = max(streams, key=len)
            for st in streams:
                assert st == longest[: len(st)], f"group {g}: delivered streams diverged"

    try:
        c.start()
        elect(c, max_ticks=150)
        stopped = {}
        k = 0
        for it in range(160):
            c.loss = 0.25 if (it // 20) % 2 == 0 else 0.0
            r = rng.random()
            if r < 0.04 and len(c.down) < N // 2:
                p = int(rng.choice([q for q in range(N) if q not in c.down]))
                collect()
                stopped[p] = c.stop(p)
                delivered[p] = None
            elif r < 0.10 and c.down:
                p = int(rng.choice(sorted(c.down)))
                restart(p)
            elif r < 0.14:
                a, b = rng.choice(N, 2, replace=False)
                c.cut.add((int(a), int(b)))
            elif r < 0.20:
                c.cut.clear()
            live = [q for q in range(N) if q not in c.down]
            for _ in range(int(rng.integers(0, 6))):
                payload = b"op%d" % k
                k += 1
                proposed.add(payload)
                c.nodes[int(rng.choice(live))].propose(int(rng.integers(0, G)), payload)
            c.step(tick=True)
            if it % 10 == 0:
                check_safety(c)
                collect()
        # heal everything and let the cluster converge
        c.loss = 0.0
        c.cut.clear()
        for p in sorted(c.down):
            restart(p)
        elect(c, max_ticks=150)
        # elect() is satisfied by a leader of an older term while a newer election is still running: keep cranking
        # until every group's commit index is the same everywhere (bounded: convergence is the claim, not its speed)
        for _ in range(60):
            c.run(5)
            c.settle()
            if all(len({int(nd.status(g).commit) for nd in c.nodes}) == 1 for g in range(G)):
                break
        check_safety(c)
        collect()
        n_committed = 0
        for g in range(G):
            commits = {int(nd.status(g).commit) for nd in c.nodes}
            assert len(commits) == 1, (g, commits)
            commit = commits.pop()
            seqs = [[d for _, d in nd.log(g)[:commit] if d] for nd in c.nodes]
            assert all(sq == seqs[0] for sq in seqs), g
            assert len(set(seqs[0])) == len(seqs[0]) and set(seqs[0]) <= proposed, g
            n_committed += len(seqs[0])
            for p in range(N):
                if delivered[p] is not None:
                    assert delivered[p][g] == seqs[0], (g, p)
        assert n_committed > k // 4, (n_committed, k)  # the cluster made real progress under chaos
        if from_wal:  # every node's disk is still one valid segment holding exactly its logs
            for p, nd in enumerate(c.nodes):
                _, logs, _ = _wal_view(bytes(c.wal[p]), G)
                for g in range(G):
                    assert logs[g] == nd.log(g), (p, g)
    finally:
        c.close()


def test_single_node_cluster_commits_immediately(Cluster):
    c = Cluster(5, 1)
    try:
        c.start()
        elect(c, max_ticks=30)
        for g in range(5):
            c.nodes[0].propose(g, b"x%d" % g)
        c.step(tick=False)
        for g in range(5):
            assert c.nodes[0].drain(g) == [None, b"x%d" % g]
    finally:
        c.close()


def test_shards_of_one_node_are_turned_at_once(Cluster):
    """raftq_shards_create / raftq_shards_turn: K handles that are the same peer slot of the same cluster for K disjoint sets of
    groups, turned all at once on the library's threads.  Here the cluster has one peer (every group elects itself and commits
    on its own), so a shard is complete without a transport: what each shard publishes is what was proposed to IT, in order; a
    set refuses handles of different clusters, the same handle twice, and the cluster crank's step."""
    import ctypes as C

    from raftsql_amd.node import RaftNode, RaftqError, Shards, _load

    sizes = (5, 3, 8)
    nodes = [RaftNode(g, 1, 0) for g in sizes]
    other = RaftNode(4, 3, 0)
    try:
        for nd in nodes:
            nd.start(10, 1, seed=3)
        with pytest.raises(RaftqError):
            Shards([nodes[0], other])  # not the same cluster
        with pytest.raises(RaftqError):
            Shards([nodes[0], nodes[0]])  # a handle is turned by one thread at a time
        with Shards(nodes) as sh:
            for _ in range(40):  # elections: every group's timer fires within 2 x election_tick ticks
                sh.turn(tick=True)
                if all((nd.roles() == 2).all() for nd in nodes):
                    break
            assert all((nd.roles() == 2).all() for nd in nodes)
            sh.turn()
            for k, nd in enumerate(nodes):
                for g in range(sizes[k]):
                    for j in range(k + 1):  # shard k: k + 1 statements per group
                        nd.propose(g, b"s%d g%d #%d" % (k, g, j))
            pub = sh.turn()
            assert pub.tolist() == [sizes[k] * (k + 1) for k in range(len(sizes))]
            assert sh.turn().tolist() == [0, 0, 0]
            for k, nd in enumerate(nodes):
                for g in range(sizes[k]):
                    assert nd.drain(g) == [None] + [b"s%d g%d #%d" % (k, g, j) for j in range(k + 1)]
            # the cluster crank's step is not for shards (there is no transport between them)
            assert _load().raftq_crank_step(sh._p, 1, 0, None, 0, None, None) != 0
    finally:
        for nd in nodes + [other]:
            nd.destroy()


def test_node_rejects_garbage_frames_and_bad_calls(Cluster):
    from raftsql_amd.engine import RaftqError

    c = Cluster(2, 3)
    try:
        nd = c.nodes[0]
        with pytest.raises(RaftqError):
            nd.propose(0, b"too early")  # not started
        c.start()
        for bad in (b"\x00" * 10, b"\xff" * 64, b"\x00" * 56 + b"\x05\x00\x00\x00\x00\x00\x00\x00"):
            with pytest.raises(RaftqError):
                nd.deliver(bad)
        with pytest.raises(RaftqError):
            nd.propose(2, b"no such group")
        assert nd.advance() == 0
        assert nd.close() == 0
        assert nd.recv(0)[0] in (1, 2)  # the sentinel is still queued, then CLOSED
    finally:
        c.close()


# ---- the byte formats on the node's two outward faces (SURVEY 8f-4 wired into 8f-2) ----------------

def _tap(c):
    """wrap every node's poll() so that the frames crossing the transport are recorded"""
    seen = []
    for p, nd in enumerate(c.nodes):
        orig = nd.poll

        def poll(q, _orig=orig, _p=p):
            b = _orig(q)
            if b:
                seen.append((_p, q, b))
            return b

        nd.poll = poll
    return seen


def test_node_speaks_raftpb_frames(Cluster):
    """What leaves a node is what rc.transport.Send puts on a rafthttp stream (raft.go:230): u64
    big-endian length | raftpb.Message.  The google.protobuf runtime parses every frame; sender,
    addressee and group are right; a proposal travels as MsgApp entries with Index and Term set."""
    from oracle import pywire as W
    from tests import pbschema as P

    c = Cluster(4, 3)
    try:
        seen = _tap(c)
        c.start()
        elect(c)
        g = 3
        lead = int(c.leaders()[g])
        c.nodes[(lead + 1) % 3].propose(g, b"INSERT INTO t (v) VALUES (42)")  # forwarded: MsgProp on the wire
        c.settle()
        c.run(2)
        c.settle()
        for nd in c.nodes:
            assert [d for d in nd.drain(g) if d is not None] == [b"INSERT INTO t (v) VALUES (42)"]
        Msg = P.classes()["Message"]
        kinds, app_entries, props = set(), [], []
        for src, dst, blob in seen:
            off, used = W.scan_frames(blob, big_endian=True)
            assert used == len(blob) and len(off) > 1
            mm, ee, bad = W.wire_decode(blob, off)
            assert bad == 0
            for i in range(len(off) - 1):
                pm = Msg()
                pm.ParseFromString(blob[int(off[i]) + 8:int(off[i + 1])])
                assert getattr(pm, "from") == src + 1 and pm.to == dst + 1 and pm.group < c.G
                assert (mm[i]["from"], mm[i]["to"], mm[i]["group"], mm[i]["type"]) == (src, dst, pm.group, pm.type)
                assert pm.HasField("snapshot") and not pm.snapshot.data  # nullable=false: always there, empty
                kinds.add(pm.type)
                if pm.type == 3 and pm.group == g:
                    for k, e in enumerate(pm.entries):
                        assert e.Index == pm.index + 1 + k and e.Term >= 1
                        app_entries.append(bytes(e.Data))
                if pm.type == 2:
                    props += [bytes(e.Data) for e in pm.entries]
        assert {2, 3, 4, 5, 6, 8, 9} <= kinds, kinds  # MsgProp, MsgApp(+Resp), MsgVote(+Resp), MsgHeartbeat(+Resp)
        assert props == [b"INSERT INTO t (v) VALUES (42)"]
        assert b"INSERT INTO t (v) VALUES (42)" in app_entries and b"" in app_entries  # and the new leader's no-op
        check_safety(c)
    finally:
        c.close()


@pytest.mark.parametrize("N,per_turn,interleave", [(3, 1, False), (5, 3, False), (3, 3, True)])
def test_device_built_msgapps_are_the_hosts_byte_for_byte(Cluster, N, per_turn, interleave, monkeypatch):
    """a leader's appendEntry + bcastAppend for what it is asked to propose run on the device
    (raftq_propose_frames: the MsgApp headers are written into the encoder's input in HBM and never exist on the host).  The same
    scripted cluster -- elections, several statements per group and turn, statements proposed on followers too (forwarded as
    MsgProp: the host's way), three WALs -- run once with RAFTQ_NODE_PROPOSE_DEVICE=0 (round 5: handle_proposal + bcast_append
    on the host) and once with the device path: every batch of frames that crosses the transport, every WAL byte and every
    commit channel are IDENTICAL, byte for byte, and the device path was really taken."""
    G = 24

    def run(device_path):
        monkeypatch.setenv("RAFTQ_NODE_PROPOSE_DEVICE", "1" if device_path else "0")
        c = Cluster(G, N, wal=True, seed=11)
        try:
            seen = _tap(c)
            c.start()
            elect(c)
            lead = c.leaders().copy()
            for wave in range(6):
                # a group's statements of a turn next to each other in its leader's queue (one pass over the queue), or -- interleave --
                # spread between other groups' (the general, two-pass shape): one MsgApp per group and peer either way
                order = [(g, k) for g in range(G) for k in range(per_turn if g % 3 else 1)]
                if interleave:
                    order.sort(key=lambda gk: (gk[1], gk[0]))
                for g, k in order:
                    proposer = int(lead[g]) if (g + wave) % 4 else (int(lead[g]) + 1) % N  # every fourth: through a follower
                    c.nodes[proposer].propose(g, b"INSERT INTO t (v) VALUES (%d) -- g%d w%d" % (k, g, wave))
                c.step()
            c.settle()
            c.run(2)
            c.settle()
            assert (c.leaders() == lead).all()
            chans = [[nd.drain(g) for g in range(G)] for nd in c.nodes]
            check_safety(c)
            built = sum(nd.stats()["msgs_built_on_device"] for nd in c.nodes)
            # every statement proposed ON its group's leader went out as device-built MsgApps: 3 of 4 waves per group
            assert built == (0 if not device_path else (N - 1) * sum(1 for wave in range(6) for g in range(G) if (g + wave) % 4))
            return [(a, b, bytes(blob)) for a, b, blob in seen], [bytes(w) for w in c.wal], chans
        finally:
            c.close()

    frames_h, wal_h, chans_h = run(False)
    frames_d, wal_d, chans_d = run(True)
    assert chans_h == chans_d
    assert len(frames_h) == len(frames_d)
    for (a, b, x), (a2, b2, y) in zip(frames_h, frames_d):
        assert (a, b) == (a2, b2) and x == y, "a batch of frames from node %d to node %d differs" % (a, b)
    assert wal_h == wal_d
    assert sum(len(x) for _, _, x in frames_d) > 6 * G * 60  # (the statements did travel)


def test_node_drops_frames_that_are_not_for_it(Cluster):
    """Well-framed bytes whose message does not parse, is addressed elsewhere, comes from no peer,
    names no group or is of a kind a peer never sends are counted and dropped; the node lives on."""
    from oracle import pywire as W

    c = Cluster(2, 3)
    try:
        c.start()
        elect(c)
        nd = c.nodes[0]
        m = np.zeros(6, W.WIRE_MSG_DT)
        m["type"], m["term"], m["from"], m["to"] = 4, 1, 1, 0
        m["to"][0] = 1  # for node 1
        m["from"][1] = 7  # no such peer
        m["group"][2] = 99  # no such group
        m["type"][3] = 7  # MsgSnap
        m["type"][4] = 0  # MsgHup from the wire
        good, _ = W.wire_encode(m[5:6])  # a stale MsgAppResp: harmless, but it is stepped
        bad, off = W.wire_encode(m[:5])
        junk = (5).to_bytes(8, "big") + b"\x0b\x0b\x0b\x0b\x0b"  # framed, does not parse
        before = nd.stats()
        nd.deliver(bytes(bad) + junk + bytes(good))
        nd.advance()
        after = nd.stats()
        assert after["frames_dropped"] - before["frames_dropped"] == 6
        assert after["msgs_stepped"] - before["msgs_stepped"] == 1
        c.nodes[int(c.leaders()[1])].propose(1, b"still alive")
        c.settle()
        c.run(2)
        c.settle()
        assert [d for d in nd.drain(1) if d is not None] == [b"still alive"]
    finally:
        c.close()


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_a_groups_messages_are_worked_off_in_arrival_order(Cluster, fuse, monkeypatch):
    """One turn of a leader receives, for one group: a forwarded MsgProp, a MsgHeartbeat of a HIGHER term from another peer, a
    second forwarded MsgProp.  In arrival order the first proposal is appended by the leader, the heartbeat makes it a follower
    of the sender, and the second proposal is forwarded to that new leader -- whichever way the round was stepped
    (RAFTQ_NODE_FUSE_INBOUND: one submission, where the heartbeat is deferred behind the held proposal, or the staged rounds).
    ADVICE r04: the fused round used to run the second proposal first, appending it in the old term."""
    from oracle import pywire as W

    monkeypatch.setenv("RAFTQ_NODE_FUSE_INBOUND", fuse)
    c = Cluster(2, 3)
    try:
        c.start()
        elect(c)
        g = 1
        lead = int(c.leaders()[g])
        a, b = [p for p in range(3) if p != lead]
        nd = c.nodes[lead]
        st0 = nd.status(g)
        log0 = nd.log(g)
        e1, e2 = b"INSERT INTO t (v) VALUES (1)", b"INSERT INTO t (v) VALUES (2)"
        m = np.zeros(3, W.WIRE_MSG_DT)
        m["group"], m["to"] = g, lead
        m["type"] = [2, 8, 2]  # MsgProp, MsgHeartbeat, MsgProp
        m["from"] = [a, b, a]
        m["term"] = [0, st0.term + 1, 0]
        m["n_ents"] = [1, 0, 1]
        m["ent_first"] = [0, 0, 1]
        ents = np.zeros(2, W.WIRE_ENT_DT)
        ents["data_len"] = [len(e1), len(e2)]
        ents["data_off"] = [0, len(e1)]
        frames, _ = W.wire_encode(m, ents, np.frombuffer(e1 + e2, np.uint8))
        for q in (a, b):
            nd.forward(q, None)  # whatever the election left queued
        nd.deliver(bytes(frames))
        nd.advance()
        st = nd.status(g)
        assert (st.role, st.term, st.lead) == (0, st0.term + 1, b + 1), (st.role, st.term, st.lead)
        log1 = nd.log(g)
        assert log1[: len(log0)] == log0 and [d for _, d in log1[len(log0):]] == [e1], log1[len(log0):]
        assert log1[-1][0] == st0.term  # appended by the old leader, in its term, before it heard of the new one
        out_b = nd.poll(b)
        off, used = W.scan_frames(out_b, big_endian=True)
        assert used == len(out_b)
        mm, ee, bad = W.wire_decode(out_b, off)
        assert bad == 0
        kinds = [int(t) for t in mm["type"][mm["group"] == g]]
        assert 9 in kinds and 2 in kinds, kinds  # MsgHeartbeatResp, and the second proposal forwarded as MsgProp
        prop = mm[(mm["group"] == g) & (mm["type"] == 2)][0]
        assert int(prop["n_ents"]) == 1
        ent = ee[int(prop["ent_first"])]
        assert bytes(out_b[int(ent["data_off"]): int(ent["data_off"]) + int(ent["data_len"])]) == e2
    finally:
        c.close()


def _wal_view(wal: bytes, G: int):
    """what a WAL holds, by the oracle: per-group log (the later record of an index wins) and last HardState"""
    from oracle import pywire as W

    off, used = W.scan_frames(wal, big_endian=False)
    assert used == len(wal)
    recs, n_valid, last = W.wal_decode(wal, off, 0)
    assert n_valid == len(recs), "the WAL's CRC chain must hold from its first byte"
    logs = [[] for _ in range(G)]
    hs = [None] * G
    for r in recs:
        g = int(r["group"])
        if r["kind"] == W.WAL_ENTRY:
            i = int(r["index"])
            assert 1 <= i <= len(logs[g]) + 1
            del logs[g][i - 1:]
            a = int(r["data_off"])
            logs[g].append((int(r["term"]), bytes(wal[a:a + int(r["data_len"])])))
        elif r["kind"] == W.WAL_STATE:
            hs[g] = (int(r["term"]), int(r["vote"]), int(r["index"]))
    return recs, logs, hs


def test_node_wal_is_a_valid_segment_and_restarts_the_node(Cluster):
    """TestRestartDB with a real WAL: every node's WAL bytes are one valid segment (wal.Create's head,
    then per turn every touched group's entries and HardState, one CRC-32C chain); they hold exactly the
    node's logs and HardStates; a node restarted from those bytes alone replays exactly its statements,
    then the nil sentinel, and keeps appending to the same chain."""
    from oracle import pywire as W

    G = 5
    c = Cluster(G, 3, wal=True)
    try:
        c.start()
        elect(c)
        stmts = [b"CREATE TABLE t (id int primary key, v int)"] + [b"INSERT INTO t (v) VALUES (%d)" % i for i in range(3)]
        for s in stmts:
            for g in range(G):
                c.nodes[int(c.leaders()[g])].propose(g, s + b" -- g%d" % g)
            c.settle()
        c.run(2)
        c.settle()
        for p, nd in enumerate(c.nodes):
            recs, logs, hs = _wal_view(bytes(c.wal[p]), G)
            assert list(recs["kind"][:3]) == [W.WAL_CRC, W.WAL_METADATA, W.WAL_SNAPSHOT]
            assert nd.stats()["wal_records"] == len(recs)
            for g in range(G):
                st = nd.status(g)
                assert logs[g] == nd.log(g), (p, g)
                assert hs[g] == (st.term, st.vote, st.commit), (p, g, hs[g])
        g = 1
        victim = (int(c.leaders()[g]) + 1) % 3
        for gg in range(G):
            c.nodes[victim].drain(gg)
        want_log = [c.nodes[victim].log(gg) for gg in range(G)]
        c.stop(victim)
        elect(c)
        c.nodes[int(c.leaders()[g])].propose(g, b"INSERT INTO t (v) VALUES ('foo')")
        c.settle()
        before = len(c.wal[victim])
        nd = c.restart_from_wal(victim)
        for gg in range(G):
            replayed = nd.drain(gg)
            assert replayed[-1] is None and replayed[:-1] == [d for _, d in want_log[gg] if d]  # exactly its log, then nil
            assert nd.log(gg) == want_log[gg] and nd.status(gg).term == 0  # HardState dropped, as the reference (F6)
        c.run(5)
        c.settle()
        assert nd.drain(g) == [b"INSERT INTO t (v) VALUES ('foo')"]
        assert len(c.wal[victim]) > before
        _, logs, hs = _wal_view(bytes(c.wal[victim]), G)  # old bytes + what the restarted node appended: still one chain
        for gg in range(G):
            assert logs[gg] == nd.log(gg)
        check_safety(c)
        # restoring HardState instead (what a correct raft needs): term, vote and commit come back
        c.stop(victim)
        nd = c.restart_from_wal(victim, restore_hard_state=True)
        for gg in range(G):
            st = nd.status(gg)
            assert (st.term, st.vote, st.commit) == hs[gg] and st.term > 0
    finally:
        c.close()


def test_replay_wal_refuses_a_corrupt_segment(Cluster):
    from raftsql_amd.engine import RaftqError
    from raftsql_amd.node import RaftNode

    c = Cluster(3, 3, wal=True)
    try:
        c.start()
        elect(c)
        for g in range(3):
            c.nodes[int(c.leaders()[g])].propose(g, b"payload %d " % g * 40)
        c.settle()
        wal = bytes(c.wal[0])
    finally:
        c.close()
    nd = RaftNode(3, 3, 0)
    try:
        assert nd.replay_wal(wal[: len(wal) - 3]) > 3  # a torn tail: the whole frames still replay
    finally:
        nd.destroy()
    from oracle import pywire as W

    off, _ = W.scan_frames(wal, big_endian=False)
    recs, _, _ = W.wal_decode(wal, off, 0)
    ent = [i for i in range(len(recs)) if recs[i]["kind"] == W.WAL_ENTRY and recs[i]["data_len"] > 10]
    # a payload byte, a record's first tag (-> does not parse), a stored crc.  (A flipped bit in a length
    # word moves the frame boundaries instead: the reader sees a torn tail there, as wal.ReadAll does.)
    for pos in (int(recs[ent[0]]["data_off"]) + 5, int(off[ent[-1]]) + 8, int(off[ent[1]]) + 11):
        bad = bytearray(wal)
        bad[pos] ^= 0x04
        nd = RaftNode(3, 3, 0)
        try:
            with pytest.raises(RaftqError) as ei:
                nd.replay_wal(bytes(bad))
            assert "CRC" in str(ei.value) or "parse" in str(ei.value) or "place" in str(ei.value)
        finally:
            nd.destroy()


def test_propose_batch_is_propose_k_times(Cluster):
    from raftsql_amd.engine import RaftqError

    c = Cluster(8, 3)
    try:
        c.start()
        elect(c)
        lead = c.leaders()
        for p, nd in enumerate(c.nodes):
            mine = [int(g) for g in np.nonzero(lead == p)[0]]
            nd.propose_batch(mine + mine, [b"a%d" % g for g in mine] + [b"" if g % 2 else b"b%d" % g for g in mine])
        c.settle()
        c.run(2)
        c.settle()
        for nd in c.nodes:
            for g in range(8):
                want = [b"a%d" % g] + ([] if g % 2 else [b"b%d" % g])  # an empty payload is never published (raft.go:85-87)
                assert [d for d in nd.drain(g) if d is not None] == want
        with pytest.raises(RaftqError):
            c.nodes[0].propose_batch([0, 8], [b"x", b"no such group"])  # refused as a whole
        c.settle()
        assert all(nd.drain(0) == [] for nd in c.nodes)
    finally:
        c.close()


def test_status_batch_is_status_g_times(Cluster):
    """raftq_node_status_batch: the same seven fields as raftq_node_status for every group of a range, in one
    call; ranges are checked."""
    from raftsql_amd.engine import RaftqError

    c = Cluster(300, 3, seed=4)
    try:
        c.start()
        elect(c, max_ticks=200)
        lead = c.leaders()
        for g in range(0, 300, 3):
            c.nodes[int(lead[g])].propose(g, b"s%d" % g)
        c.settle()
        for nd in c.nodes:
            st = nd.statuses()
            assert st.shape == (300,)
            for g in range(300):
                one = nd.status(g)
                for f in ("term", "commit", "last_index", "applied", "lead", "vote", "role"):
                    assert int(st[f][g]) == int(getattr(one, f)), (g, f)
            part = nd.statuses(17, 40)
            assert part.tobytes() == st[17:57].tobytes()
            assert nd.statuses(300, 0).shape == (0,)
            with pytest.raises(RaftqError):
                nd.statuses(290, 11)
            assert np.array_equal(nd.roles(), st["role"])
    finally:
        c.close()


def test_threaded_cluster_is_the_serial_cluster(Cluster):
    """Cluster(threads=True) runs every node's turn on its own thread (what N machines do); the transport still
    moves the bytes after all turns, so the run is the serial run: same leaders, same logs, same statistics -- and the
    same again when the library moves the frames from node to node itself (native_transport, raftq_node_forward)."""
    runs = []
    for threads, native in ((False, False), (True, False), (True, True), (False, True)):
        # native: the frames go from node to node inside the library (raftq_node_forward) instead of through Python bytes
        c = Cluster(1500, 3, seed=21, threads=threads, native_transport=native)
        try:
            c.start()
            elect(c, max_ticks=200)
            lead = c.leaders()
            for r in range(3):
                for p, nd in enumerate(c.nodes):
                    mine = np.nonzero(lead == p)[0]
                    nd.propose_batch(mine, [b"r%d g%d" % (r, g) for g in mine])
                c.settle()
                c.run(2)
            c.settle()
            check_safety(c)
            runs.append((lead.copy(), [nd.statuses().tobytes() for nd in c.nodes], [nd.stats() for nd in c.nodes],
                         [[nd.drain(g) for g in range(0, 1500, 37)] for nd in c.nodes]))
        finally:
            c.close()
    a = runs[0]
    for b in runs[1:]:
        assert np.array_equal(a[0], b[0])
        assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3]
    assert all(ch == [None, b"r0 g%d" % g, b"r1 g%d" % g, b"r2 g%d" % g] for g, ch in zip(range(0, 1500, 37), a[3][0]))


def test_many_proposals_per_group_in_one_turn(Cluster):
    """A burst of 200 proposals for ONE group (and one each for the others) in one turn: the entries keep their
    order through the arena-backed log, MsgApp batching and the per-group rounds of the log-tail report."""
    c = Cluster(16, 3, seed=9)
    try:
        c.start()
        elect(c, max_ticks=200)
        lead = c.leaders()
        hot = 5
        burst = [b"hot-%03d-" % i + b"x" * (i % 50) for i in range(200)]
        nd = c.nodes[int(lead[hot])]
        nd.propose_batch([hot] * len(burst), burst)
        for g in range(16):
            if g != hot:
                c.nodes[int(lead[g])].propose(g, b"cold%d" % g)
        c.settle()
        c.run(2)
        c.settle()
        for node in c.nodes:
            assert [d for d in node.drain(hot) if d is not None] == burst
            for g in range(16):
                if g != hot:
                    assert [d for d in node.drain(g) if d is not None] == [b"cold%d" % g]
            assert [e[1] for e in node.log(hot)][-200:] == burst
        check_safety(c)
    finally:
        c.close()


def test_forward_moves_or_drops_whole_queues(Cluster):
    """raftq_node_forward: what a node queued for a peer reaches that peer's next turn byte for byte (the same frames
    raftq_node_poll would hand out), or -- to == NULL -- is gone; nothing is delivered twice, nothing stays queued."""
    c = Cluster(40, 3, seed=5, native_transport=True)
    try:
        c.start()
        elect(c)
        lead = c.leaders()
        p = int(lead[7])
        c.nodes[p].propose(7, b"one")
        c.nodes[p].advance()
        others = [q for q in range(3) if q != p]
        queued = c.nodes[p].poll(others[0])  # the frames themselves, for comparison ...
        assert queued and c.nodes[p].poll(others[0]) == b""
        c.nodes[others[0]].deliver(queued)   # ... delivered the classic way
        moved = c.nodes[p].forward(others[1], c.nodes[others[1]])  # the other follower gets its copy through forward
        assert moved == len(queued) and c.nodes[p].forward(others[1], c.nodes[others[1]]) == 0
        c.settle()
        assert [nd.status(7).commit for nd in c.nodes] == [c.nodes[p].status(7).commit] * 3
        assert all(nd.drain(7)[-1] == b"one" for nd in c.nodes)
        c.nodes[p].propose(7, b"two")
        c.nodes[p].advance()
        assert c.nodes[p].forward(others[0], None) > 0 and c.nodes[p].forward(others[1], None) > 0  # both transfers lost
        assert c.nodes[p].poll(others[0]) == b"" and c.nodes[p].poll(others[1]) == b""
        c.settle()
        assert all(nd.drain(7) == [] for nd in c.nodes)  # nobody can have committed "two" yet
        c.run(3)  # heartbeats find the followers behind and resend
        c.settle()
        assert all(nd.drain(7) == [b"two"] for nd in c.nodes)
    finally:
        c.close()


def test_crank_steps_like_the_serial_cluster_and_refuses_bad_calls(Cluster):
    """raftq_crank_step: one call turns every live node on a thread of its own and moves the frames per addressee -- the
    delivered streams are those of the serial crank (the 4-way test above compares every frame); here: its bookkeeping
    (entries published per node, the two halves' wall time), a partly polled queue still forwarded frame for frame, a
    stopped node's NULL slot, a live bit for a node that is not there, and proposals handed over as arrays."""
    import ctypes as C

    from raftsql_amd import node as ND

    G, N = 300, 3
    c = Cluster(G, N, seed=11, threads=True, native_transport=True)
    try:
        c.start()
        elect(c)
        lead = c.leaders()
        for p, nd in enumerate(c.nodes):  # raftq_node_propose_batch from arrays: no Python list of payloads
            mine = np.nonzero(lead == p)[0]
            stmt = b"INSERT %d" % p
            nd.propose_blob(mine, np.arange(len(mine) + 1, dtype=np.uint64) * len(stmt), stmt * len(mine))
        total = [0] * N
        for _ in range(8):
            c.step(tick=False)
            total = [a + b for a, b in zip(total, c.last_published)]
        assert total == [G] * N and c.seconds["turns"] > 0 and c.seconds["transport"] > 0
        for p, nd in enumerate(c.nodes):
            for g in range(G):
                assert [d for d in nd.drain(g) if d is not None] == [b"INSERT %d" % int(lead[g])]
        # a queue somebody polled from has lost its frame ends: forward still moves whole frames (it walks them again)
        p = int(lead[5])
        q = (p + 1) % N
        c.nodes[p].propose(5, b"x" * 40)
        c.nodes[p].propose(6 if lead[6] == p else 5, b"y" * 50)
        c.nodes[p].advance()
        first = c.nodes[p].poll(q, cap=170)  # the first frame or two, not all of them
        assert 0 < len(first) <= 170
        c.nodes[q].deliver(first)
        c.run(4, tick=False)
        check_safety(c)
        assert all(nd.status(5).commit == c.nodes[p].status(5).commit for nd in c.nodes)
        # the crank itself: a stopped node is a NULL slot; asking for its turn is refused, the others go on
        lib = ND._load()
        c.stop(2)
        c.step(tick=True)
        assert c._crank is not None
        pub = np.zeros(N, np.uint64)
        assert lib.raftq_crank_step(c._crank, 0b111, 0, None, 0, pub.ctypes.data, None) != 0  # node 2 is not there
        assert lib.raftq_crank_step(c._crank, 0b1000, 0, None, 0, None, None) != 0         # nor is a fourth
        assert lib.raftq_crank_step(c._crank, 0b011, 0, None, 1, pub.ctypes.data, None) == 0
        out = C.c_void_p()
        assert lib.raftq_crank_create(None, 3, None, C.byref(out)) != 0
        two = (C.c_void_p * 2)(c.nodes[0]._p, c.nodes[1]._p)
        assert lib.raftq_crank_create(two, 2, None, C.byref(out)) != 0  # nodes of a 3-peer cluster are not a 2-peer one
    finally:
        c.close()


def test_followers_without_tail_appends(Cluster, monkeypatch):
    """RAFTQ_NODE_TAIL_APPENDS=0: MsgApps go to Step as headers only, every follower append runs raftLog.maybeAppend on the
    node's log and reports its tail (raftq_apply_log_deltas) -- round 2's division of labour, now only taken by gaps, conflicts
    and stale indices, kept whole here: plain replication, then a chaos seed."""
    monkeypatch.setenv("RAFTQ_NODE_TAIL_APPENDS", "0")
    c = Cluster(50, 3, seed=13)
    try:
        c.start()
        elect(c)
        lead = c.leaders()
        for g in range(50):
            c.nodes[int(lead[g])].propose(g, b"stmt %d" % g)
        c.run(6, tick=False)
        c.settle()
        for nd in c.nodes:
            for g in range(50):
                assert [d for d in nd.drain(g) if d is not None] == [b"stmt %d" % g]
        check_safety(c)
    finally:
        c.close()
    test_chaos_safety_and_convergence(Cluster, 8, False, False)


def test_one_group_with_a_very_long_log(Cluster):
    """70,000 entries in ONE group's log (and on its commit channels): the per-group arrays outgrow the node's pool classes
    (1 MB blocks) and move to allocations of their own, on the leader and on the followers; every node delivers every
    statement in order, and the group next door is untouched."""
    c = Cluster(3, 3, seed=3, threads=True, native_transport=True)
    try:
        c.start()
        elect(c)
        lead = int(c.leaders()[1])
        n, per = 70000, 3500
        for k in range(0, n, per):
            g = np.full(per, 1, dtype=np.uint64)
            stmt = [b"s%07d" % i for i in range(k, k + per)]
            off = np.arange(per + 1, dtype=np.uint64) * 8
            c.nodes[lead].propose_blob(g, off, b"".join(stmt))
            c.run(3, tick=False)
        c.settle()
        for nd in c.nodes:
            got = [d for d in nd.drain(1) if d is not None]
            assert len(got) == n and got[0] == b"s0000000" and got[-1] == b"s%07d" % (n - 1) and got == sorted(got)
            assert [d for d in nd.drain(0) if d is not None] == []
            assert int(nd.status(1).commit) >= n
        check_safety(c)
    finally:
        c.close()


def test_payloads_around_and_beyond_64_kib(Cluster):
    """Statements of 65,534 / 65,535 / 65,536 / 200,000 bytes (and one of a single byte) are committed, delivered everywhere
    byte for byte, read back with raftq_node_entry, written to the WAL and replayed."""
    rng = np.random.default_rng(4)
    sizes = [1, 65534, 65535, 65536, 200000, 17]
    stmts = [bytes(rng.integers(1, 255, k, dtype=np.uint8)) for k in sizes]
    c = Cluster(4, 3, seed=9, wal=True)
    try:
        c.start()
        elect(c)
        lead = int(c.leaders()[2])
        for st in stmts:
            c.nodes[lead].propose(2, st)
            c.run(3, tick=False)
        c.settle()
        for nd in c.nodes:
            assert [d for d in nd.drain(2) if d is not None] == stmts
            assert [d for _, d in nd.log(2) if d] == stmts
        check_safety(c)
        follower = (lead + 1) % 3
        c.stop(follower)
        nd = c.restart_from_wal(follower)
        got = nd.drain(2)
        assert got[-1] is None and [d for d in got if d is not None] == stmts  # the replay, then the nil sentinel
    finally:
        c.close()
