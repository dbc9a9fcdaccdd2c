"""GPU: the multi-group propose -> commit pipe (include/raftq_pipe.h), written
after the reference's own integration tests (raftsql_test.go:92-171): propose
statements, let a quorum acknowledge, and check what the commit channel shows.
The reference tests run 3 processes over loopback; here the followers are the
test itself (it acknowledges log indices), the quorum check is the GPU sweep."""
import threading
import time

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("stage_mode")]  # every test, both staging forms


@pytest.fixture()
def pipe_cls(gpu_engine_cls):
    from raftsql_amd.pipe import MultiRaftPipe

    return MultiRaftPipe


def test_new_db_analog_three_peers(pipe_cls):
    """TestNewDB (raftsql_test.go:92-115): a CREATE and three INSERTs become visible,
    in order, once 2 of 3 peers hold them -- and not before."""
    from raftsql_amd import pipe as P

    with pipe_cls(4, 3) as rp:
        rp.start()
        for g in range(4):
            assert rp.drain(g) == [None]  # fresh log: just the nil sentinel (raft.go:131-132)
        stmts = [b"CREATE TABLE t (id int primary key, v int)"] + [b"INSERT INTO t (v) VALUES (%d)" % i for i in range(3)]
        for s in stmts:
            rp.propose(1, s)
        assert rp.last_index(1) == 1 + len(stmts)  # the leader's empty entry + 4 statements
        assert rp.flush() == 0  # only the leader holds them: 1 of 3 is not a quorum
        assert rp.drain(1) == [] and rp.committed(1) == 0
        rp.process_app_resp(1, 1, 3)  # follower 1 has the no-op, CREATE and the first INSERT
        assert rp.flush() == 1
        assert rp.drain(1) == stmts[:2] and rp.committed(1) == 3
        rp.process_app_resp(1, 2, 5)  # follower 2 caught up completely
        assert rp.flush() == 1
        assert rp.drain(1) == stmts[2:] and rp.committed(1) == 5
        # other groups saw nothing (the "main.x does not exist" half of TestNewDB)
        for g in (0, 2, 3):
            assert rp.drain(g) == [] and rp.committed(g) == 0
        assert rp.close() == 0
        assert rp.recv(1)[0] == P.CLOSED


def test_restart_db_analog(pipe_cls):
    """TestRestartDB (raftsql_test.go:117-171): a restarted node replays exactly its 4 logged
    entries, then the nil sentinel; a new statement needs a quorum again."""
    with pipe_cls(2, 3) as rp:
        logged = [(1, b"CREATE TABLE t (id int primary key, v int)"), (1, b"INSERT INTO t (v) VALUES (0)"),
                  (1, b"INSERT INTO t (v) VALUES (1)"), (2, b"INSERT INTO t (v) VALUES (2)")]
        rp.replay(0, logged)
        rp.start()
        got = rp.drain(0)
        assert got[:4] == [d for _, d in logged] and got[4] is None and len(got) == 5
        assert rp.term(0) == 3 and rp.last_index(0) == 5  # new term, leader's empty entry appended
        rp.propose(0, b"INSERT INTO t (v) VALUES (3)")  # "foo" of the reference test
        rp.flush()
        assert rp.drain(0) == []  # absent right after replay ...
        rp.process_app_resp(0, 2, 6)  # ... present once a second peer has it
        rp.flush()
        assert rp.drain(0) == [b"INSERT INTO t (v) VALUES (3)"]
        assert rp.drain(1) == [None]


def test_old_term_entries_wait_for_the_current_term(pipe_cls):
    """raftLog.maybeCommit's gate (Raft 5.4.2): a quorum on an index of an OLD term does not
    commit it; it commits together with the first entry of the leader's own term."""
    with pipe_cls(1, 3) as rp:
        rp.replay(0, [(1, b"a"), (1, b"b")])
        rp.start()
        assert rp.drain(0) == [b"a", b"b", None]
        rp.propose(0, b"c")  # index 4, term 2 (index 3 is the term-2 no-op)
        rp.process_app_resp(0, 1, 2)  # follower only confirms the old-term prefix
        assert rp.flush() == 0 and rp.committed(0) == 2
        rp.process_app_resp(0, 1, 3)  # now the no-op of term 2 is on a quorum
        assert rp.flush() == 1 and rp.committed(0) == 3 and rp.drain(0) == []  # no-op is never delivered
        rp.process_app_resp(0, 1, 4)
        assert rp.flush() == 1 and rp.drain(0) == [b"c"]


def test_five_peers_need_three(pipe_cls):
    with pipe_cls(1, 5) as rp:
        rp.start()
        rp.drain(0)
        rp.propose(0, b"x")
        rp.process_app_resp(0, 1, 2)
        assert rp.flush() == 0
        rp.process_app_resp(0, 4, 2)
        assert rp.flush() == 1 and rp.drain(0) == [b"x"]
        # stale / duplicate acks change nothing (Progress.maybeUpdate only raises Match)
        rp.process_app_resp(0, 1, 1)
        rp.process_app_resp(0, 4, 2)
        assert rp.flush() == 0


def test_argument_and_state_errors(pipe_cls):
    from raftsql_amd.engine import RaftqError

    with pipe_cls(2, 3) as rp:
        with pytest.raises(RaftqError) as ei:
            rp.propose(0, b"early")  # not started
        assert ei.value.code == -4
        with pytest.raises(RaftqError):
            rp.replay(0, [(2, b"a"), (1, b"b")])  # terms must not decrease
        rp.start()
        with pytest.raises(RaftqError) as ei:
            rp.replay(0, [(1, b"late")])
        assert ei.value.code == -4
        for bad in ((5, 1, 1), (0, 0, 1), (0, 3, 1), (0, 1, 99)):
            with pytest.raises(RaftqError) as ei:
                rp.process_app_resp(*bad)
            assert ei.value.code == -1
        assert rp.close() == 0
        with pytest.raises(RaftqError) as ei:
            rp.propose(0, b"after close")
        assert ei.value.code == -4


def test_appends_are_exposed_for_the_transport(pipe_cls):
    with pipe_cls(2, 3) as rp:
        rp.start()
        assert rp.take_appends() == [(0, 1, 1, 0), (1, 1, 1, 0)]  # the leaders' empty entries
        rp.propose(1, b"hello")
        assert rp.take_appends() == [(1, 2, 1, 5)]
        assert rp.entry(1, 2) == (1, b"hello")
        assert rp.take_appends() == []


def test_background_batching_thread(pipe_cls):
    """The batching goroutine form: proposals and acks come from other threads, commits are read
    with a blocking receive."""
    from raftsql_amd import pipe as P

    G, per_group = 64, 20
    with pipe_cls(G, 3) as rp:
        rp.start(max_batch=256, max_wait_us=100, background=True)
        for g in range(G):
            assert rp.recv(g, 1000)[0] == P.SENTINEL

        def client(g0, g1):
            for i in range(per_group):
                for g in range(g0, g1):
                    rp.propose(g, b"g%d-%d" % (g, i))
                    rp.process_app_resp(g, 1 + (i & 1), rp.last_index(g))

        ts = [threading.Thread(target=client, args=(k * 16, (k + 1) * 16)) for k in range(4)]
        for t in ts:
            t.start()
        for g in range(G):
            for i in range(per_group):
                kind, data = rp.recv(g, 5000)
                assert kind == P.ENTRY and data == b"g%d-%d" % (g, i)
        for t in ts:
            t.join()
        assert rp.close() == 0


@pytest.mark.parametrize("segments", [True, False])
def test_many_groups_random_ack_order(pipe_cls, segments, monkeypatch):
    """100k groups: every statement is delivered exactly once, in log order, and only after a
    quorum -- with acks arriving in random order across several batching turns.  The pipe reads its turns' advance lists in
    segments (RAFTQ_CYCLE_SEGMENTED); with RAFTQ_CYCLE_SEGMENTS=0 the library produces the contiguous list instead, longer
    than the pipe's buffer here (all 100,000 groups advance in one turn): the pipe's way out through raftq_collect_changed."""
    if not segments:
        monkeypatch.setenv("RAFTQ_CYCLE_SEGMENTS", "0")
    rng = np.random.default_rng(3)
    G, N = 100_000, 5
    with pipe_cls(G, N) as rp:
        rp.start()
        t0 = time.time()
        for g in range(G):
            rp.propose(g, b"s%d" % g)
        acks = [(g, p) for g in range(0, G, 1) for p in (1, 2, 3)]
        order = rng.permutation(len(acks))
        delivered = 0
        for chunk in np.array_split(order, 5 if segments else 1):  # (one turn: all 100,000 groups advance, the buffer holds 65,536)
            for k in chunk:
                g, p = acks[k]
                rp.process_app_resp(g, p, 2)
            delivered += rp.flush()
        assert delivered == G
        sample = np.unique(rng.integers(0, G, 2000))
        for g in sample:
            assert rp.drain(int(g)) == [None, b"s%d" % g]
        assert rp.flush() == 0
        assert time.time() - t0 < 120
