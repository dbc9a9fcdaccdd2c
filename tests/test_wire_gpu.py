"""GPU parity: the batched wire / WAL codecs (include/raftq_wire.h, raftq_wire_kernels.hpp) through the
C-ABI against the CPU oracle (oracle/raftq_wire_oracle.c, itself pinned to the protobuf runtime and
RFC 3720 in tests/test_wire_oracle.py), byte for byte and record for record: canonical and
non-canonical streams, malformed / corrupted / random-garbage frames, long payloads (the
wave-cooperative CRC and copy paths), CRC chains with re-seeding records, the committed fixtures,
and -- at sizes the oracle would not finish quickly -- round-trip and chain-splitting properties.
Step fed from frames must equal Step fed the decoded records."""
import json
import os

import numpy as np
import pytest

from oracle import pywire as W
from tests import _wiregen

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wire_golden.json")


class _PinnedCalls:
    """The engine with every codec call made on page-locked buffers (raftq_host_alloc) -- what a node hands the codecs every
    turn, and what takes the streaming form of the call (one kernel: readers | workers).  Same methods, same results."""

    def __init__(self, e):
        self._e = e

    def __getattr__(self, k):
        return getattr(self._e, k)

    @staticmethod
    def _pin(a, dt=None):
        from raftsql_amd.engine import pinned_copy

        a = np.ascontiguousarray(a if dt is None else np.asarray(a, dtype=dt))
        return pinned_copy(a) if a.size else a

    def wire_decode(self, stream, frame_off, want_ents=True, **kw):
        from raftsql_amd.engine import pinned_empty

        s, off = self._pin(_wiregen_u8(stream)), self._pin(frame_off, np.uint64)
        n = len(off) - 1
        msgs = pinned_empty(max(n, 1), W.WIRE_MSG_DT)
        if not want_ents:
            m, e, bad = self._e.wire_decode(s, off, want_ents=False, msgs=msgs)
            return m.copy(), e, bad
        m, _, _ = self._e.wire_decode(s, off, want_ents=False, msgs=msgs)  # (headers only: how many entry headers there are)
        ents = pinned_empty(int(m["n_ents"].sum()) + 1, W.WIRE_ENT_DT)
        m, e, bad = self._e.wire_decode(s, off, msgs=msgs, ents=ents)
        return m.copy(), e.copy(), bad

    def wal_decode(self, data, frame_off, prev_crc=0, **kw):
        from raftsql_amd.engine import pinned_empty

        b, off = self._pin(_wiregen_u8(data)), self._pin(frame_off, np.uint64)
        r, nv, lc = self._e.wal_decode(b, off, prev_crc, recs=pinned_empty(max(len(off) - 1, 1), W.WAL_REC_DT))
        return r.copy(), nv, lc

    def wire_encode(self, msgs, ents=None, pool=b"", out=None, off=None):
        from raftsql_amd.engine import pinned_empty

        if out is not None:
            return self._e.wire_encode(msgs, ents, pool, out=out, off=off)
        m = self._pin(msgs, W.WIRE_MSG_DT)
        e = self._pin(ents if ents is not None else np.zeros(0, W.WIRE_ENT_DT), W.WIRE_ENT_DT)
        p = self._pin(_wiregen_u8(pool))
        try:
            size = len(W.wire_encode(np.asarray(msgs), np.asarray(e), np.asarray(p))[0])  # (the size only: the call needs a buffer)
        except Exception:  # noqa: BLE001 - input the oracle refuses: any buffer will do, the call has to refuse too
            size = 128 * len(m) + len(p) + 1024
        o, f = pinned_empty(size + 16, np.uint8), pinned_empty(len(m) + 1, np.uint64)
        got, goff = self._e.wire_encode(m, e, p, out=o, off=f)
        return got.copy(), goff.copy()

    def wal_encode(self, recs, pool=b"", prev_crc=0, out=None, off=None):
        from raftsql_amd.engine import pinned_empty

        if out is not None:
            return self._e.wal_encode(recs, pool, prev_crc, out=out, off=off)
        r, p = self._pin(recs, W.WAL_REC_DT), self._pin(_wiregen_u8(pool))
        try:
            size = len(W.wal_encode(np.asarray(recs), np.asarray(p), prev_crc)[0])
        except Exception:  # noqa: BLE001
            size = 128 * len(r) + len(p) + 1024
        o, f = pinned_empty(size + 16, np.uint8), pinned_empty(len(r) + 1, np.uint64)
        got, goff, last = self._e.wal_encode(r, p, prev_crc, out=o, off=f)
        return got.copy(), goff.copy(), last


@pytest.fixture(scope="module", params=["pageable", "page-locked"])
def eng(request):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a visible GPU")
    from raftsql_amd.wire import WireEngine

    with WireEngine(4096, 5, self_peer=0) as e:
        yield e if request.param == "pageable" else _PinnedCalls(e)


def _same(a: np.ndarray, b: np.ndarray, what=""):
    assert a.dtype.itemsize == b.dtype.itemsize and len(a) == len(b), what
    if a.tobytes() != b.tobytes():
        for i in range(len(a)):
            assert a[i].tobytes() == b[i].tobytes(), (what, i, a[i], b[i])


def _be(body: bytes) -> bytes:
    return len(body).to_bytes(8, "big") + body


# ---- raftpb.Message ------------------------------------------------------------------------------

@pytest.mark.parametrize("seed,n,big", [(101, 1, 0), (102, 63, 0), (103, 1000, 7), (104, 5000, 3), (105, 257, 1)])
def test_encode_parity(eng, seed, n, big):
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, n, big_every=big, ent_frac=0.3)
    want, want_off = W.wire_encode(m, e, pool)
    got, off = eng.wire_encode(m, e, pool)
    assert np.array_equal(off, want_off)
    assert got.tobytes() == want.tobytes()


def test_encode_hand_vector(eng):
    m = np.zeros(1, W.WIRE_MSG_DT)
    m["type"], m["to"], m["from"], m["term"] = 6, 1, 0, 5  # MsgVoteResp{To: 2, From: 1, Term: 5}
    s, off = eng.wire_encode(m)
    body = bytes.fromhex("0806" "1002" "1801" "2005" "2800" "3000" "4000" "4a0812060a0010001800" "5000" "5800" "6000")
    assert s.tobytes() == _be(body) and list(off) == [0, 38]


def test_encode_shared_and_unordered_entry_ranges(eng):
    """ent_first is arbitrary: messages may share entries (a leader sends the same suffix to every
    follower) and reference them out of order."""
    rng = np.random.default_rng(7)
    m, e, pool = _wiregen.random_msgs(rng, 40, ent_frac=1.0, big_every=4)
    m["type"] = 3
    m["n_ents"] = np.minimum(3, len(e))
    m["ent_first"] = rng.integers(0, len(e) - 3, len(m))
    want, want_off = W.wire_encode(m, e, pool)
    got, off = eng.wire_encode(m, e, pool)
    assert np.array_equal(off, want_off) and got.tobytes() == want.tobytes()


def test_encode_rejects_bad_input(eng):
    from raftsql_amd.engine import RaftqError

    rng = np.random.default_rng(8)
    m, e, pool = _wiregen.random_msgs(rng, 100, ent_frac=0.5)
    for field, val in (("to", 255), ("from", 255), ("from", 0xFFFFFFFF)):
        bad = m.copy()
        bad[field][50] = val
        with pytest.raises(RaftqError):
            eng.wire_encode(bad, e, pool)
    i = int(np.nonzero(m["n_ents"])[0][0])
    bad = m.copy()
    bad["ent_first"][i] = len(e)  # entry range past ents[]
    with pytest.raises(RaftqError):
        eng.wire_encode(bad, e, pool)
    j = int(np.nonzero(e["data_len"])[0][0])
    bade = e.copy()
    bade["data_off"][j] = len(pool)  # payload past the pool
    mm = m.copy()
    mm["ent_first"][i], mm["n_ents"][i] = j, 1
    with pytest.raises(RaftqError):
        eng.wire_encode(mm, bade, pool)
    # a too-small buffer is refused and the size needed is reported
    import ctypes as C

    from raftsql_amd import _lib

    c = _lib.WireCounts()
    out = np.zeros(10, np.uint8)
    rc = eng._lib.raftq_wire_encode(eng._h, m.ctypes.data, len(m), e.ctypes.data, len(e), pool.ctypes.data, len(pool),
                                    out.ctypes.data, len(out), None, C.byref(c))
    assert rc == _lib.RAFTQ_EINVAL and c.bytes == len(W.wire_encode(m, e, pool)[0]) and not out.any()
    # and the engine still works afterwards
    got, _ = eng.wire_encode(m, e, pool)
    assert got.tobytes() == W.wire_encode(m, e, pool)[0].tobytes()


def test_empty_batches(eng):
    s, off = eng.wire_encode(np.zeros(0, W.WIRE_MSG_DT))
    assert len(s) == 0 and list(off) == [0]
    mm, ee, bad = eng.wire_decode(b"", np.zeros(1, np.uint64))
    assert len(mm) == 0 and len(ee) == 0 and bad == 0
    out, off, last = eng.wal_encode(np.zeros(0, W.WAL_REC_DT), b"", 77)
    assert len(out) == 0 and last == 77
    rr, nv, last = eng.wal_decode(b"", np.zeros(1, np.uint64), 78)
    assert len(rr) == 0 and nv == 0 and last == 78


@pytest.mark.parametrize("seed,n,big", [(111, 700, 5), (112, 4000, 0)])
def test_decode_parity_canonical(eng, seed, n, big):
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, n, big_every=big, ent_frac=0.4)
    s, off = W.wire_encode(m, e, pool)
    wm, we, wbad = W.wire_decode(s, off)
    gm, ge, gbad = eng.wire_decode(s, off)
    assert gbad == wbad == 0
    _same(gm, wm, "msgs")
    _same(ge, we, "ents")
    hm, he, _ = eng.wire_decode(s, off, want_ents=False)  # headers only
    _same(hm, wm, "msgs (no entries requested)")
    assert len(he) == 0


@pytest.mark.parametrize("seed", [121, 122])
def test_decode_parity_noncanonical(eng, seed):
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, 300, ent_frac=0.4)
    bodies = [_wiregen.noncanonical_message(rng, m[i], e, pool) for i in range(len(m))]
    stream = b"".join(_be(b) for b in bodies)
    off = np.concatenate([[0], np.cumsum([len(b) + 8 for b in bodies])]).astype(np.uint64)
    wm, we, wbad = W.wire_decode(stream, off)
    gm, ge, gbad = eng.wire_decode(stream, off)
    assert wbad == gbad == 0
    _same(gm, wm, "msgs")
    _same(ge, we, "ents")


def test_decode_malformed_frames(eng):
    good = _be(bytes.fromhex("0806100218012005"))
    cases = [
        (9).to_bytes(8, "big") + bytes.fromhex("0806100218012005"),  # length word disagrees
        _be(bytes.fromhex("080610021801208080")),  # truncated varint
        _be(bytes.fromhex("20" + "80" * 10 + "01")),  # 11-byte varint
        _be(bytes.fromhex("08033a0508001005")),  # entry length overruns
        _be(bytes.fromhex("2201aa")),  # wrong wire type on term
        _be(bytes.fromhex("0001")),  # tag 0
        _be(bytes.fromhex("6b")),  # group wire type
        _be(bytes.fromhex("3a021280")),  # bad entry inside
        _be(bytes.fromhex("4a021201")),  # bad snapshot inside
        b"\x00\x00\x00",  # shorter than a length word
        _be(bytes.fromhex("08" + "ff" * 9 + "01" + "20" + "ff" * 9 + "7f")),  # 10-byte varints: fine
        _be(bytes.fromhex("4a00")),  # empty snapshot, nothing else
        _be(bytes.fromhex("4a061204" "0a020801" "3a00" "3a021001")),  # snapshot with a conf_state member; two entries
    ]
    stream, off = good, [0, len(good)]
    for b in cases:
        stream += b + good
        off += [off[-1] + len(b), off[-1] + len(b) + len(good)]
    off = np.array(off, np.uint64)
    wm, we, wbad = W.wire_decode(stream, off)
    gm, ge, gbad = eng.wire_decode(stream, off)
    assert gbad == wbad == 10
    assert list(wm["flags"][1::2]) == [W.F_MALFORMED] * 10 + [0, 0, W.F_SNAPSHOT] and wm["n_ents"][-2] == 2
    _same(gm, wm, "msgs")
    _same(ge, we, "ents")
    # frame extents that are themselves nonsense: decreasing, past the end
    off2 = np.array([0, len(good), 5, len(stream) + 100, len(stream)], np.uint64)
    wm, we, wbad = W.wire_decode(stream, off2)
    gm, ge, gbad = eng.wire_decode(stream, off2)
    assert gbad == wbad
    _same(gm, wm, "msgs")


@pytest.mark.parametrize("seed", [131, 132, 133])
def test_decode_fuzz(eng, seed):
    """Mutated valid frames and pure noise: whatever the bytes, GPU and oracle agree on every record
    (and nothing faults: every read stays inside the frame)."""
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, 1500, ent_frac=0.5, max_payload=60)
    s, off = W.wire_encode(m, e, pool)
    s = s.copy()
    n_mut = len(s) // 40
    pos = rng.integers(0, len(s), n_mut)
    s[pos] = rng.integers(0, 256, n_mut, dtype=np.uint8)
    # keep most length words intact so that the bodies get parsed
    for i in rng.choice(len(m), len(m) * 9 // 10, replace=False):
        a, b = int(off[i]), int(off[i + 1])
        s[a:a + 8] = np.frombuffer((b - a - 8).to_bytes(8, "big"), np.uint8)
    wm, we, wbad = W.wire_decode(s, off)
    gm, ge, gbad = eng.wire_decode(s, off)
    assert gbad == wbad and wbad > 0
    _same(gm, wm, "msgs")
    _same(ge, we, "ents")
    noise = rng.integers(0, 256, 20000, dtype=np.uint8)
    cuts = np.sort(rng.choice(np.arange(1, len(noise)), 600, replace=False))
    off = np.concatenate([[0], cuts, [len(noise)]]).astype(np.uint64)
    for i in range(len(off) - 1):  # valid length words, random bodies
        a, b = int(off[i]), int(off[i + 1])
        if b - a >= 8:
            noise[a:a + 8] = np.frombuffer((b - a - 8).to_bytes(8, "big"), np.uint8)
    wm, we, wbad = W.wire_decode(noise, off)
    gm, ge, gbad = eng.wire_decode(noise, off)
    assert gbad == wbad
    _same(gm, wm, "noise msgs")
    _same(ge, we, "noise ents")


def test_decode_entry_capacity(eng):
    import ctypes as C

    from raftsql_amd import _lib

    rng = np.random.default_rng(9)
    m, e, pool = _wiregen.random_msgs(rng, 200, ent_frac=1.0)
    s, off = W.wire_encode(m, e, pool)
    assert len(e) > 10
    msgs = np.zeros(len(m), W.WIRE_MSG_DT)
    ents = np.zeros(10, W.WIRE_ENT_DT)
    c = _lib.WireCounts()
    rc = eng._lib.raftq_wire_decode(eng._h, s.ctypes.data, len(s), off.ctypes.data, len(m), msgs.ctypes.data,
                                    ents.ctypes.data, 10, C.byref(c))
    assert rc == _lib.RAFTQ_EINVAL and c.n_ents == len(e)


def _form(monkeypatch, form):
    """streaming: one kernel, readers | workers (the default on page-locked buffers); copying: the runtime's copies around
    the kernel chain (RAFTQ_WIRE_STREAMING=0: what pageable buffers always get)"""
    monkeypatch.setenv("RAFTQ_WIRE_STREAMING", "1" if form == "streaming" else "0")


@pytest.mark.parametrize("copies", ["streaming", "streaming-128", "copying"])
@pytest.mark.parametrize("seed,n,big", [(171, 1, 0), (172, 900, 4), (173, 6000, 0), (174, 333, 1)])
def test_codecs_on_page_locked_buffers(seed, n, big, copies, monkeypatch):
    """What a node hands the codecs every turn: every buffer page-locked.  The call is then ONE launch with ONE wait and has to
    say and write exactly what the copying form does (the `runtime` rows): streams, offsets, records, entries, counts, the
    entry capacity, refusals.  A refused call of the copying form leaves the output alone; the streaming form has
    tiles on their way out before the verdict exists, so its promise is the ABI's: nothing at or behind out[cap] is touched."""
    import ctypes as C

    from raftsql_amd import _lib
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    if copies == "streaming-128":  # the decoder's 128-frame tile (34 KB of LDS, four workgroups per CU; the default is 256 frames, 68 KB)
        monkeypatch.setenv("RAFTQ_WIRE_TILE", "128")
        copies = "streaming"
    _form(monkeypatch, copies)
    rng = np.random.default_rng(seed)
    with WireEngine(4096, 5, self_peer=0) as eng:
        for rnd in range(3):
            m, e, pool = _wiregen.random_msgs(rng, n, big_every=big, ent_frac=(0.0, 0.4, 1.0)[rnd])
            want, want_off = W.wire_encode(m, e, pool)
            pm, pe, pp = pinned_copy(m), pinned_copy(e) if len(e) else e, pinned_copy(_wiregen_u8(pool))
            out, off = pinned_empty(len(want) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
            out[:] = 0xEE
            got, goff = eng.wire_encode(pm, pe, pp, out=out, off=off)
            assert np.array_equal(goff, want_off) and got.tobytes() == want.tobytes()
            assert bytes(out[len(want):]) == b"\xee" * 64  # nothing behind the stream was touched
            # too small a buffer: refused, the size needed is reported, nothing written at or behind out[cap]
            cap = max(1, len(want) - 1)
            small = pinned_empty(cap + 64, np.uint8)
            small[:] = 0xEE
            c = _lib.WireCounts()
            rc = eng._lib.raftq_wire_encode(eng._h, pm.ctypes.data, n, pe.ctypes.data if len(e) else None, len(e), pp.ctypes.data, len(pp),
                                            small.ctypes.data, cap, off.ctypes.data, C.byref(c))
            assert rc == _lib.RAFTQ_EINVAL and c.bytes == len(want) and bytes(small[cap:]) == b"\xee" * 64
            assert copies == "streaming" or bytes(small) == b"\xee" * len(small)
            # a bad message (addressee 255): refused
            bad = pinned_copy(m)
            bad["to"][n // 2] = 255
            out[:] = 0xEE
            rc = eng._lib.raftq_wire_encode(eng._h, bad.ctypes.data, n, pe.ctypes.data if len(e) else None, len(e), pp.ctypes.data, len(pp),
                                            out.ctypes.data, len(want), off.ctypes.data, C.byref(c))
            assert rc == _lib.RAFTQ_EINVAL and bytes(out[len(want):]) == b"\xee" * 64
            assert copies == "streaming" or bytes(out) == b"\xee" * len(out)
            # and the next call is fine again (the flag word was left zero)
            got, goff = eng.wire_encode(pm, pe, pp, out=out, off=off)
            assert got.tobytes() == want.tobytes()
            # decode: canonical, then with every 7th frame damaged
            for damage in (False, True):
                s = want.copy()
                if damage:
                    for k in range(0, n, 7):
                        s[int(want_off[k]) + 8] = 0x0B
                wm, we, wbad = W.wire_decode(s, want_off)
                ps, po = pinned_copy(s), pinned_copy(want_off)
                dm, de = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 3, W.WIRE_ENT_DT)
                gm, ge, gbad = eng.wire_decode(ps, po, msgs=dm, ents=de)
                assert gbad == wbad
                _same(gm, wm, "msgs")
                _same(ge, we, "ents")
                hm, he, hbad = eng.wire_decode(ps, po, want_ents=False, msgs=dm)
                _same(hm, wm, "msgs (headers only)")
                assert hbad == wbad
                if len(we) > 1:  # one entry too many for the caller's array: refused, the count needed is reported
                    few = pinned_empty(len(we) - 1, W.WIRE_ENT_DT)
                    rc = eng._lib.raftq_wire_decode(eng._h, ps.ctypes.data, len(ps), po.ctypes.data, n, dm.ctypes.data, few.ctypes.data,
                                                    len(few), C.byref(c))
                    assert rc == _lib.RAFTQ_EINVAL and c.n_ents == len(we)
                    gm, ge, _ = eng.wire_decode(ps, po, msgs=dm, ents=de)  # and the next call is whole again
                    _same(ge, we, "ents after a refusal")


@pytest.mark.parametrize("copies", ["streaming", "copying"])
@pytest.mark.parametrize("seed,n,big,prev", [(181, 1, 0, 0), (182, 700, 3, 0xDEADBEEF), (183, 5000, 40, 7)])
def test_wal_codecs_on_page_locked_buffers(seed, n, big, prev, copies, monkeypatch):
    """the WAL codecs the same way: every buffer page-locked -> one chain, one wait; bytes, offsets, CRC chain, records,
    valid count and refusals as the copying form's"""
    import ctypes as C

    from raftsql_amd import _lib
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    _form(monkeypatch, copies)
    rng = np.random.default_rng(seed)
    with WireEngine(4096, 5, self_peer=0) as eng:
        r, pool = _wiregen.random_wal(rng, n, max_payload=300, big_every=big)
        want, want_off, want_last = W.wal_encode(r, pool, prev)
        pr, pp = pinned_copy(r), pinned_copy(_wiregen_u8(pool))
        out, off = pinned_empty(len(want) + 64, np.uint8), pinned_empty(len(r) + 1, np.uint64)
        out[:] = 0xEE
        got, goff, glast = eng.wal_encode(pr, pp, prev, out=out, off=off)
        assert np.array_equal(goff, want_off) and got.tobytes() == want.tobytes() and glast == want_last
        assert bytes(out[len(want):]) == b"\xee" * 64
        cap = max(1, len(want) - 1)
        small = pinned_empty(cap + 64, np.uint8)
        small[:] = 0xEE
        c = _lib.WalCounts()
        rc = eng._lib.raftq_wal_encode(eng._h, pr.ctypes.data, len(r), pp.ctypes.data, len(pp), prev, small.ctypes.data, cap,
                                       off.ctypes.data, C.byref(c))
        assert rc == _lib.RAFTQ_EINVAL and c.bytes == len(want) and bytes(small[cap:]) == b"\xee" * 64
        assert copies == "streaming" or bytes(small) == b"\xee" * len(small)
        bad = pinned_copy(r)
        bad["kind"][len(r) // 2] = 99
        out[:] = 0xEE
        rc = eng._lib.raftq_wal_encode(eng._h, bad.ctypes.data, len(r), pp.ctypes.data, len(pp), prev, out.ctypes.data, len(want),
                                       off.ctypes.data, C.byref(c))
        assert rc == _lib.RAFTQ_EINVAL and bytes(out[len(want):]) == b"\xee" * 64
        assert copies == "streaming" or bytes(out) == b"\xee" * len(out)
        got, goff, glast = eng.wal_encode(pr, pp, prev, out=out, off=off)  # the flag word was left zero
        assert got.tobytes() == want.tobytes() and glast == want_last
        for damage in (False, True):
            s = want.copy()
            if damage and len(s) > 40:
                s[len(s) * 2 // 3] ^= 0x5A
            wr, wnv, wl = W.wal_decode(s, want_off, prev)
            ps, po = pinned_copy(s), pinned_copy(want_off)
            recs = pinned_empty(len(r), W.WAL_REC_DT)
            gr, gnv, gl = eng.wal_decode(ps, po, prev, recs=recs)
            assert (gnv, gl) == (wnv, wl) and gr.tobytes() == wr.tobytes()


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("wgs", [5, 64, 160, 1024])
def test_streaming_decode_at_bench_size(wgs, tile, monkeypatch):
    """The one-kernel form of raftq_wire_decode (page-locked buffers) at the bench's size -- 65,536 frames, a MsgApp share with
    entries, some damaged frames -- with few workgroups walking many tiles each (every one a tile ahead of itself), the
    default grid, and more workgroups than tiles: records, entry headers and counts are the oracle's whatever the grid."""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    monkeypatch.setenv("RAFTQ_WIRE_WGS", str(wgs))
    monkeypatch.setenv("RAFTQ_WIRE_TILE", str(tile))  # frames per tile = threads per decoder workgroup (256 is the default)
    rng = np.random.default_rng(1900 + wgs)
    n = 65536
    m, e, pool = _wiregen.random_msgs(rng, n, big_every=0, ent_frac=0.15)
    s, off = W.wire_encode(m, e, pool)
    for k in rng.integers(0, n, 300):
        s[int(off[k]) + 8 + int(rng.integers(0, 6))] ^= 0x5B
    wm, we, wbad = W.wire_decode(s, off)
    assert wbad > 0 and len(we) > 1000
    with WireEngine(4096, 5, self_peer=0) as eng:
        ps, po = pinned_copy(s), pinned_copy(off)
        dm, de = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
        for rep in range(4):  # consecutive calls share the control block: tickets and epochs carry over
            dm[:] = np.zeros(1, W.WIRE_MSG_DT)[0]
            gm, ge, gbad = eng.wire_decode(ps, po, msgs=dm, ents=de)
            assert gbad == wbad
            _same(gm, wm, "msgs")
            _same(ge, we, "ents")


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("wgs", [7, 0])
def test_streaming_codecs_without_readers(wgs, tile, monkeypatch):
    """liveness of a codec call is structural, not a property of the dispatcher.  RAFTQ_WIRE_READERS=0
    launches every streaming codec with NO reader workgroups: nobody is ever going to claim a chunk, so every worker finds its
    chunk unclaimed and the ticket still, takes the reader's role itself (raftq_wire_kernels.hpp feed_poll), and the call
    completes -- byte-identical to the oracle, bench-sized, all four codecs, a few workers walking many tiles and the default
    grid.  Before round 6 this launch shape spun for a second per wait and failed with RAFTQ_EHIP."""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    monkeypatch.setenv("RAFTQ_WIRE_READERS", "0")
    monkeypatch.setenv("RAFTQ_WIRE_TILE", str(tile))
    if wgs:
        monkeypatch.setenv("RAFTQ_WIRE_WGS", str(wgs))
    rng = np.random.default_rng(6100 + wgs + tile)
    n = 65536 if wgs == 0 else 20000
    m, e, pool = _wiregen.random_msgs(rng, n, big_every=9000, ent_frac=0.15)
    s, off = W.wire_encode(m, e, pool)
    wm, we, wbad = W.wire_decode(s, off)
    r, rpool = _wiregen.random_wal(rng, n, max_payload=200, big_every=7000)
    wal, wal_off, wal_last = W.wal_encode(r, rpool, 0xC0FFEE)
    wr, wnv, wl = W.wal_decode(wal, wal_off, 0xC0FFEE)
    with WireEngine(4096, 5, self_peer=0) as eng:
        ps, po = pinned_copy(s), pinned_copy(off)
        dm, de = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
        pm, pe, pp = pinned_copy(m), pinned_copy(e), pinned_copy(_wiregen_u8(pool))
        out, ooff = pinned_empty(len(s) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
        pr, prp = pinned_copy(r), pinned_copy(_wiregen_u8(rpool))
        wout, woff = pinned_empty(len(wal) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
        recs = pinned_empty(n, W.WAL_REC_DT)
        pw, pwo = pinned_copy(wal), pinned_copy(wal_off)
        for rep in range(3):  # consecutive calls share the control block: the chunk ticket's base carries over
            dm[:] = np.zeros(1, W.WIRE_MSG_DT)[0]
            gm, ge, gbad = eng.wire_decode(ps, po, msgs=dm, ents=de)
            assert gbad == wbad
            _same(gm, wm, "msgs")
            _same(ge, we, "ents")
            out[:] = 0xEE
            got, goff = eng.wire_encode(pm, pe, pp, out=out, off=ooff)
            assert got.tobytes() == s.tobytes() and np.array_equal(goff, off) and bytes(out[len(s):]) == b"\xee" * 64
            got, goff, glast = eng.wal_encode(pr, prp, 0xC0FFEE, out=wout, off=woff)
            assert got.tobytes() == wal.tobytes() and np.array_equal(goff, wal_off) and glast == wal_last
            gr, gnv, gl = eng.wal_decode(pw, pwo, 0xC0FFEE, recs=recs)
            assert (gnv, gl) == (wnv, wl)
            _same(gr, wr, "recs")
    # and with the readers back the very next call of a fresh handle is the ordinary path
    monkeypatch.delenv("RAFTQ_WIRE_READERS")
    with WireEngine(4096, 5, self_peer=0) as eng:
        gm, ge, gbad = eng.wire_decode(pinned_copy(s), pinned_copy(off), msgs=pinned_empty(n, W.WIRE_MSG_DT), ents=pinned_empty(len(we) + 1, W.WIRE_ENT_DT))
        assert gbad == wbad
        _same(gm, wm, "msgs")


def test_streaming_codecs_of_many_handles_resident_together():
    """Several handles' one-kernel codec calls on the device AT ONCE (the nodes of one process: bench.py's node legs, any host
    with a handle per shard): six launches of hundreds of workgroups each (a decoder workgroup held 102 KB of LDS until round 6,
    one per CU; 34 KB now) oversubscribe the chip and a launch's workgroups become resident XCD by XCD, late.  Every wait inside a launch must be for
    something a RUNNING workgroup has claimed (tiles by ticket; the readers' chunks by ticket too, round 5 -- they used to belong
    to reader c % readers, resident or not, and two launches could wait for each other's readers until the bounded waits gave
    up: `a workgroup waited a second for its predecessor's tile`).  Six threads, a handle each, decode + encode in a loop:
    every call's results are the oracle's."""
    import threading

    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    K, n, reps = 6, 32768, 12
    rng = np.random.default_rng(4242)
    work = []
    for k in range(K):
        m, e, pool = _wiregen.random_msgs(rng, n, big_every=0, ent_frac=0.15)
        s, off = W.wire_encode(m, e, pool)
        wm, we, wbad = W.wire_decode(s, off)
        work.append((m, e, pool, s, off, wm, we, wbad))
    engines = [WireEngine(4096, 5, self_peer=0) for _ in range(K)]
    errors = []
    gate = threading.Barrier(K)

    def run(k):
        try:
            m, e, pool, s, off, wm, we, wbad = work[k]
            eng = engines[k]
            ps, po = pinned_copy(s), pinned_copy(off)
            dm, de = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
            pm, pe, pp = pinned_copy(m), pinned_copy(e), pinned_copy(_wiregen_u8(pool))
            out, ooff = pinned_empty(len(s) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
            gate.wait()
            for rep in range(reps):
                gm, ge, gbad = eng.wire_decode(ps, po, msgs=dm, ents=de)
                assert gbad == wbad
                _same(gm, wm, "msgs")
                _same(ge, we, "ents")
                got, goff = eng.wire_encode(pm, pe, pp, out=out, off=ooff)
                assert got.tobytes() == s.tobytes() and np.array_equal(goff, off)
        except BaseException as ex:  # noqa: BLE001
            errors.append("handle %d: %r" % (k, ex))
            gate.abort()

    ths = [threading.Thread(target=run, args=(k,)) for k in range(K)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for eng in engines:
        eng.close()
    assert not errors, errors[:2]


@pytest.mark.parametrize("wgs", [3, 64, 208])
def test_streaming_encode_at_bench_size(wgs, monkeypatch):
    """The one-kernel form of raftq_wire_encode at the bench's size -- 65,536 messages, a MsgApp share with 1-3 entries of
    payload -- for a few workgroups walking many tiles, and the default grid: the stream and its offsets are the oracle's."""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    monkeypatch.setenv("RAFTQ_WIRE_WGS", str(wgs))
    rng = np.random.default_rng(2900 + wgs)
    n = 65536
    m, e, pool = _wiregen.random_msgs(rng, n, big_every=9000, ent_frac=0.15)
    want, want_off = W.wire_encode(m, e, pool)
    with WireEngine(4096, 5, self_peer=0) as eng:
        pm, pe, pp = pinned_copy(m), pinned_copy(e), pinned_copy(_wiregen_u8(pool))
        out, off = pinned_empty(len(want) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
        for rep in range(3):
            out[:] = 0xEE
            got, goff = eng.wire_encode(pm, pe, pp, out=out, off=off)
            assert np.array_equal(goff, want_off)
            assert got.tobytes() == want.tobytes() and bytes(out[len(want):]) == b"\xee" * 64


@pytest.mark.parametrize("wgs", [3, 208])
def test_streaming_wal_codecs_at_bench_size(wgs, monkeypatch):
    """The one-kernel forms of raftq_wal_encode / _decode at the bench's size: 65,536 records (entries with payloads -- a few
    long enough for the wave-cooperative CRC --, hard states, a crcType record in mid-segment), the CRC chain crossing every
    tile boundary; then the same bytes with damage: first bad record, the chain's value there, per-record flags."""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    monkeypatch.setenv("RAFTQ_WIRE_WGS", str(wgs))
    rng = np.random.default_rng(3900 + wgs)
    n = 65536
    r, pool = _wiregen.random_wal(rng, n, max_payload=200, big_every=7000)
    r["kind"][40000] = W.WAL_CRC
    for k in ("group", "term", "index", "vote", "data_len", "data_off", "entry_type"):
        r[k][40000] = 0
    want, want_off, want_last = W.wal_encode(r, pool, 0xC0FFEE)
    with WireEngine(4096, 5, self_peer=0) as eng:
        pr, pp = pinned_copy(r), pinned_copy(_wiregen_u8(pool))
        out, off = pinned_empty(len(want) + 64, np.uint8), pinned_empty(n + 1, np.uint64)
        for rep in range(2):
            out[:] = 0xEE
            got, goff, glast = eng.wal_encode(pr, pp, 0xC0FFEE, out=out, off=off)
            assert np.array_equal(goff, want_off) and glast == want_last
            assert got.tobytes() == want.tobytes() and bytes(out[len(want):]) == b"\xee" * 64
        recs = pinned_empty(n, W.WAL_REC_DT)
        for damage in (None, 50001, 17):
            s = want.copy()
            if damage is not None:
                s[int(want_off[damage]) + 9] ^= 0x10
            wr, wnv, wl = W.wal_decode(s, want_off, 0xC0FFEE)
            gr, gnv, gl = eng.wal_decode(pinned_copy(s), pinned_copy(want_off), 0xC0FFEE, recs=recs)
            assert (gnv, gl) == (wnv, wl) and (damage is None or gnv <= damage)
            _same(gr, wr, "recs")


def _wiregen_u8(pool):
    return np.ascontiguousarray(np.frombuffer(bytes(pool), np.uint8) if not isinstance(pool, np.ndarray) else pool.view(np.uint8))


def test_scan_frames_host(eng):
    from raftsql_amd import wire

    rng = np.random.default_rng(5)
    m, e, pool = _wiregen.random_msgs(rng, 50)
    s, off = W.wire_encode(m, e, pool)
    for cut in (len(s), len(s) - 1, int(off[20]) + 3, int(off[20]) + 8, 0, 7):
        want, wused = W.scan_frames(s[:cut], big_endian=True)
        got, used = wire.scan_frames(s[:cut], big_endian=True)
        assert used == wused and np.array_equal(got, want)
    got, used = wire.scan_frames(s, big_endian=True, cap=10)
    assert len(got) == 11 and used == off[10]


def test_roundtrip_at_size(eng):
    """200k messages: decode(encode(x)) == x field for field, payload bytes included."""
    rng = np.random.default_rng(77)
    m, e, pool = _wiregen.random_msgs(rng, 200_000, ent_frac=0.25, max_payload=120)
    s, off = eng.wire_encode(m, e, pool)
    assert off[-1] == len(s)
    mm, ee, bad = eng.wire_decode(s, off)
    assert bad == 0 and np.all(mm["flags"] == W.F_GROUP)
    for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject", "to", "n_ents", "ent_first"):
        assert np.array_equal(mm[k], m[k]), k
    for k in ("term", "index", "data_len", "type"):
        assert np.array_equal(ee[k], e[k]), k
    # payload bytes: a vectorised gather of every payload byte from both sides
    ln = e["data_len"].astype(np.int64)
    tot = int(ln.sum())
    rel = np.arange(tot) - np.repeat(np.cumsum(ln) - ln, ln)
    src = np.repeat(e["data_off"].astype(np.int64), ln) + rel
    dst = np.repeat(ee["data_off"].astype(np.int64), ln) + rel
    assert np.array_equal(s[dst], pool[src])
    # the frames found from the length words alone are the ones encode reported
    from raftsql_amd import wire

    off2, used = wire.scan_frames(s, big_endian=True)
    assert used == len(s) and np.array_equal(off2, off)


# ---- Step from the wire ---------------------------------------------------------------------------

def _step_traffic(rng, n, n_groups, n_peers):
    m = np.zeros(n, W.WIRE_MSG_DT)
    m["group"] = rng.integers(0, n_groups, n)
    m["type"] = rng.choice([3, 4, 5, 6, 8, 9], n)
    m["term"] = rng.integers(1, 4, n)
    m["from"] = rng.integers(1, n_peers, n)
    m["to"] = 0
    m["index"] = rng.integers(0, 50, n)
    m["log_term"] = rng.integers(0, 4, n)
    m["commit"] = rng.integers(0, 30, n)
    m["reject"] = rng.integers(0, 2, n)
    m["reject_hint"] = rng.integers(0, 50, n)
    return m


@pytest.mark.usefixtures("stage_mode")
def test_step_from_wire_equals_step_from_records():
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError
    from raftsql_amd.wire import WireEngine

    G, N = 2048, 5
    rng = np.random.default_rng(31)
    with WireEngine(G, N, 0) as a, S.NodeEngine(G, N, 0) as b:
        for rnd in range(6):
            m = _step_traffic(rng, 5000, G, N)
            e = np.zeros(0, W.WIRE_ENT_DT)
            pool = np.zeros(1, np.uint8)
            if rnd % 2:  # MsgApp frames carry entries; Step reads only the header
                app = np.nonzero(m["type"] == 3)[0]
                m["n_ents"][app] = 2
                m["ent_first"][app] = np.arange(len(app)) * 2
                e = np.zeros(2 * len(app), W.WIRE_ENT_DT)
                e["term"], e["index"], e["data_len"] = 3, np.arange(len(e)), 5
                e["data_off"] = np.arange(len(e)) * 5
                pool = rng.integers(0, 256, 5 * len(e) + 1, dtype=np.uint8)
            s, off = W.wire_encode(m, e, pool)
            a.step_submit_wire(s, off)
            got, touched = a.step_collect()
            rec = np.zeros(len(m), S.MSG_DT)
            for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject"):
                rec[k] = m[k]
            want, wtouched = b.step_batch(rec)
            assert touched == wtouched
            _same(got, want, "step results")
            wm, we, _ = W.wire_decode(s, off)
            _same(a.step_wire_msgs(), wm, "decoded records of the batch")
            _same(a.step_wire_entries(), we, "decoded entries of the batch")
        na, nb = a.read_node(), b.read_node()
        for k in na:
            assert np.array_equal(na[k], nb[k]), k
        # a frame that does not parse fails the whole batch, nothing is applied
        m = _step_traffic(rng, 100, G, N)
        s, off = W.wire_encode(m)
        s = s.copy()
        s[int(off[40]) + 8] = 0x0B
        a.step_submit_wire(s, off)
        with pytest.raises(RaftqError):
            a.step_collect()
        assert a.step_wire_msgs()[40]["flags"] == W.F_MALFORMED  # which one: the decoded records say
        # so does one addressed to no peer of the cluster, or a snapshot message
        for field, val in (("to", 9), ("type", 7)):
            mm = m.copy()
            mm[field][3] = val
            s, off = W.wire_encode(mm)
            a.step_submit_wire(s, off)
            with pytest.raises(RaftqError):
                a.step_collect()
        n2 = a.read_node()
        for k in na:
            assert np.array_equal(na[k], n2[k]), k
        # two wire batches in flight, collected in order
        m1, m2 = _step_traffic(rng, 3000, G, N), _step_traffic(rng, 3000, G, N)
        for mm in (m1, m2):
            s, off = W.wire_encode(mm)
            a.step_submit_wire(s, off)
        for mm in (m1, m2):
            got, _ = a.step_collect()
            rec = np.zeros(len(mm), S.MSG_DT)
            for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject"):
                rec[k] = mm[k]
            want, _ = b.step_batch(rec)
            _same(got, want, "pipelined step results")


# ---- WAL ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("seed,n,big,prev", [(141, 1, 0, 0), (142, 300, 9, 0), (143, 300, 2, 0xDEADBEEF), (144, 5000, 50, 0),
                                             (145, 65, 1, 1)])
def test_wal_encode_parity(eng, seed, n, big, prev):
    rng = np.random.default_rng(seed)
    r, pool = _wiregen.random_wal(rng, n, big_every=big, head=(prev == 0 and n >= 3))
    want, want_off, want_last = W.wal_encode(r, pool, prev)
    got, off, last = eng.wal_encode(r, pool, prev)
    assert np.array_equal(off, want_off) and last == want_last
    assert got.tobytes() == want.tobytes()


def test_wal_encode_rejects_bad_input(eng):
    from raftsql_amd.engine import RaftqError

    r, pool = _wiregen.random_wal(np.random.default_rng(10), 50)
    for kind in (0, 6, 255):
        bad = r.copy()
        bad["kind"][20] = kind
        with pytest.raises(RaftqError):
            eng.wal_encode(bad, pool, 0)
    j = int(np.nonzero(r["data_len"])[0][0])
    bad = r.copy()
    bad["data_off"][j] = len(pool) - 1
    bad["data_len"][j] = 2
    with pytest.raises(RaftqError):
        eng.wal_encode(bad, pool, 0)


@pytest.mark.parametrize("seed,big", [(151, 11), (152, 2)])
def test_wal_decode_parity(eng, seed, big):
    rng = np.random.default_rng(seed)
    r, pool = _wiregen.random_wal(rng, 400, big_every=big)
    r["kind"][200] = W.WAL_CRC  # a segment cut in mid-stream
    for k in ("group", "term", "index", "vote", "data_len", "data_off", "entry_type"):
        r[k][200] = 0
    out, off, last = W.wal_encode(r, pool, 0)

    def both(data, offs, prev):
        wr, wnv, wl = W.wal_decode(data, offs, prev)
        gr, gnv, gl = eng.wal_decode(data, offs, prev)
        assert (gnv, gl) == (wnv, wl)
        _same(gr, wr, "recs")
        return gr, gnv, gl

    rr, nv, lc = both(out, off, 0)
    assert nv == len(r) and lc == last
    both(out, off, 12345)  # wrong seed
    for j in rng.choice(np.nonzero(r["data_len"] > 0)[0], 12, replace=False):  # a flipped payload bit
        bad = out.copy()
        bad[int(rr[j]["data_off"]) + int(rng.integers(0, r[j]["data_len"]))] ^= 1 << int(rng.integers(0, 8))
        _, nv, _ = both(bad, off, 0)
        assert nv == j
    for _ in range(12):  # a flipped bit anywhere: header fields, crc field, length words
        bad = out.copy()
        bad[int(rng.integers(0, len(out)))] ^= 1 << int(rng.integers(0, 8))
        both(bad, off, 0)
    bad = out.copy()
    bad[int(off[10]) + 8] = 0x0B  # does not parse
    _, nv, _ = both(bad, off, 0)
    assert nv == 10
    bad = out.copy()
    bad[int(off[200]) + 11] ^= 0x40  # the re-seeding record's own crc is wrong: it and what follows mismatch
    both(bad, off, 0)
    cut = int(off[300]) + 5  # a torn tail
    o3, used = W.scan_frames(out[:cut], big_endian=False)
    _, nv, lc = both(out[:cut], o3, 0)
    assert nv == 300
    rest, _, last3 = eng.wal_encode(r[300:], pool, lc)  # appending continues the chain
    assert rest.tobytes() == out[int(off[300]):].tobytes() and last3 == last


def test_wal_decode_fuzz(eng):
    rng = np.random.default_rng(161)
    r, pool = _wiregen.random_wal(rng, 2000, max_payload=80)
    out, off, _ = W.wal_encode(r, pool, 0)
    s = out.copy()
    n_mut = len(s) // 60
    s[rng.integers(0, len(s), n_mut)] = rng.integers(0, 256, n_mut, dtype=np.uint8)
    for i in rng.choice(len(r), len(r) * 9 // 10, replace=False):
        a, b = int(off[i]), int(off[i + 1])
        s[a:a + 8] = np.frombuffer((b - a - 8).to_bytes(8, "little"), np.uint8)
    wr, wnv, wl = W.wal_decode(s, off, 0)
    gr, gnv, gl = eng.wal_decode(s, off, 0)
    assert (gnv, gl) == (wnv, wl)
    _same(gr, wr, "recs")


def test_wal_chain_properties_at_size(eng):
    """300k records (with 4-20 KB payloads sprinkled in): decode(encode(x)) validates completely and
    returns x; the chain splits anywhere -- encoding [0, k) then [k, n) seeded with the first half's
    last_crc gives the same bytes as one call; a corrupted byte is found at its record."""
    rng = np.random.default_rng(171)
    n = 300_000
    r, pool = _wiregen.random_wal(rng, n, max_payload=100, big_every=997)
    out, off, last = eng.wal_encode(r, pool, 0)
    assert off[-1] == len(out)
    rr, nv, lc = eng.wal_decode(out, off, 0)
    assert nv == n and lc == last
    for k in ("group", "term", "index", "vote", "data_len", "kind", "entry_type"):
        assert np.array_equal(rr[k], r[k]), k
    ln = r["data_len"].astype(np.int64)
    tot = int(ln.sum())
    rel = np.arange(tot) - np.repeat(np.cumsum(ln) - ln, ln)
    assert np.array_equal(out[np.repeat(rr["data_off"].astype(np.int64), ln) + rel],
                          pool[np.repeat(r["data_off"].astype(np.int64), ln) + rel])
    k = 123_457
    a, _, la = eng.wal_encode(r[:k], pool, 0)
    b, _, lb = eng.wal_encode(r[k:], pool, la)
    assert lb == last and a.tobytes() == out[:int(off[k])].tobytes() and b.tobytes() == out[int(off[k]):].tobytes()
    # the oracle agrees on a prefix it can do quickly
    want, woff, wlast = W.wal_encode(r[:20000], pool, 0)
    assert want.tobytes() == out[:int(off[20000])].tobytes()
    j = int(np.nonzero(r["data_len"] > 4000)[0][3])
    bad = out.copy()
    bad[int(rr[j]["data_off"]) + 2000] ^= 0x10
    _, nv, _ = eng.wal_decode(bad, off, 0)
    assert nv == j


def test_wal_hand_vector(eng):
    r = np.zeros(5, W.WAL_REC_DT)
    r["kind"] = [W.WAL_CRC, W.WAL_METADATA, W.WAL_SNAPSHOT, W.WAL_ENTRY, W.WAL_STATE]
    r[3]["term"], r[3]["index"], r[3]["data_len"] = 1, 1, 1
    r[4]["term"], r[4]["vote"], r[4]["index"] = 1, 2, 1
    out, off, last = eng.wal_encode(r, b"a", 0)
    snap = bytes.fromhex("08001000")
    ent = bytes.fromhex("080010011801" "220161" "2800")
    st = bytes.fromhex("0801100218012000")
    c1 = W.crc32c(snap)
    c2 = W.crc32c(ent, c1)
    c3 = W.crc32c(st, c2)
    vi = _wiregen._varint

    def le(b):
        return len(b).to_bytes(8, "little") + b

    want = b"".join(le(x) for x in (
        bytes.fromhex("08041000"),
        bytes.fromhex("08011000"),
        bytes.fromhex("0805") + b"\x10" + vi(c1) + b"\x1a\x04" + snap,
        bytes.fromhex("0802") + b"\x10" + vi(c2) + b"\x1a\x0b" + ent,
        bytes.fromhex("0803") + b"\x10" + vi(c3) + b"\x1a\x08" + st))
    assert out.tobytes() == want and last == c3 and off[-1] == len(want)


# ---- committed fixtures ---------------------------------------------------------------------------

def test_golden_fixtures(eng):
    g = json.load(open(GOLD))
    w = g["wire"]
    m = np.frombuffer(bytes.fromhex(w["msgs"]), W.WIRE_MSG_DT)
    e = np.frombuffer(bytes.fromhex(w["ents"]), W.WIRE_ENT_DT)
    pool = bytes.fromhex(w["pool"])
    s, off = eng.wire_encode(m, e, pool)
    assert s.tobytes().hex() == w["stream"] and list(map(int, off)) == w["frame_off"]
    mm, ee, bad = eng.wire_decode(s, off)
    assert bad == 0 and mm.tobytes().hex() == w["decoded_msgs"] and ee.tobytes().hex() == w["decoded_ents"]
    nc = g["wire_noncanonical"]
    mm, ee, bad = eng.wire_decode(bytes.fromhex(nc["stream"]), np.array(nc["frame_off"], np.uint64))
    assert bad == nc["n_malformed"] and mm.tobytes().hex() == nc["decoded_msgs"] and ee.tobytes().hex() == nc["decoded_ents"]
    a = g["wal"]
    r = np.frombuffer(bytes.fromhex(a["recs"]), W.WAL_REC_DT)
    out, off, last = eng.wal_encode(r, bytes.fromhex(a["pool"]), a["prev_crc"])
    assert out.tobytes().hex() == a["bytes"] and last == a["last_crc"] and list(map(int, off)) == a["frame_off"]
    rr, nv, lc = eng.wal_decode(out, off, a["prev_crc"])
    assert nv == len(r) and lc == last and rr.tobytes().hex() == a["decoded_recs"]
    # the CRC fixtures, through the WAL path: one metadata record whose Data is the vector, seeded
    for c in g["crc32c"]:
        data = bytes.fromhex(c["data"])
        rec = np.zeros(1, W.WAL_REC_DT)
        rec["kind"], rec["data_len"] = W.WAL_METADATA, len(data)
        _, _, last = eng.wal_encode(rec, data, c["seed"])
        assert last == c["crc"], c


@pytest.mark.usefixtures("stage_mode")
def test_step_from_frames_staged_in_place():
    """raftq_step_stage_wire: the frames are written straight into the arrays the library hands out (device memory behind
    a large BAR) and decoded where they lie -- same results as the copying form and as Step from the decoded records,
    three batches in flight, entries fetched afterwards; asking for a slot's arrays again ends the validity of the
    decoded records of the batch that lived there."""
    from raftsql_amd import step as S
    from raftsql_amd.engine import RaftqError
    from raftsql_amd.wire import WireEngine

    G, N = 3000, 5
    rng = np.random.default_rng(47)
    with WireEngine(G, N, 0) as a, S.NodeEngine(G, N, 0) as b:
        def make(n, with_entries):
            m = _step_traffic(rng, n, G, N)
            e, pool = np.zeros(0, W.WIRE_ENT_DT), np.zeros(1, np.uint8)
            if with_entries:
                app = np.nonzero(m["type"] == 3)[0]
                m["n_ents"][app] = 2
                m["ent_first"][app] = np.arange(len(app)) * 2
                e = np.zeros(2 * len(app), W.WIRE_ENT_DT)
                e["term"], e["index"], e["data_len"] = 3, np.arange(len(e)), 7
                e["data_off"] = np.arange(len(e)) * 7
                pool = rng.integers(0, 256, 7 * len(e) + 1, dtype=np.uint8)
            s, off = W.wire_encode(m, e, pool)
            rec = np.zeros(n, S.MSG_DT)
            for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject"):
                rec[k] = m[k]
            return s, off, rec

        def submit_staged(s, off, slack=0):
            n = len(off) - 1
            so, ss = a.step_stage_wire(n + slack, len(s) + 3 * slack)  # capacity may exceed what is used
            so[: n + 1] = off
            ss[: len(s)] = s
            a.step_submit_wire_staged(so, ss, n, len(s))

        pending = []
        for it in range(15):
            s, off, rec = make(int(rng.integers(1, 4000)), it % 2 == 0)
            if it % 5 == 4:
                a.step_submit_wire(s, off)  # the copying form in between
            else:
                submit_staged(s, off, slack=int(rng.integers(0, 50)))
            pending.append((b.step_batch(rec), s, off))
            if len(pending) == 3:
                (want, wt), ws, woff = pending.pop(0)
                got, touched = a.step_collect()
                assert touched == wt
                _same(got, want, "step results (staged frames)")
                if it % 3 == 0:  # the decoder's second pass, from the staged bytes
                    wm, we, _ = W.wire_decode(ws, woff)
                    _same(a.step_wire_msgs(), wm, "decoded records")
                    _same(a.step_wire_entries(), we, "decoded entries")
        while pending:
            (want, wt), ws, woff = pending.pop(0)
            got, touched = a.step_collect()
            assert touched == wt
            _same(got, want, "step results (staged frames)")
        na, nb = a.read_node(), b.read_node()
        for k in na:
            assert np.array_equal(na[k], nb[k]), k
        # the last collected batch's records are readable ... until its slot's arrays are asked for again
        s, off, rec = make(500, True)
        submit_staged(s, off)
        a.step_collect()
        b.step_batch(rec)
        assert len(a.step_wire_msgs()) == 500
        for _ in range(3):  # three stagings later the rotation is back at that slot
            s2, off2, rec2 = make(10, False)
            submit_staged(s2, off2)
            a.step_collect()
            b.step_batch(rec2)
        assert len(a.step_wire_msgs()) == 10  # the newest batch's
        a.step_stage_wire(10, 100)  # hands out the slot after the newest one: nothing changes for the newest batch
        assert len(a.step_wire_msgs()) == 10
        # a malformed staged frame fails the batch like any other
        s, off, _ = make(100, False)
        s = s.copy()
        s[int(off[17]) + 8] = 0x0B
        submit_staged(s, off)
        with pytest.raises(RaftqError):
            a.step_collect()
        assert a.step_wire_msgs()[17]["flags"] == W.F_MALFORMED
        na, nb = a.read_node(), b.read_node()
        for k in na:
            assert np.array_equal(na[k], nb[k]), k


# ---- raftq_step_frames: a node's inbound half-turn as one submission ---------------------------------------------------

def _node_frames(rng, n, st, self_peer):
    """n frames as a node of st's cluster might receive them in one turn, consistent with st (an oracle NodeState): MsgApps
    with entries (half of them exactly on their group's tail), MsgProps with entries, the payload-free kinds, and what a node
    has to shrug off -- frames for another slot, from no peer, for no group, of kinds a peer never sends, and garbage bytes.
    -> (stream, frame_off)"""
    G, N = st.G, st.N
    m = np.zeros(n, W.WIRE_MSG_DT)
    g = rng.integers(0, G, n)
    m["group"] = g
    m["type"] = rng.choice([2, 3, 4, 5, 6, 8, 9], n, p=[0.08, 0.30, 0.30, 0.06, 0.06, 0.10, 0.10])
    m["term"] = np.maximum(1, st.term[g].astype(np.int64) + rng.choice([-1, 0, 0, 0, 0, 1], n)).astype(np.uint64)
    m["from"] = (self_peer + 1 + rng.integers(0, N - 1, n)) % N
    m["to"] = self_peer
    li = st.last_index[g].astype(np.int64)
    m["index"] = np.maximum(0, li + rng.integers(-2, 3, n)).astype(np.uint64)
    m["log_term"] = np.maximum(0, st.last_term[g].astype(np.int64) + rng.integers(-1, 2, n)).astype(np.uint64)
    m["commit"] = np.maximum(0, li + rng.integers(-3, 3, n)).astype(np.uint64)
    m["reject"] = rng.random(n) < 0.2
    m["reject_hint"] = m["index"]
    tail = (m["type"] == 3) & (rng.random(n) < 0.6)
    m["index"] = np.where(tail, st.last_index[g], m["index"])
    m["log_term"] = np.where(tail, st.last_term[g], m["log_term"])
    carries = ((m["type"] == 3) & (rng.random(n) < 0.8)) | (m["type"] == 2)
    k = np.where(carries, rng.choice([1, 1, 2, 3, 6], n), 0)  # (6: more than a decoder lane keeps)
    m["n_ents"] = k
    m["ent_first"] = np.concatenate([[0], np.cumsum(k)[:-1]])
    ne = int(k.sum())
    e = np.zeros(ne, W.WIRE_ENT_DT)
    owner = np.repeat(np.arange(n), k)
    e["term"] = np.maximum(m["log_term"][owner], m["term"][owner] - (rng.random(ne) < 0.5))
    e["index"] = m["index"][owner] + 1 + (np.arange(ne) - m["ent_first"][owner])
    e["data_len"] = rng.integers(0, 40, ne)
    e["data_off"] = np.concatenate([[0], np.cumsum(e["data_len"])[:-1]]) if ne else 0
    pool = rng.integers(0, 256, int(e["data_len"].sum()) + 1, dtype=np.uint8)
    # what a node has to shrug off
    odd = rng.random(n)
    m["to"] = np.where(odd < 0.03, (self_peer + 1) % N, m["to"])
    m["from"] = np.where((odd >= 0.03) & (odd < 0.05), N + 3, m["from"])
    m["group"] = np.where((odd >= 0.05) & (odd < 0.07), G + rng.integers(0, 1000, n), m["group"])
    m["type"] = np.where((odd >= 0.07) & (odd < 0.10), rng.choice([0, 1, 7, 10, 11, 200], n), m["type"])
    s, off = W.wire_encode(m, e, pool)
    s = s.copy()
    for i in np.nonzero((odd >= 0.10) & (odd < 0.12))[0]:  # garbage where the message's first key was
        s[int(off[i]) + 8] = 0x0B
    return s, off


def _node_filter(wm, we, G, N, self_peer, tail_appends):
    """what raftq_step_frames' decoder does to the decoded records (include/raftq_wire.h), restated -> (records as handed to
    the caller, the same as Step reads them)"""
    from raftsql_amd import step as S

    out = wm.copy()
    t = out["type"]
    kind_ok = np.isin(t, [2, 3, 4, 5, 6, 8, 9])
    skip = ((out["flags"] & W.F_MALFORMED) != 0) | ~kind_ok | (out["group"] >= G) | (out["from"] >= N) | (out["to"] != self_peer)
    hold = ~skip & (t == 2)
    app = ~skip & (t == 3)
    out["flags"] |= np.where(skip, S.MSGF_SKIP, 0).astype(np.uint8) | np.where(hold, S.MSGF_HOLD, 0).astype(np.uint8)
    out["flags"] |= np.where(app, S.MSGF_BARRIER | (S.MSGF_ENTRIES if tail_appends else 0), 0).astype(np.uint8)
    last = np.where(out["n_ents"] > 0, out["ent_first"].astype(np.int64) + out["n_ents"] - 1, 0)
    last_term = we["term"][np.minimum(last, max(len(we) - 1, 0))] if len(we) else np.zeros(len(out), np.uint64)
    out["reject_hint"] = np.where(app, np.where(out["n_ents"] > 0, last_term, 0), out["reject_hint"])
    rec = np.zeros(len(out), S.MSG_DT)
    for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject"):
        rec[k] = out[k]
    rec["_pad"][:, 0], rec["_pad"][:, 1] = out["to"], out["flags"]
    rec["_resv"] = out["n_ents"]  # (the oracle reads the count where a caller's record keeps it)
    return out, rec


@pytest.mark.parametrize("walk", ["lists", "sort"])
@pytest.mark.parametrize("tail_appends", [True, False])
def test_step_frames_equals_decode_filter_step(oracle, walk, tail_appends, monkeypatch):
    """raftq_step_frames against its parts, all from the oracle: decode, the node's checks (restated above), Step over every
    frame with the flags the checks set -- decoded records, entry headers, every result byte, the state after."""
    from raftsql_amd import step as S
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine
    from tests import _stepgen

    if walk == "sort":
        monkeypatch.setenv("RAFTQ_STEP_WALK", "sort")
    G, N, me = 3000, 5, 2
    rng = np.random.default_rng(515 + tail_appends)
    st = _stepgen.random_state(rng, G, N, self_peer=me)
    with WireEngine(G, N, me) as e:
        _stepgen.load_engine(e, st)
        for it, n in enumerate([1, 255, 257, 4000, 9000, 700]):
            s, off = _node_frames(rng, n, st, me)
            wm, we, _ = W.wire_decode(s, off)
            want_m, rec = _node_filter(wm, we, G, N, me, tail_appends)
            want_o = st.step_batch(rec)
            ps, po = pinned_copy(np.ascontiguousarray(s)), pinned_copy(np.ascontiguousarray(off, np.uint64))
            msgs, ents = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
            gm, ge, go, c = e.step_frames(ps, po, msgs, ents, tail_appends=tail_appends)
            assert (c.n_msgs, c.n_ents, c.bytes) == (n, len(we), len(s)) and c.n_malformed == int(((wm["flags"] & W.F_MALFORMED) != 0).sum())
            _same(gm, want_m, f"records, call {it}")
            _same(ge, we, f"entry headers, call {it}")
            _same(go, want_o, f"results, call {it}")
            types = set(int(t) for t in go["type"])
            if n >= 4000:
                assert {S.OUT_SKIPPED, S.OUT_HELD, S.OUT_DEFERRED} <= types and (not tail_appends or S.OUT_APPENDED in types)
        _stepgen.assert_same_state(e, st)
        # fewer entry headers than the frames hold: not an error here (the frames HAVE been stepped), the count says so
        s, off = _node_frames(rng, 2000, st, me)
        wm, we, _ = W.wire_decode(s, off)
        _, rec = _node_filter(wm, we, G, N, me, tail_appends)
        want_o = st.step_batch(rec)
        ps, po = pinned_copy(np.ascontiguousarray(s)), pinned_copy(np.ascontiguousarray(off, np.uint64))
        msgs, few = pinned_empty(2000, W.WIRE_MSG_DT), pinned_empty(64, W.WIRE_ENT_DT)
        few[:] = 0
        gm, ge, go, c = e.step_frames(ps, po, msgs, few, tail_appends=tail_appends)
        assert c.n_ents == len(we) > 64 and len(ge) == 64
        _same(go, want_o, "results with a short entry array")
        _same(ge, we[:64], "the entry headers that fit")
        _stepgen.assert_same_state(e, st)


@pytest.mark.parametrize("N,me", [(2, 1), (3, 0), (4, 3), (7, 2), (9, 8)])
def test_step_frames_for_every_cluster_size(oracle, N, me):
    """the same comparison for 2 ... 9 peers and every kind of own slot (the vote word is 16 or 32 bits wide, the quorum even
    or odd, `to` / `from` checks against other bounds)"""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine
    from tests import _stepgen

    G = 2000
    rng = np.random.default_rng(7000 + N)
    st = _stepgen.random_state(rng, G, N, self_peer=me)
    with WireEngine(G, N, me) as e:
        _stepgen.load_engine(e, st)
        for it, n in enumerate([3000, 513, 6000]):
            s, off = _node_frames(rng, n, st, me)
            wm, we, _ = W.wire_decode(s, off)
            want_m, rec = _node_filter(wm, we, G, N, me, it != 1)
            want_o = st.step_batch(rec)
            msgs, ents = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
            gm, ge, go, c = e.step_frames(pinned_copy(np.ascontiguousarray(s)), pinned_copy(np.ascontiguousarray(off, np.uint64)), msgs, ents,
                                          tail_appends=it != 1)
            _same(gm, want_m, f"records, call {it}")
            _same(ge, we, f"entry headers, call {it}")
            _same(go, want_o, f"results, call {it}")
        _stepgen.assert_same_state(e, st)


def test_step_frames_with_long_runs_goes_through_the_sorted_walk(oracle):
    """Hundreds of frames per group in one call: the list walk gives the batch up (a run longer than it takes) and the call
    replays it through the sorted walk -- the records the decoder left in HBM are read a second time, skipped and held frames
    included.  Same answers, same state; and the calls after it are back on the list walk."""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine
    from tests import _stepgen

    G, N, me = 40, 3, 1
    rng = np.random.default_rng(99)
    st = _stepgen.random_state(rng, G, N, self_peer=me)
    with WireEngine(G, N, me) as e:
        _stepgen.load_engine(e, st)
        for it, n in enumerate([6000, 300, 20000, 50]):
            s, off = _node_frames(rng, n, st, me)
            wm, we, _ = W.wire_decode(s, off)
            want_m, rec = _node_filter(wm, we, G, N, me, True)
            want_o = st.step_batch(rec)
            msgs, ents = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
            gm, ge, go, c = e.step_frames(pinned_copy(np.ascontiguousarray(s)), pinned_copy(np.ascontiguousarray(off, np.uint64)), msgs, ents)
            _same(gm, want_m, f"records, call {it}")
            _same(ge, we, f"entry headers, call {it}")
            _same(go, want_o, f"results, call {it}")
        _stepgen.assert_same_state(e, st)


def test_step_frames_refuses_what_it_cannot_stream(oracle):
    from raftsql_amd import _lib
    from raftsql_amd.engine import RaftqError, pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine
    from tests import _stepgen

    rng = np.random.default_rng(3)
    st = _stepgen.random_state(rng, 64, 3, 0)
    s, off = _node_frames(rng, 100, st, 0)
    ps, po = pinned_copy(np.ascontiguousarray(s)), pinned_copy(np.ascontiguousarray(off, np.uint64))
    msgs = pinned_empty(100, W.WIRE_MSG_DT)
    with WireEngine(64, 3, 0) as e:
        _stepgen.load_engine(e, st)
        with pytest.raises(RaftqError) as ei:  # pageable memory: the two calls are the caller's to make
            e.step_frames(np.ascontiguousarray(s), po, msgs)
        assert ei.value.code == _lib.RAFTQ_EINVAL
        e.step_submit(_stepgen.random_batch(rng, st, 10))
        with pytest.raises(RaftqError) as ei:  # a batch in flight
            e.step_frames(ps, po, msgs)
        assert ei.value.code == _lib.RAFTQ_ESTATE
        e.step_collect()
    with WireEngine(64, 3, 0, msg_flags=False) as e:  # a handle whose pad bytes are padding cannot be told what a frame is
        _stepgen.load_engine(e, st)
        with pytest.raises(RaftqError) as ei:
            e.step_frames(ps, po, msgs)
        assert ei.value.code == _lib.RAFTQ_ESTATE


def test_wal_encode_begin_end_is_one_submission_with_the_call_behind_it():
    """raftq_wal_encode_begin enqueues and returns; the wait of the raftq_wire_encode called next covers it; _end reports what
    raftq_wal_encode would have.  Bytes, offsets, CRC against the oracle; also with nothing in between (_end waits itself),
    with a refused batch (the error surfaces at _end), and begun twice."""
    from raftsql_amd import _lib
    from raftsql_amd.engine import RaftqError, pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    rng = np.random.default_rng(808)
    with WireEngine(4096, 5, self_peer=0) as e:
        for it, (nr, nm) in enumerate([(3000, 5000), (1, 1), (70000, 300), (257, 66000)]):
            recs, wpool = _wiregen.random_wal(rng, nr, head=it % 2 == 0)
            want_w, want_woff, want_crc = W.wal_encode(recs, wpool, prev_crc=77)
            m, ents, pool = _wiregen.random_msgs(rng, nm, ent_frac=0.2)
            want_s, want_soff = W.wire_encode(m, ents, pool)
            p_recs, p_wpool = pinned_copy(np.ascontiguousarray(recs)), pinned_copy(np.ascontiguousarray(_wiregen_u8(wpool)))
            w_out, w_off = pinned_empty(len(want_w) + 64, np.uint8), pinned_empty(len(recs) + 1, np.uint64)
            e.wal_encode_begin(p_recs, p_wpool, 77, w_out, w_off)
            if it == 2:  # the copying form (pageable arrays) in between: ordered behind the begun encode by the stream alone
                got_s, got_soff = e.wire_encode(m, ents, pool)
                assert np.array_equal(got_soff, want_soff) and got_s.tobytes() == want_s.tobytes(), it
            elif it != 1:  # (it == 1: nothing in between, _end makes the wait)
                pm, pe, pp = pinned_copy(np.ascontiguousarray(m)), pinned_copy(np.ascontiguousarray(ents)) if len(ents) else ents, pinned_copy(np.ascontiguousarray(_wiregen_u8(pool)))
                s_out, s_off = pinned_empty(len(want_s) + 64, np.uint8), pinned_empty(len(m) + 1, np.uint64)
                got_s, got_soff = e.wire_encode(pm, pe, pp, out=s_out, off=s_off)
                assert np.array_equal(got_soff, want_soff) and got_s.tobytes() == want_s.tobytes(), it
            nbytes, crc = e.wal_encode_end()
            assert (nbytes, crc) == (len(want_w), want_crc), it
            assert w_out[:nbytes].tobytes() == want_w.tobytes() and np.array_equal(w_off, want_woff), it
        # a refused batch: enqueued all the same, the verdict is _end's
        recs, wpool = _wiregen.random_wal(rng, 500)
        recs = recs.copy()
        recs["kind"][123] = 9
        p_recs, p_wpool = pinned_copy(np.ascontiguousarray(recs)), pinned_copy(np.ascontiguousarray(_wiregen_u8(wpool)))
        w_out = pinned_empty(1 << 20, np.uint8)
        e.wal_encode_begin(p_recs, p_wpool, 0, w_out)
        with pytest.raises(RaftqError) as ei:
            e.wal_encode_begin(p_recs, p_wpool, 0, w_out)  # one at a time
        assert ei.value.code == _lib.RAFTQ_ESTATE
        with pytest.raises(RaftqError) as ei:
            e.wal_encode_end()
        assert ei.value.code == _lib.RAFTQ_EINVAL
        with pytest.raises(RaftqError) as ei:
            e.wal_encode_end()  # nothing begun any more
        assert ei.value.code == _lib.RAFTQ_ESTATE
        with pytest.raises(RaftqError) as ei:  # pageable memory: raftq_wal_encode's business
            e.wal_encode_begin(np.ascontiguousarray(recs), p_wpool, 0, w_out)
        assert ei.value.code == _lib.RAFTQ_EINVAL
        good, gpool = _wiregen.random_wal(rng, 50)
        got, _, crc = e.wal_encode(good, gpool)  # and the handle is as good as before
        assert got.tobytes() == W.wal_encode(good, gpool)[0].tobytes()


def test_step_frames_at_bench_size(oracle):
    """bench.py's inbound half-turn: 65,536 frames for 1M x 5 groups in one call (256 tiles through ~208 workers, the Step
    kernels behind the decoder) -- every record, entry header and result against the oracle, the state after."""
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine
    from tests import _stepgen

    G, N, me = 1 << 20, 5, 0
    rng = np.random.default_rng(2026)
    st = _stepgen.random_state(rng, G, N, self_peer=me)
    with WireEngine(G, N, me) as e:
        _stepgen.load_engine(e, st)
        for it in range(2):
            n = 65536
            s, off = _node_frames(rng, n, st, me)
            wm, we, _ = W.wire_decode(s, off)
            want_m, rec = _node_filter(wm, we, G, N, me, True)
            want_o = st.step_batch(rec)
            msgs, ents = pinned_empty(n, W.WIRE_MSG_DT), pinned_empty(len(we) + 1, W.WIRE_ENT_DT)
            e.set_compact(it == 1)
            gm, ge, go, c = e.step_frames(pinned_copy(np.ascontiguousarray(s)), pinned_copy(np.ascontiguousarray(off, np.uint64)), msgs, ents)
            _same(gm, want_m, f"records, call {it}")
            _same(ge, we, f"entry headers, call {it}")
            if it == 1:
                from raftsql_amd import step as S

                live = want_o["type"] != S.OUT_SKIPPED
                full = S.expand_compact(gm.view(S.MSG_DT), go)
                assert np.array_equal(full[live].view(np.uint8), want_o[live].view(np.uint8)) and (go["type"][~live] == S.OUT_SKIPPED).all()
            else:
                _same(go, want_o, f"results, call {it}")
        _stepgen.assert_same_state(e, st)


# ---- raftq_propose_frames: appendEntry + bcastAppend on the device, into the encoder's input (round 6) ------------------------

def _propose_setup(rng, G, N, me, n_props, n_host_msgs, max_per_group=3):
    """a handle whose node leads every group but a few, random tails; -> (state dict, props, prop_ents, pool, host msgs / ents)"""
    from raftsql_amd.wire import PROP_DT, PROP_ENT_DT

    term = rng.integers(2, 9, G).astype(np.uint64)
    last = rng.integers(1, 1000, G).astype(np.uint64)
    last_term = np.minimum(term, rng.integers(1, 9, G).astype(np.uint64))
    committed = (last * rng.random(G)).astype(np.uint64)
    role = np.full(G, 2, np.uint8)
    groups = rng.choice(G, n_props, replace=False).astype(np.uint64)
    cnt = rng.integers(1, max_per_group + 1, n_props).astype(np.uint32)
    props = np.zeros(n_props, PROP_DT)
    props["group"], props["n_ents"] = groups, cnt
    props["ent_first"] = np.cumsum(cnt) - cnt
    ne = int(cnt.sum())
    pe = np.zeros(ne, PROP_ENT_DT)
    pe["data_len"] = rng.integers(0, 200, ne)
    pe["data_len"][rng.random(ne) < 0.1] = 0  # (an empty statement: Entry.Data omitted)
    # the host's own part of the turn: responses and a resend with entries, payloads in the same pool
    hm, he, hpool = _wiregen.random_msgs(rng, n_host_msgs, big_every=0, ent_frac=0.2) if n_host_msgs else (np.zeros(0, W.WIRE_MSG_DT), np.zeros(0, W.WIRE_ENT_DT), b"")
    hpool = _wiregen_u8(hpool)
    pe["data_off"] = len(hpool) + np.cumsum(pe["data_len"]) - pe["data_len"]
    pool = np.concatenate([hpool, rng.integers(0, 256, int(pe["data_len"].sum()), dtype=np.uint8)])
    st = dict(term=term, last=last, last_term=last_term, committed=committed, role=role)
    return st, props, pe, pool, hm, he


def _propose_expect(st, N, me, props, pe, hm, he):
    """what raft.go:211-215 -> appendEntry + bcastAppend make of props[]: the messages (host part, then one run per peer), the
    entry headers, and the state afterwards -- built here the way raftq_node.cpp's send_append builds them"""
    n_props = len(props)
    msgs = np.zeros(len(hm) + n_props * (N - 1), W.WIRE_MSG_DT)
    ents = np.zeros(len(he) + len(pe), W.WIRE_ENT_DT)
    msgs[: len(hm)] = hm
    ents[: len(he)] = he
    new_last = st["last"].copy()
    new_last_term = st["last_term"].copy()
    for i, p in enumerate(props):
        g, k, f = int(p["group"]), int(p["n_ents"]), int(p["ent_first"])
        for j in range(k):
            e = ents[len(he) + f + j]
            e["term"], e["index"] = st["term"][g], st["last"][g] + 1 + j
            e["data_len"], e["type"] = pe["data_len"][f + j], pe["type"][f + j]
            e["data_off"] = pe["data_off"][f + j] if pe["data_len"][f + j] else 0
            ents[len(he) + f + j] = e
        run = 0
        for to in range(N):
            if to == me:
                continue
            m = msgs[len(hm) + run * n_props + i]
            m["group"], m["term"], m["log_term"], m["index"], m["commit"] = g, st["term"][g], st["last_term"][g], st["last"][g], st["committed"][g]
            m["from"], m["to"], m["type"], m["ent_first"], m["n_ents"] = me, to, 3, len(he) + f, k
            msgs[len(hm) + run * n_props + i] = m
            run += 1
        new_last[g] += k
        new_last_term[g] = st["term"][g]
    return msgs, ents, new_last, new_last_term


def _propose_engine(G, N, me, st):
    from raftsql_amd.wire import WireEngine

    e = WireEngine(G, N, self_peer=me)
    match = np.tile(st["committed"], (N, 1))
    match[me] = st["last"]
    e.load_match(match, st["committed"])
    e.load_terms(st["term"], np.ones(G, np.uint64))
    e.load_roles(st["role"])
    e.load_node(st["term"], np.full(G, me + 1, np.uint32), np.full(G, me + 1, np.uint32), st["last"], st["last_term"])
    return e


@pytest.mark.parametrize("N,me,n_props,n_host", [(3, 0, 1, 0), (3, 2, 700, 300), (5, 1, 5000, 0), (7, 6, 2000, 4000), (2, 1, 33, 7)])
def test_propose_frames_is_append_entry_and_bcast_append(N, me, n_props, n_host):
    """raftq_propose_frames against the oracle's encoder over the messages bcastAppend would have built on the host: the stream
    and every frame offset byte for byte -- the caller's own messages first, then one run of MsgApps per peer --, and the
    device-resident state afterwards: lastIndex / lastTerm moved, the leader's own Match with them, nothing else touched."""
    from raftsql_amd.engine import pinned_copy, pinned_empty

    G = 8192
    rng = np.random.default_rng(7000 + n_props + N)
    st, props, pe, pool, hm, he = _propose_setup(rng, G, N, me, n_props, n_host)
    want_m, want_e, new_last, new_last_term = _propose_expect(st, N, me, props, pe, hm, he)
    want, want_off = W.wire_encode(want_m, want_e, pool)
    with _propose_engine(G, N, me, st) as e:
        before = e.read_node()
        match0 = e.read_match()
        out, off = pinned_empty(len(want) + 64, np.uint8), pinned_empty(len(want_m) + 1, np.uint64)
        out[:] = 0xEE
        got, goff, c = e.propose_frames(pinned_copy(props), pinned_copy(pe), pinned_copy(hm), pinned_copy(he), pinned_copy(pool), out, off)
        assert np.array_equal(goff, want_off)
        assert got.tobytes() == want.tobytes() and bytes(out[len(want):]) == b"\xee" * 64
        assert (c.n_msgs, c.n_ents, c.bytes) == (len(want_m), len(want_e), len(want))
        after = e.read_node()
        assert np.array_equal(after["last_index"], new_last) and np.array_equal(after["last_term"], new_last_term)
        for k in ("term", "vote", "lead", "first_idx", "role", "committed"):
            assert np.array_equal(after[k], before[k]), k
        match1 = e.read_match()
        want_match = match0.copy()
        want_match[me] = np.maximum(match0[me], new_last)
        assert np.array_equal(match1, want_match)
        # and Step goes on from the new tail: an acknowledgement of the last new entry by a quorum commits it
        from raftsql_amd import step as S_

        g = props["group"].astype(np.uint64)
        for p in [q for q in range(N) if q != me][: N // 2]:
            outs, _ = e.step_batch(S_.pack_msgs(g, S_.MSG_APP_RESP, term=st["term"][g.astype(np.int64)], frm=p, index=new_last[g.astype(np.int64)]))
        assert np.array_equal(e.read_committed()[g.astype(np.int64)], new_last[g.astype(np.int64)])


def test_propose_frames_refuses_and_applies_nothing():
    """a record that names a group this node does not lead, a group twice, no entries, entries outside prop_ents[], a payload
    outside the pool, a single-peer handle, pageable arrays: RAFTQ_EINVAL, the state untouched, and the next call is whole"""
    import ctypes as C

    from raftsql_amd import _lib
    from raftsql_amd.engine import pinned_copy, pinned_empty
    from raftsql_amd.wire import WireEngine

    G, N, me = 4096, 3, 1
    rng = np.random.default_rng(7777)
    st, props, pe, pool, hm, he = _propose_setup(rng, G, N, me, 600, 50)
    st["role"][int(props["group"][17])] = 0  # a follower among them
    st2 = st.copy()
    dup = props.copy()
    dup["group"][5] = dup["group"][400]
    none = props.copy()
    none["n_ents"][9] = 0
    outside = props.copy()
    outside["ent_first"][599] = len(pe)
    far = pe.copy()
    far["data_off"][3], far["data_len"][3] = len(pool), 8
    with _propose_engine(G, N, me, st) as e:
        before, match0 = e.read_node(), e.read_match()
        out, off = pinned_empty(1 << 20, np.uint8), pinned_empty(len(hm) + 600 * (N - 1) + 1, np.uint64)
        ph, pee, ppool = pinned_copy(hm), pinned_copy(he), pinned_copy(pool)
        for what, (pr, ents_, pl) in {"follower": (props, pe, pool), "twice": (dup, pe, pool), "empty": (none, pe, pool),
                                      "range": (outside, pe, pool), "payload": (props, far, pool)}.items():
            if what != "follower":
                e.load_roles(np.full(G, 2, np.uint8))
            with pytest.raises(Exception) as ei:
                e.propose_frames(pinned_copy(pr), pinned_copy(ents_), ph, pee, ppool, out, off)
            assert "propose_frames" in str(ei.value), (what, ei.value)
            after = e.read_node()
            for k in ("term", "last_index", "last_term", "committed"):
                assert np.array_equal(after[k], before[k]), (what, k)
            assert np.array_equal(e.read_match(), match0), what
        # pageable arrays: refused before anything is enqueued
        c = _lib.WireCounts()
        rc = e._lib.raftq_propose_frames(e._h, props.ctypes.data, len(props), pe.ctypes.data, len(pe), None, 0, None, 0, ppool.ctypes.data, len(ppool),
                                         out.ctypes.data, len(out), None, C.byref(c))
        assert rc == _lib.RAFTQ_EINVAL
        # ... and the whole call right behind the refusals
        want_m, want_e, new_last, _ = _propose_expect(st2 | {"role": np.full(G, 2, np.uint8)}, N, me, props, pe, hm, he)
        want, want_off = W.wire_encode(want_m, want_e, pool)
        got, goff, _ = e.propose_frames(pinned_copy(props), pinned_copy(pe), ph, pee, ppool, out, off)
        assert got.tobytes() == want.tobytes() and np.array_equal(goff[: len(want_off)], want_off)
        assert np.array_equal(e.read_node()["last_index"], new_last)
    with WireEngine(64, 1, self_peer=0) as e1:
        with pytest.raises(Exception):
            e1.propose_frames(pinned_copy(props[:1]), pinned_copy(pe), pinned_copy(hm[:0]), pinned_copy(he[:0]), pinned_copy(pool), out, None)


def test_propose_frames_at_bench_size():
    """32,768 groups x 3 peers, one statement each, behind 65,536 queued messages (the one-node leg's turn): the stream is the
    oracle's, three calls in a row (the control block carries over)"""
    from raftsql_amd.engine import pinned_copy, pinned_empty

    G, N, me = 32768, 3, 0
    rng = np.random.default_rng(7321)
    st, props, pe, pool, hm, he = _propose_setup(rng, G, N, me, G, 65536, max_per_group=1)
    with _propose_engine(G, N, me, st) as e:
        pp, ppe, ph, phe, ppool = pinned_copy(props), pinned_copy(pe), pinned_copy(hm), pinned_copy(he), pinned_copy(pool)
        for rep in range(3):
            want_m, want_e, new_last, new_lt = _propose_expect(st, N, me, props, pe, hm, he)
            want, want_off = W.wire_encode(want_m, want_e, pool)
            out, off = pinned_empty(len(want) + 64, np.uint8), pinned_empty(len(want_m) + 1, np.uint64)
            got, goff, _ = e.propose_frames(pp, ppe, ph, phe, ppool, out, off)
            assert np.array_equal(goff, want_off) and got.tobytes() == want.tobytes()
            st["last"], st["last_term"] = new_last, new_lt
