"""CPU: the per-frame Unmarshal code the decode kernels run per lane (raftsql_amd/csrc/raftq_wire_parse.hpp), compiled for
the HOST from the same source, against the codec oracle -- on the corpora of the GPU parity tests (canonical,
non-canonical, hand-made malformed frames, mutated frames, pure noise, the committed fixtures).  What this buys: a change
to the parser (round 3 turned Message.Unmarshal into one flat, switch-free loop) is proven equal to the oracle here, on
every run of the CPU suite, before it is ever sent to a GPU; tests/test_wire_gpu.py then checks the kernels around it."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import pywire as W
from tests import _wiregen
from tests import test_wire_gpu as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "wire_parse_host.cpp")
HDR = os.path.join(ROOT, "raftsql_amd", "csrc", "raftq_wire_parse.hpp")
LIB = os.path.join(ROOT, "tests", "c", "libwire_parse_host.so")


def _build() -> C.CDLL:
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
                               "-I" + os.path.dirname(HDR), "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.host_wire_decode.restype = C.c_uint64
    lib.host_wire_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.host_crc32c.restype = C.c_uint32
    lib.host_crc32c.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
    lib.host_wal_parse.restype = None
    lib.host_wal_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class HostDecoder:
    """The decode half of raftsql_amd.wire.WireEngine's interface over the host build of the parser."""

    def __init__(self):
        self.lib = _build()

    def wire_decode(self, stream, frame_off, want_ents=True):
        # the kernels may read up to the buffer's end with 8-byte loads, never beyond: a buffer of exactly nbytes, with a
        # poisoned guard behind it that a stray read would pick up as different bytes on the second run
        s = np.frombuffer(bytes(stream), np.uint8) if not isinstance(stream, np.ndarray) else stream
        off = np.ascontiguousarray(frame_off, np.uint64)
        n = len(off) - 1
        outs = []
        for guard in (0x00, 0xFF):
            buf = np.full(len(s) + 64, guard, np.uint8)
            buf[: len(s)] = s
            msgs = np.zeros(n, W.WIRE_MSG_DT)
            ne = C.c_uint64(0)
            cap = len(s) // 2 + 1
            ents = np.zeros(cap, W.WIRE_ENT_DT)
            bad = self.lib.host_wire_decode(buf.ctypes.data, len(s), off.ctypes.data, n, msgs.ctypes.data, ents.ctypes.data, cap,
                                            C.byref(ne))
            outs.append((msgs, ents[: ne.value].copy(), int(bad)))
        assert outs[0][0].tobytes() == outs[1][0].tobytes() and outs[0][1].tobytes() == outs[1][1].tobytes(), \
            "the parse depends on bytes behind the end of the buffer"
        msgs, ents, bad = outs[0]
        return (msgs, ents if want_ents else ents[:0], bad)

    def wal_parse(self, data, frame_off):
        b = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data
        off = np.ascontiguousarray(frame_off, np.uint64)
        n = len(off) - 1
        buf = np.zeros(len(b) + 64, np.uint8)
        buf[: len(b)] = b
        recs = np.zeros(n, W.WAL_REC_DT)
        so, sl = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        self.lib.host_wal_parse(buf.ctypes.data, len(b), off.ctypes.data, n, recs.ctypes.data, so.ctypes.data, sl.ctypes.data)
        return recs, so, sl


@pytest.fixture(scope="module")
def host():
    return HostDecoder()


@pytest.mark.parametrize("seed,n,big", [(111, 700, 5), (112, 4000, 0), (113, 3000, 2)])
def test_canonical_streams(host, seed, n, big):
    G.test_decode_parity_canonical(host, seed, n, big)


@pytest.mark.parametrize("seed", [121, 122, 123, 124])
def test_noncanonical_streams(host, seed):
    G.test_decode_parity_noncanonical(host, seed)


def test_hand_made_malformed_frames(host):
    G.test_decode_malformed_frames(host)


@pytest.mark.parametrize("seed", [131, 132, 133, 134, 135, 136, 137, 138])
def test_mutated_frames_and_noise(host, seed):
    G.test_decode_fuzz(host, seed)


def test_traffic_without_entries_and_short_frames(host):
    """What Step-from-frames sees (no entries, small numbers: 30-40 byte frames) plus frames of every tiny length: the
    fast one-load-per-field form is taken everywhere but in the last seven bytes of the buffer."""
    rng = np.random.default_rng(7)
    m, e, pool = _wiregen.random_msgs(rng, 5000, ent_frac=0.0)
    for k in ("group", "term", "log_term", "index", "commit", "reject_hint"):
        m[k] %= np.uint64(1 << 20)
    s, off = W.wire_encode(m, e, pool)
    wm, we, wbad = W.wire_decode(s, off)
    gm, ge, gbad = host.wire_decode(s, off)
    assert gbad == wbad == 0 and len(ge) == 0
    G._same(gm, wm, "msgs")
    for cut in range(1, 60):  # a buffer that ends inside / right behind a frame
        o2 = np.array([0, min(cut, int(off[1])), cut], np.uint64)
        wm, we, wbad = W.wire_decode(s[:cut], o2)
        gm, ge, gbad = host.wire_decode(s[:cut], o2)
        assert gbad == wbad
        G._same(gm, wm, "cut %d" % cut)


def test_fields_at_every_varint_length_and_every_field_order(host):
    """Every field of the message at every value length 1..10 bytes, in random order with unknown fields between them:
    the boundary between the one-load form (values up to seven bytes) and the byte loop."""
    rng = np.random.default_rng(17)
    bodies = []
    for _ in range(3000):
        parts = []
        for fn in rng.permutation(np.array([1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 15, 16, 200])):
            nb = int(rng.integers(1, 11))
            v = int(rng.integers(0, 1 << 62)) >> max(0, 62 - 7 * nb + int(rng.integers(0, 7)))
            key = _wiregen._varint(int(fn) << 3)
            parts.append(key + _wiregen._varint(v, pad=int(rng.integers(0, 3)) if rng.random() < 0.2 else 0))
        if rng.random() < 0.5:
            ent = b"".join(_wiregen._varint(f << 3) + _wiregen._varint(int(rng.integers(0, 1 << 50))) for f in rng.permutation([1, 2, 3]))
            ent += b"\x22" + _wiregen._varint(5) + b"hello"
            parts.insert(int(rng.integers(0, len(parts))), b"\x3a" + _wiregen._varint(len(ent)) + ent)
        if rng.random() < 0.5:
            parts.insert(int(rng.integers(0, len(parts))), bytes.fromhex("4a0812060a0010001800"))
        bodies.append(b"".join(parts))
    stream = b"".join(G._be(b) for b in bodies)
    off = np.concatenate([[0], np.cumsum([len(b) + 8 for b in bodies])]).astype(np.uint64)
    wm, we, wbad = W.wire_decode(stream, off)
    gm, ge, gbad = host.wire_decode(stream, off)
    assert gbad == wbad
    G._same(gm, wm, "msgs")
    G._same(ge, we, "ents")


def test_committed_fixtures(host):
    """tests/golden/wire_golden.json: the decoded bytes frozen there (no oracle involved)."""
    g = json.load(open(G.GOLD))
    w = g["wire"]
    mm, ee, bad = host.wire_decode(bytes.fromhex(w["stream"]), np.array(w["frame_off"], np.uint64))
    assert bad == 0 and mm.tobytes().hex() == w["decoded_msgs"] and ee.tobytes().hex() == w["decoded_ents"]
    nc = g["wire_noncanonical"]
    mm, ee, bad = host.wire_decode(bytes.fromhex(nc["stream"]), np.array(nc["frame_off"], np.uint64))
    assert bad == nc["n_malformed"] and mm.tobytes().hex() == nc["decoded_msgs"] and ee.tobytes().hex() == nc["decoded_ents"]
    a = g["wal"]
    rr, _, _ = host.wal_parse(bytes.fromhex(a["bytes"]), np.array(a["frame_off"], np.uint64))
    assert rr.tobytes().hex() == a["decoded_recs"]


@pytest.mark.parametrize("seed", [151, 152, 161, 162])
def test_wal_records_parse_like_the_oracle(host, seed):
    """walpb.Record + the Data unmarshal per type, on valid segments and on mutated ones (the CRC verdict is the kernels'
    business: the flag bit it sets is masked out here)."""
    rng = np.random.default_rng(seed)
    r, pool = _wiregen.random_wal(rng, 1500, max_payload=120, big_every=13)
    out, off, _ = W.wal_encode(r, pool, 0)
    for mutate in (False, True):
        s = out.copy()
        if mutate:
            n_mut = len(s) // 60
            s[rng.integers(0, len(s), n_mut)] = rng.integers(0, 256, n_mut, dtype=np.uint8)
            for i in rng.choice(len(r), len(r) * 9 // 10, replace=False):
                a, b = int(off[i]), int(off[i + 1])
                s[a:a + 8] = np.frombuffer((b - a - 8).to_bytes(8, "little"), np.uint8)
        wr, _, _ = W.wal_decode(s, off, 0)
        gr, so, sl = host.wal_parse(s, off)
        wr = wr.copy()
        wr["flags"] &= np.uint8(~W.WAL_F_BADCRC & 0xFF)
        G._same(gr, wr, "wal recs (mutated=%s)" % mutate)


def test_crc32c_eight_bytes_per_step_equals_the_oracle(host):
    """The slicing-by-8 update the kernels run (tables t[k][i], eight reads per eight bytes) against the oracle's bitwise
    CRC-32C and RFC 3720's vectors: every length 0..70, long buffers, chained seeds."""
    rng = np.random.default_rng(5)
    g = json.load(open(G.GOLD))
    for c in g["crc32c"]:
        d = np.frombuffer(bytes.fromhex(c["data"]), np.uint8)
        buf = np.concatenate([d, np.zeros(8, np.uint8)])
        assert host.lib.host_crc32c(c["seed"], buf.ctypes.data, len(d)) == c["crc"], c
    for n in list(range(0, 71)) + [255, 256, 257, 4096, 100003]:
        d = rng.integers(0, 256, n + 8, dtype=np.uint8)
        seed = int(rng.integers(0, 1 << 32))
        assert host.lib.host_crc32c(seed, d.ctypes.data, n) == W.crc32c(d[:n].tobytes(), seed), n
