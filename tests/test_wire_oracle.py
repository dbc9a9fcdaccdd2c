"""CPU: the wire / WAL codec oracle (oracle/raftq_wire_oracle.c) pinned against independent
references -- the google.protobuf runtime on the recalled raftpb / walpb schema (tests/pbschema.py),
RFC 3720's CRC-32C vectors, hand-derived byte strings -- and against the committed fixtures
(tests/golden/wire_golden.json, generator tests/golden/make_wire_golden.py)."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import pywire as W
from tests import _wiregen, pbschema as P

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wire_golden.json")


# ---- CRC-32C ------------------------------------------------------------------------------------

RFC3720 = [  # RFC 3720 appendix B.4 "CRC Examples" (+ the classic check string)
    (bytes(32), 0x8A9136AA),
    (b"\xff" * 32, 0x62A8AB43),
    (bytes(range(32)), 0x46DD794E),
    (bytes(range(31, -1, -1)), 0x113FDB5C),
    (b"123456789", 0xE3069283),
    (b"", 0x00000000),
]


@pytest.mark.parametrize("data,want", RFC3720)
def test_crc32c_published_vectors(data, want):
    assert W.crc32c(data) == want
    assert W.crc32c(data, table=True) == want


def test_crc32c_iscsi_read_pdu():
    # RFC 3720 B.4, "An iSCSI - SCSI Read (10) Command PDU"
    pdu = bytes.fromhex("01c00000" "00000000" "00000000" "00000000" "14000000" "00000400" "00000014" "00000018"
                        "28000000" "00000000" "02000000" "00000000")
    assert W.crc32c(pdu) == 0xD9963A56


def test_crc32c_bitwise_equals_table_and_chains():
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8))
        b = bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8))
        seed = int(rng.integers(0, 1 << 32))
        assert W.crc32c(a, seed) == W.crc32c(a, seed, table=True)
        # hash/crc32 Update continues a CRC: Update(Update(s, a), b) == Update(s, a || b)
        assert W.crc32c(b, W.crc32c(a, seed)) == W.crc32c(a + b, seed)
        # the identity the GPU scan uses: crc(A||B) = crc(A) * x^(8|B|) + crc(B)  (also for a seeded A)
        assert W.combine(W.crc32c(a, seed), W.crc32c(b), len(b)) == W.crc32c(a + b, seed)


def test_crc32c_combine_algebra():
    """(crc, x^(8 len)) pairs under  (a, ma) . (b, mb) = (a*mb + b, ma*mb)  form a monoid -- what makes
    the chain a parallel scan; and a 'reset' element (v, 0) absorbs everything before it."""
    rng = np.random.default_rng(2)
    one = 0x80000000  # x^0
    assert W.xpow8(0) == one and W.xpow8(1) == 0x00800000
    for _ in range(100):
        a, b, c = (int(x) for x in rng.integers(0, 1 << 32, 3))
        assert W.mulmod(a, one) == a and W.mulmod(one, a) == a
        assert W.mulmod(a, b) == W.mulmod(b, a)
        assert W.mulmod(W.mulmod(a, b), c) == W.mulmod(a, W.mulmod(b, c))
        assert W.mulmod(a ^ b, c) == W.mulmod(a, c) ^ W.mulmod(b, c)
        n1, n2 = int(rng.integers(0, 5000)), int(rng.integers(0, 1 << 40))
        assert W.mulmod(W.xpow8(n1), W.xpow8(n2)) == W.xpow8(n1 + n2)

    def op(x, y):
        return (W.mulmod(x[0], y[1]) ^ y[0], W.mulmod(x[1], y[1]))

    for _ in range(50):
        x, y, z = [(int(rng.integers(0, 1 << 32)), W.xpow8(int(rng.integers(0, 300)))) for _ in range(3)]
        assert op(op(x, y), z) == op(x, op(y, z))
        r = (int(rng.integers(0, 1 << 32)), 0)
        assert op(x, r) == r and op(op(x, r), y) == op(r, y)


def test_crc32c_is_not_zlib_crc32():
    assert W.crc32c(b"123456789") != zlib.crc32(b"123456789")  # IEEE polynomial: cbf43926


# ---- raftpb.Message ------------------------------------------------------------------------------

def test_message_hand_vectors():
    """Bytes worked out by hand from the protobuf encoding rules."""
    m = np.zeros(1, W.WIRE_MSG_DT)
    m["type"], m["to"], m["from"], m["term"] = 6, 1, 0, 5  # MsgVoteResp{To: 2, From: 1, Term: 5}
    s, off = W.wire_encode(m)
    body = bytes.fromhex("0806" "1002" "1801" "2005" "2800" "3000" "4000" "4a0812060a0010001800" "5000" "5800" "6000")
    assert bytes(s) == (30).to_bytes(8, "big") + body and list(off) == [0, 38]
    # MsgApp{To:3 From:1 Term:300 LogTerm:299 Index:2^32 Commit:127, reject hint 128, group 2^63}, one entry "hi"
    m["type"], m["to"], m["term"], m["log_term"], m["index"], m["commit"] = 3, 2, 300, 299, 1 << 32, 127
    m["reject"], m["reject_hint"], m["group"], m["n_ents"] = 1, 128, 1 << 63, 1
    e = np.zeros(1, W.WIRE_ENT_DT)
    e["term"], e["index"], e["data_len"], e["type"] = 300, (1 << 32) + 1, 2, 1
    s, off = W.wire_encode(m, e, b"hi")
    body = bytes.fromhex("0803" "1003" "1801" "20ac02" "28ab02" "308080808010"
                         "3a0f" "0801" "10ac02" "188180808010" "22026869"
                         "407f" "4a0812060a0010001800" "5001" "588001" "6080808080808080808001")
    assert bytes(s) == len(body).to_bytes(8, "big") + body
    mm, ee, bad = W.wire_decode(s, off)
    assert bad == 0 and mm[0]["flags"] == W.F_GROUP
    for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject", "to", "n_ents"):
        assert mm[0][k] == m[0][k], k
    assert (ee[0]["term"], ee[0]["index"], ee[0]["data_len"], ee[0]["type"]) == (300, (1 << 32) + 1, 2, 1)
    assert bytes(s[int(ee[0]["data_off"]):][:2]) == b"hi"


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_encode_equals_protobuf_runtime(seed):
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, 400, big_every=7)
    s, off = W.wire_encode(m, e, pool)
    pb = bytes(pool)
    for i in range(len(m)):
        want = P.frame_be(P.message_bytes(m[i], e, pb))
        assert bytes(s[int(off[i]):int(off[i + 1])]) == want, i


def _check_decoded(mm, ee, stream, m, e, pool):
    for k in ("group", "term", "log_term", "index", "commit", "reject_hint", "from", "type", "reject", "to", "n_ents"):
        assert np.array_equal(mm[k], m[k]), k
    assert np.array_equal(mm["ent_first"], m["ent_first"])
    for k in ("term", "index", "data_len", "type"):
        assert np.array_equal(ee[k], e[k]), k
    for j in range(len(e)):
        a, n = int(ee[j]["data_off"]), int(e[j]["data_len"])
        assert bytes(stream[a:a + n]) == bytes(pool[int(e[j]["data_off"]):int(e[j]["data_off"]) + n])


@pytest.mark.parametrize("seed", [21, 22])
def test_decode_inverts_encode(seed):
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, 500, big_every=5)
    s, off = W.wire_encode(m, e, pool)
    mm, ee, bad = W.wire_decode(s, off)
    assert bad == 0 and np.all(mm["flags"] == W.F_GROUP)
    _check_decoded(mm, ee, s, m, e, pool)
    # frames found from the length words alone
    off2, used = W.scan_frames(s, big_endian=True)
    assert used == len(s) and np.array_equal(off2, off)


@pytest.mark.parametrize("seed", [31, 32])
def test_decode_equals_protobuf_runtime_on_noncanonical_streams(seed):
    """Any valid protobuf serialisation of the message decodes the same: the runtime's parse of
    the stream and the oracle's must agree field for field."""
    rng = np.random.default_rng(seed)
    m, e, pool = _wiregen.random_msgs(rng, 300)
    bodies = [_wiregen.noncanonical_message(rng, m[i], e, pool) for i in range(len(m))]
    stream = b"".join(P.frame_be(b) for b in bodies)
    off = np.concatenate([[0], np.cumsum([len(b) + 8 for b in bodies])]).astype(np.uint64)
    mm, ee, bad = W.wire_decode(stream, off)
    assert bad == 0
    Msg = P.classes()["Message"]
    j = 0
    for i, b in enumerate(bodies):
        pm = Msg()
        pm.ParseFromString(b)
        assert mm[i]["type"] == pm.type and mm[i]["to"] == pm.to - 1 and mm[i]["from"] == getattr(pm, "from") - 1
        assert (mm[i]["term"], mm[i]["log_term"], mm[i]["index"], mm[i]["commit"]) == (pm.term, pm.logTerm, pm.index, pm.commit)
        assert (mm[i]["reject"], mm[i]["reject_hint"], mm[i]["group"]) == (int(pm.reject), pm.rejectHint, pm.group)
        assert mm[i]["n_ents"] == len(pm.entries)
        for pe in pm.entries:
            assert (ee[j]["type"], ee[j]["term"], ee[j]["index"], ee[j]["data_len"]) == (pe.Type, pe.Term, pe.Index, len(pe.Data))
            a = int(ee[j]["data_off"])
            assert stream[a:a + len(pe.Data)] == pe.Data
            j += 1
    assert j == len(ee)
    _check_decoded(mm, ee, np.frombuffer(stream, np.uint8), m, e, pool)


def test_decode_malformed_frames_are_flagged_not_fatal():
    good = P.frame_be(bytes.fromhex("0806100218012005"))
    cases = {
        "length word disagrees": (9).to_bytes(8, "big") + bytes.fromhex("0806100218012005"),
        "truncated varint": P.frame_be(bytes.fromhex("080610021801208080")),
        "11-byte varint": P.frame_be(bytes.fromhex("20" + "80" * 10 + "01")),
        "entry length overruns": P.frame_be(bytes.fromhex("08033a0508001005")),
        "wrong wire type on term": P.frame_be(bytes.fromhex("2201aa")),
        "tag 0": P.frame_be(bytes.fromhex("0001")),
        "group wire type (3)": P.frame_be(bytes.fromhex("6b")),
        "bad entry inside": P.frame_be(bytes.fromhex("3a021280")),
        "bad snapshot inside": P.frame_be(bytes.fromhex("4a021201")),
        "shorter than a length word": b"\x00\x00\x00",
    }
    stream = good
    off = [0, len(good)]
    for b in cases.values():
        stream += b + good
        off += [off[-1] + len(b), off[-1] + len(b) + len(good)]
    mm, ee, bad = W.wire_decode(stream, np.array(off, np.uint64))
    flags = mm["flags"]
    assert bad == len(cases) and len(ee) == 0
    assert list(flags[0::2]) == [0] * (len(cases) + 1), "the good frames around them still decode"
    assert list(flags[1::2]) == [W.F_MALFORMED] * len(cases)
    assert np.all(mm["term"][0::2] == 5) and np.all(mm["term"][1::2] == 0)
    # 10-byte varints are fine (the high bits fall off, as in Go), negative int32 type clamps to 255
    ok = P.frame_be(bytes.fromhex("08" + "ff" * 9 + "01" + "20" + "ff" * 9 + "7f"))
    mm, _, bad = W.wire_decode(ok, np.array([0, len(ok)], np.uint64))
    assert bad == 0 and mm[0]["type"] == 255 and mm[0]["term"] == (1 << 64) - 1
    assert mm[0]["to"] == 0xFF and mm[0]["from"] == 0xFFFFFFFF, "absent IDs decode to the invalid slot"


def test_decode_snapshot_flag():
    Msg = P.classes()["Message"]
    pm = Msg()
    pm.type = 7
    pm.snapshot.metadata.index = 9
    pm.snapshot.metadata.conf_state.nodes.extend([1, 2, 3])
    pm.snapshot.data = b"x"
    a = P.frame_be(pm.SerializeToString())
    pm2 = Msg()
    pm2.snapshot.metadata.conf_state.SetInParent()
    pm2.snapshot.metadata.index = 0
    b = P.frame_be(pm2.SerializeToString())
    mm, _, bad = W.wire_decode(a + b, np.array([0, len(a), len(a) + len(b)], np.uint64))
    assert bad == 0 and mm[0]["flags"] & W.F_SNAPSHOT and not mm[1]["flags"] & W.F_SNAPSHOT and mm[0]["type"] == 7


def test_scan_frames_stops_at_a_torn_tail():
    rng = np.random.default_rng(5)
    m, e, pool = _wiregen.random_msgs(rng, 50)
    s, off = W.wire_encode(m, e, pool)
    for cut in (len(s) - 1, int(off[20]) + 3, int(off[20]) + 8, 0, 7):
        o, used = W.scan_frames(s[:cut], big_endian=True)
        k = int(np.searchsorted(off, cut, side="right")) - 1
        assert used == off[k] and np.array_equal(o, off[:k + 1])
    o, used = W.scan_frames(s, big_endian=True, cap=10)
    assert len(o) == 11 and used == off[10]


# ---- WAL ------------------------------------------------------------------------------------------

def _pb_wal(recs, pool: bytes, prev_crc=0) -> bytes:
    """wal.encoder.encode restated with the protobuf runtime + the oracle's bitwise CRC."""
    C = P.classes()
    out, crc = b"", prev_crc
    for r in recs:
        k = int(r["kind"])
        if k == W.WAL_ENTRY:
            d = pool[int(r["data_off"]): int(r["data_off"]) + int(r["data_len"])]
            data = P.entry_pb(r["entry_type"], r["term"], r["index"], d, group=r["group"]).SerializeToString()
        elif k == W.WAL_STATE:
            h = C["HardState"]()
            h.term, h.vote, h.commit, h.group = int(r["term"]), int(r["vote"]), int(r["index"]), int(r["group"])
            data = h.SerializeToString()
        elif k == W.WAL_SNAPSHOT:
            sn = C["WalSnapshot"]()
            sn.index, sn.term = int(r["index"]), int(r["term"])
            data = sn.SerializeToString()
        elif k == W.WAL_METADATA:
            data = pool[int(r["data_off"]): int(r["data_off"]) + int(r["data_len"])]
        else:
            data = b""
        crc = W.crc32c(data, crc)
        rec = C["Record"]()
        rec.type, rec.crc = k, crc
        if k != W.WAL_CRC and not (k == W.WAL_METADATA and not data):
            rec.data = data
        out += P.frame_le(rec.SerializeToString())
    return out


def test_wal_hand_vector():
    """wal.Create's head + one Save, bytes by hand: crc record, empty metadata, snapshot{0,0},
    entry{Term 1, Index 1, "a", group 0}, state{1, 2, 1, group 0}."""
    r = np.zeros(5, W.WAL_REC_DT)
    r["kind"] = [W.WAL_CRC, W.WAL_METADATA, W.WAL_SNAPSHOT, W.WAL_ENTRY, W.WAL_STATE]
    r[3]["term"], r[3]["index"], r[3]["data_len"] = 1, 1, 1
    r[4]["term"], r[4]["vote"], r[4]["index"] = 1, 2, 1
    out, off, last = W.wal_encode(r, b"a", 0)
    snap = bytes.fromhex("08001000")
    ent = bytes.fromhex("080010011801" "220161" "2800")
    st = bytes.fromhex("0801100218012000")
    c1 = W.crc32c(snap)
    c2 = W.crc32c(ent, c1)
    c3 = W.crc32c(st, c2)
    vi = _wiregen._varint
    want = b"".join(P.frame_le(x) for x in (
        bytes.fromhex("08041000"),
        bytes.fromhex("08011000"),
        bytes.fromhex("0805") + b"\x10" + vi(c1) + b"\x1a\x04" + snap,
        bytes.fromhex("0802") + b"\x10" + vi(c2) + b"\x1a\x0b" + ent,
        bytes.fromhex("0803") + b"\x10" + vi(c3) + b"\x1a\x08" + st))
    assert bytes(out) == want and last == c3 and off[-1] == len(want)


@pytest.mark.parametrize("seed,prev", [(41, 0), (42, 0xDEADBEEF)])
def test_wal_encode_equals_protobuf_runtime(seed, prev):
    rng = np.random.default_rng(seed)
    r, pool = _wiregen.random_wal(rng, 300, big_every=9, head=(prev == 0))
    out, off, last = W.wal_encode(r, pool, prev)
    want = _pb_wal(r, bytes(pool), prev)
    assert bytes(out) == want
    Rec = P.classes()["Record"]
    pr = Rec()
    pr.ParseFromString(bytes(out[int(off[-2]) + 8:]))
    assert pr.crc == last


@pytest.mark.parametrize("seed", [51, 52])
def test_wal_decode_inverts_encode_and_validates(seed):
    rng = np.random.default_rng(seed)
    r, pool = _wiregen.random_wal(rng, 400, big_every=11)
    r["kind"][200] = W.WAL_CRC  # a segment cut in mid-stream
    for k in ("group", "term", "index", "vote", "data_len", "data_off", "entry_type"):
        r[k][200] = 0
    out, off, last = W.wal_encode(r, pool, 0)
    o2, used = W.scan_frames(out, big_endian=False)
    assert used == len(out) and np.array_equal(o2, off)
    rr, n_valid, last2 = W.wal_decode(out, off, 0)
    assert n_valid == len(r) and last2 == last
    for k in ("group", "term", "index", "vote", "data_len", "kind", "entry_type"):
        assert np.array_equal(rr[k], r[k]), k
    has_g = (r["kind"] == W.WAL_ENTRY) | (r["kind"] == W.WAL_STATE)
    assert np.array_equal(rr["flags"], np.where(has_g, W.WAL_F_GROUP, 0))
    for j in np.nonzero(r["data_len"])[0]:
        a, n = int(rr[j]["data_off"]), int(r[j]["data_len"])
        assert bytes(out[a:a + n]) == bytes(pool[int(r[j]["data_off"]):int(r[j]["data_off"]) + n])
    # every single-byte corruption inside a record's Data is caught at that record (CRC-32 detects any burst <= 32 bits)
    for j in rng.choice(np.nonzero(r["data_len"] > 0)[0], 25, replace=False):
        bad = out.copy()
        bad[int(rr[j]["data_off"]) + int(rng.integers(0, r[j]["data_len"]))] ^= 1 << int(rng.integers(0, 8))
        r3, nv, lc = W.wal_decode(bad, off, 0)
        assert nv == j and r3[j]["flags"] & W.WAL_F_BADCRC
        # ... and, the chain being cumulative, every later record up to the next CRC record mismatches too
        nxt = 200 if j < 200 else len(r)
        assert np.all(r3["flags"][j:nxt] & W.WAL_F_BADCRC) and not np.any(r3["flags"][:j] & W.WAL_F_BADCRC)
        if j < 200:
            assert r3[200]["flags"] & W.WAL_F_BADCRC and not np.any(r3["flags"][201:] & W.WAL_F_BADCRC)
    # the wrong seed fails at the first record that carries data; a zero seed skips the head check (ReadAll: `crc != 0 &&`)
    _, nv, _ = W.wal_decode(out, off, 12345)
    assert nv == 0
    # a torn tail: the frames that are whole still validate
    cut = int(off[300]) + 5
    o3, used = W.scan_frames(out[:cut], big_endian=False)
    assert used == off[300]
    _, nv, lc = W.wal_decode(out[:cut], o3, 0)
    assert nv == 300
    # ... and appending continues the chain from there: encode(rest, prev = lc) reproduces the original bytes
    rest, _, last3 = W.wal_encode(r[300:], pool, lc)
    assert bytes(rest) == bytes(out[int(off[300]):]) and last3 == last


def test_wal_decode_malformed_record():
    r, pool = _wiregen.random_wal(np.random.default_rng(6), 20)
    out, off, _ = W.wal_encode(r, pool, 0)
    bad = out.copy()
    bad[int(off[10]) + 8] = 0x0B  # field 1 with wire type 3 (group start): does not parse
    rr, nv, _ = W.wal_decode(bad, off, 0)
    assert nv == 10 and rr[10]["flags"] == W.WAL_F_MALFORMED
    bad = out.copy()
    bad[int(off[10]) + 9] = 9  # Record.type 9: "unexpected block type"
    rr, nv, _ = W.wal_decode(bad, off, 0)
    assert nv == 10 and rr[10]["flags"] == W.WAL_F_MALFORMED


# ---- committed fixtures ---------------------------------------------------------------------------

def test_golden_fixtures():
    g = json.load(open(GOLD))
    for c in g["crc32c"]:
        assert W.crc32c(bytes.fromhex(c["data"]), c["seed"]) == c["crc"]
    w = g["wire"]
    m = np.frombuffer(bytes.fromhex(w["msgs"]), W.WIRE_MSG_DT)
    e = np.frombuffer(bytes.fromhex(w["ents"]), W.WIRE_ENT_DT)
    pool = bytes.fromhex(w["pool"])
    s, off = W.wire_encode(m, e, pool)
    assert bytes(s).hex() == w["stream"] and list(map(int, off)) == w["frame_off"]
    mm, ee, bad = W.wire_decode(s, off)
    assert bad == 0 and mm.tobytes().hex() == w["decoded_msgs"] and ee.tobytes().hex() == w["decoded_ents"]
    nc = g["wire_noncanonical"]
    mm, ee, bad = W.wire_decode(bytes.fromhex(nc["stream"]), np.array(nc["frame_off"], np.uint64))
    assert bad == nc["n_malformed"] and mm.tobytes().hex() == nc["decoded_msgs"] and ee.tobytes().hex() == nc["decoded_ents"]
    a = g["wal"]
    r = np.frombuffer(bytes.fromhex(a["recs"]), W.WAL_REC_DT)
    out, off, last = W.wal_encode(r, bytes.fromhex(a["pool"]), a["prev_crc"])
    assert bytes(out).hex() == a["bytes"] and last == a["last_crc"] and list(map(int, off)) == a["frame_off"]
    rr, nv, lc = W.wal_decode(out, off, a["prev_crc"])
    assert nv == len(r) and lc == last and rr.tobytes().hex() == a["decoded_recs"]
