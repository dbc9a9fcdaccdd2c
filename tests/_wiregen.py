"""Seeded generators for the wire / WAL codec tests (CPU oracle tests and GPU parity tests share them).

Values are drawn to hit every varint length (1..10 bytes), empty and large payloads, every
message type, entries on MsgApp / MsgProp, and -- for the decoder -- protobuf streams that are
valid but not canonical (shuffled fields, unknown fields, over-long varints, repeated scalars)."""
import numpy as np

from oracle import pywire as W


def _varlen_u64(rng, n):
    """uint64 values whose bit length is uniform in 0..64 -> every varint size."""
    bits = rng.integers(0, 65, n)
    v = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    shift = (64 - bits).astype(np.uint64)
    out = np.where(bits == 0, np.uint64(0), v >> np.minimum(shift, np.uint64(63)))
    return np.where(bits == 0, np.uint64(0), out).astype(np.uint64)


def random_msgs(rng, n, n_peers=5, ent_frac=0.15, max_ents=5, max_payload=300, big_every=0):
    """-> (msgs, ents, pool bytes).  Entries only on MsgApp (3) / MsgProp (2)."""
    m = np.zeros(n, W.WIRE_MSG_DT)
    m["group"] = _varlen_u64(rng, n)
    m["term"] = _varlen_u64(rng, n)
    m["log_term"] = _varlen_u64(rng, n)
    m["index"] = _varlen_u64(rng, n)
    m["commit"] = _varlen_u64(rng, n)
    m["reject_hint"] = _varlen_u64(rng, n)
    small = rng.random(n) < 0.5  # half the batch looks like real traffic: small numbers
    for k in ("group", "term", "log_term", "index", "commit", "reject_hint"):
        m[k] = np.where(small, m[k] % np.uint64(1000), m[k])
    m["from"] = rng.integers(0, n_peers, n)
    m["to"] = rng.integers(0, n_peers, n)
    m["type"] = rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 9, 11], n)
    m["reject"] = rng.integers(0, 2, n)
    with_ents = ((m["type"] == 3) | (m["type"] == 2)) & (rng.random(n) < ent_frac / 0.2)
    cnt = np.where(with_ents, rng.integers(1, max_ents + 1, n), 0).astype(np.uint32)
    m["n_ents"] = cnt
    first = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.uint32)
    m["ent_first"] = np.where(cnt > 0, first, 0)
    ne = int(cnt.sum())
    e = np.zeros(ne, W.WIRE_ENT_DT)
    e["term"] = _varlen_u64(rng, ne) % np.uint64(1 << 40)
    e["index"] = _varlen_u64(rng, ne)
    e["type"] = rng.integers(0, 2, ne)
    ln = rng.integers(0, max_payload + 1, ne).astype(np.uint32)
    ln[rng.random(ne) < 0.1] = 0  # becomeLeader's empty entry
    if big_every and ne:
        ln[::big_every] = rng.integers(4000, 20000, len(ln[::big_every]))
    e["data_len"] = ln
    # payloads scattered through the pool with gaps and arbitrary alignment, not in entry order
    order = rng.permutation(ne)
    gaps = rng.integers(0, 7, ne)
    off = np.zeros(ne, np.uint64)
    pos = 0
    for j in order:
        pos += int(gaps[j])
        off[j] = pos
        pos += int(ln[j])
    e["data_off"] = np.where(ln > 0, off, 0)
    pool = rng.integers(0, 256, max(pos, 1), dtype=np.uint8)
    return m, e, pool


def random_wal(rng, n, max_payload=300, big_every=0, head=True):
    """-> (recs, pool).  A segment head (crc, metadata, snapshot) then entries / states mixed,
    with an occasional mid-stream CRC record (a segment cut, wal.cut -> saveCrc)."""
    r = np.zeros(n, W.WAL_REC_DT)
    kind = rng.choice([W.WAL_ENTRY, W.WAL_STATE, W.WAL_SNAPSHOT, W.WAL_METADATA], n, p=[0.6, 0.3, 0.05, 0.05])
    if head and n >= 3:
        kind[0], kind[1], kind[2] = W.WAL_CRC, W.WAL_METADATA, W.WAL_SNAPSHOT
    r["kind"] = kind
    r["group"] = np.where(rng.random(n) < 0.5, rng.integers(0, 1000, n), _varlen_u64(rng, n))
    r["term"] = np.where(rng.random(n) < 0.5, rng.integers(0, 100, n), _varlen_u64(rng, n))
    r["index"] = np.where(rng.random(n) < 0.5, rng.integers(0, 100000, n), _varlen_u64(rng, n))
    r["vote"] = np.where(kind == W.WAL_STATE, rng.integers(0, 10, n), 0)
    r["entry_type"] = np.where(kind == W.WAL_ENTRY, rng.integers(0, 2, n), 0)
    has_payload = (kind == W.WAL_ENTRY) | (kind == W.WAL_METADATA)
    ln = np.where(has_payload, rng.integers(0, max_payload + 1, n), 0).astype(np.uint32)
    ln[rng.random(n) < 0.1] = 0
    if big_every:
        idx = np.nonzero(has_payload)[0][::big_every]
        ln[idx] = rng.integers(4000, 20000, len(idx))
    r["data_len"] = ln
    order = rng.permutation(n)
    gaps = rng.integers(0, 7, n)
    off = np.zeros(n, np.uint64)
    pos = 0
    for j in order:
        pos += int(gaps[j])
        off[j] = pos
        pos += int(ln[j])
    r["data_off"] = np.where(ln > 0, off, 0)
    # only the fields each kind carries (so decode(encode(x)) == x field for field)
    for k in ("group",):
        r[k] = np.where((kind == W.WAL_ENTRY) | (kind == W.WAL_STATE), r[k], 0)
    r["term"] = np.where((kind == W.WAL_CRC) | (kind == W.WAL_METADATA), 0, r["term"])
    r["index"] = np.where((kind == W.WAL_CRC) | (kind == W.WAL_METADATA), 0, r["index"])
    pool = rng.integers(0, 256, max(pos, 1), dtype=np.uint8)
    return r, pool


def _varint(v: int, pad: int = 0) -> bytes:
    """protobuf varint; pad > 0 appends that many redundant continuation bytes (valid, non-canonical)."""
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    for _ in range(min(pad, 10 - len(out))):  # a varint is at most 10 bytes
        out[-1] |= 0x80
        out.append(0)
    return bytes(out)


def noncanonical_message(rng, m, ents, pool) -> bytes:
    """The same Message as message_bytes(m, ...) but as an arbitrary VALID protobuf stream: fields in
    random order (entries keep their relative order), scalars sometimes written twice (last wins),
    unknown fields of every skippable wire type, over-long varints, absent zero fields."""
    parts = []

    def scalar(fn, v, allow_absent=True):
        if v == 0 and allow_absent and rng.random() < 0.5:
            return
        if rng.random() < 0.2:  # an earlier, overridden occurrence
            parts.append((rng.random(), _varint(fn << 3) + _varint(int(rng.integers(0, 1 << 30)))))
            base = 1.0
        else:
            base = 0.0
        parts.append((base + rng.random(), _varint(fn << 3, int(rng.integers(0, 2))) + _varint(int(v), int(rng.integers(0, 3)))))

    scalar(1, int(m["type"]))
    scalar(2, int(m["to"]) + 1, False)
    scalar(3, int(m["from"]) + 1, False)
    scalar(4, int(m["term"]))
    scalar(5, int(m["log_term"]))
    scalar(6, int(m["index"]))
    scalar(8, int(m["commit"]))
    scalar(10, int(m["reject"]))
    scalar(11, int(m["reject_hint"]))
    scalar(12, int(m["group"]), False)
    if rng.random() < 0.5:
        parts.append((rng.random() * 2, bytes([0x4A, 0x00]) if rng.random() < 0.5 else bytes.fromhex("4a0812060a0010001800")))
    for _ in range(int(rng.integers(0, 4))):  # unknown fields
        fn = int(rng.integers(13, 3000))
        wt = int(rng.choice([0, 1, 2, 5]))
        body = {0: lambda: _varint(int(rng.integers(0, 1 << 62))), 1: lambda: bytes(rng.integers(0, 256, 8, dtype=np.uint8)),
                2: lambda: (lambda b: _varint(len(b)) + b)(bytes(rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8))),
                5: lambda: bytes(rng.integers(0, 256, 4, dtype=np.uint8))}[wt]()
        parts.append((rng.random() * 2, _varint(fn << 3 | wt) + body))
    keys = sorted(rng.random(int(m["n_ents"])) * 2)
    for k in range(int(m["n_ents"])):
        e = ents[int(m["ent_first"]) + k]
        d = bytes(pool[int(e["data_off"]): int(e["data_off"]) + int(e["data_len"])])
        fields = [_varint(1 << 3) + _varint(int(e["type"])), _varint(2 << 3) + _varint(int(e["term"]), int(rng.integers(0, 2))),
                  _varint(3 << 3) + _varint(int(e["index"]))]
        if d or rng.random() < 0.3:
            fields.append(bytes([0x22]) + _varint(len(d)) + d)
        if rng.random() < 0.3:
            fields.append(_varint(99 << 3 | 0) + _varint(7))
        body = b"".join(fields[i] for i in rng.permutation(len(fields)))
        parts.append((keys[k], bytes([0x3A]) + _varint(len(body)) + body))
    parts.sort(key=lambda t: t[0])
    return b"".join(p for _, p in parts)
