#!/usr/bin/env python3
"""Generates tests/golden/wire_golden.json -- the frozen wire / WAL codec fixtures.

Every expected byte string in the file comes from the google.protobuf runtime on the recalled
raftpb / walpb schema (tests/pbschema.py), NOT from the oracle; the decoded record arrays are the
oracle's output after it was checked field by field against the runtime's ParseFromString.  CRC
values are RFC 3720 appendix B.4's plus chained values checked by two implementations.

    python tests/golden/make_wire_golden.py          (needs google.protobuf; run from the repo root)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pywire as W  # noqa: E402
from tests import _wiregen, pbschema as P  # noqa: E402
from tests.test_wire_oracle import RFC3720, _pb_wal  # noqa: E402


def main():
    rng = np.random.default_rng(0xC0FFEE)
    g = {"note": __doc__.strip().splitlines()[0]}
    g["crc32c"] = [{"data": d.hex(), "seed": 0, "crc": c} for d, c in RFC3720]
    for _ in range(6):
        d = bytes(rng.integers(0, 256, int(rng.integers(1, 80)), dtype=np.uint8))
        seed = int(rng.integers(0, 1 << 32))
        c = W.crc32c(d, seed)
        assert c == W.crc32c(d, seed, table=True)
        g["crc32c"].append({"data": d.hex(), "seed": seed, "crc": c})

    m, e, pool = _wiregen.random_msgs(rng, 64, max_payload=40)
    pb = bytes(pool)
    stream = b"".join(P.frame_be(P.message_bytes(m[i], e, pb)) for i in range(len(m)))
    s, off = W.wire_encode(m, e, pool)
    assert bytes(s) == stream, "oracle encoder disagrees with the protobuf runtime"
    mm, ee, bad = W.wire_decode(stream, off)
    assert bad == 0
    g["wire"] = {"msgs": m.tobytes().hex(), "ents": e.tobytes().hex(), "pool": pb.hex(), "stream": stream.hex(),
                 "frame_off": [int(x) for x in off], "decoded_msgs": mm.tobytes().hex(), "decoded_ents": ee.tobytes().hex()}

    # valid-but-not-canonical streams, with malformed frames in between
    bodies = [_wiregen.noncanonical_message(rng, m[i], e, pool) for i in range(24)]
    junk = [bytes.fromhex(x) for x in ("2201aa", "080610021801208080", "0001", "3a021280")]
    frames = []
    for i, b in enumerate(bodies):
        frames.append(P.frame_be(b))
        if i % 6 == 5:
            frames.append(P.frame_be(junk[i // 6]))
    stream = b"".join(frames)
    off = np.concatenate([[0], np.cumsum([len(f) for f in frames])]).astype(np.uint64)
    mm, ee, bad = W.wire_decode(stream, off)
    Msg = P.classes()["Message"]
    k = 0
    for i, f in enumerate(frames):
        if (i + 1) % 7 == 0:
            assert mm[i]["flags"] == W.F_MALFORMED
            continue
        pm = Msg()
        pm.ParseFromString(f[8:])
        assert (mm[i]["type"], mm[i]["term"], mm[i]["index"], mm[i]["group"], mm[i]["n_ents"]) == \
            (pm.type, pm.term, pm.index, pm.group, len(pm.entries)), i
        k += 1
    assert bad == 4
    g["wire_noncanonical"] = {"stream": stream.hex(), "frame_off": [int(x) for x in off], "n_malformed": bad,
                              "decoded_msgs": mm.tobytes().hex(), "decoded_ents": ee.tobytes().hex()}

    r, pool = _wiregen.random_wal(rng, 48, max_payload=40)
    prev = 0
    want = _pb_wal(r, bytes(pool), prev)
    out, off, last = W.wal_encode(r, pool, prev)
    assert bytes(out) == want, "oracle WAL encoder disagrees with the protobuf runtime"
    rr, nv, lc = W.wal_decode(out, off, prev)
    assert nv == len(r) and lc == last
    g["wal"] = {"recs": r.tobytes().hex(), "pool": bytes(pool).hex(), "prev_crc": prev, "bytes": want.hex(),
                "frame_off": [int(x) for x in off], "last_crc": last, "decoded_recs": rr.tobytes().hex()}
    path = os.path.join(ROOT, "tests", "golden", "wire_golden.json")
    json.dump(g, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
