"""CPU: the one host-only entry point of include/raftq_wire.h -- raftq_wire_scan_frames, the serial
length-word walk both decoders and raftq_node_deliver / _replay_wal start with -- against the oracle's
walk on encoded streams cut anywhere, and the record dtypes of the host mirror against the oracle's
independent statement of the same C structs."""
import numpy as np
import pytest

from oracle import pywire as W
from tests import _wiregen


@pytest.fixture(scope="module")
def wire():
    from raftsql_amd import build, wire as w

    build.build_lib()
    return w


@pytest.mark.parametrize("big_endian", [True, False])
def test_scan_frames_equals_oracle(wire, big_endian):
    rng = np.random.default_rng(17 + big_endian)
    if big_endian:
        m, e, pool = _wiregen.random_msgs(rng, 300, ent_frac=0.4)
        s, off = W.wire_encode(m, e, pool)
    else:
        r, pool = _wiregen.random_wal(rng, 300)
        s, off, _ = W.wal_encode(r, pool, 0)
    cuts = [0, 1, 7, 8, 9, len(s) - 1, len(s)] + [int(x) for x in rng.integers(0, len(s), 40)] + \
           [int(off[k]) + d for k in (1, 5, 100) for d in (-1, 0, 1, 7, 8)]
    for cut in cuts:
        want, wused = W.scan_frames(s[:cut], big_endian=big_endian)
        got, used = wire.scan_frames(s[:cut], big_endian=big_endian)
        assert used == wused and np.array_equal(got, want), cut
        assert used == off[np.searchsorted(off, cut, side="right") - 1]
    got, used = wire.scan_frames(s, big_endian=big_endian, cap=17)  # at most cap frames
    assert len(got) == 18 and used == off[17]
    got, used = wire.scan_frames(s, big_endian=not big_endian)  # the wrong byte order: the first length is absurd
    assert used == 0 and len(got) == 1
    noise = rng.integers(0, 256, 4096, dtype=np.uint8)
    assert np.array_equal(wire.scan_frames(noise, big_endian)[0], W.scan_frames(noise, big_endian)[0])


def test_mirror_dtypes_match_the_oracle_binding(wire):
    for a, b in ((wire.WIRE_MSG_DT, W.WIRE_MSG_DT), (wire.WIRE_ENT_DT, W.WIRE_ENT_DT), (wire.WAL_REC_DT, W.WAL_REC_DT)):
        assert a.itemsize == b.itemsize and a.names == b.names
        assert [a.fields[n][1] for n in a.names] == [b.fields[n][1] for n in b.names]
    assert (wire.F_MALFORMED, wire.F_SNAPSHOT, wire.F_GROUP) == (W.F_MALFORMED, W.F_SNAPSHOT, W.F_GROUP)
    assert (wire.WAL_METADATA, wire.WAL_ENTRY, wire.WAL_STATE, wire.WAL_CRC, wire.WAL_SNAPSHOT) == \
           (W.WAL_METADATA, W.WAL_ENTRY, W.WAL_STATE, W.WAL_CRC, W.WAL_SNAPSHOT)
