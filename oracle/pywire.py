"""ctypes binding of the wire / WAL codec oracle (oracle/raftq_wire_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Record layouts are written out here independently of raftsql_amd/wire.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import pyoracle

WIRE_MSG_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("log_term", "<u8"), ("index", "<u8"), ("commit", "<u8"),
                        ("reject_hint", "<u8"), ("from", "<u4"), ("type", "u1"), ("reject", "u1"), ("to", "u1"),
                        ("flags", "u1"), ("ent_first", "<u4"), ("n_ents", "<u4")])
WIRE_ENT_DT = np.dtype([("term", "<u8"), ("index", "<u8"), ("data_off", "<u8"), ("data_len", "<u4"), ("type", "<u4")])
WAL_REC_DT = np.dtype([("group", "<u8"), ("term", "<u8"), ("index", "<u8"), ("data_off", "<u8"), ("data_len", "<u4"),
                       ("vote", "<u4"), ("crc", "<u4"), ("kind", "u1"), ("entry_type", "u1"), ("flags", "u1"),
                       ("_pad", "u1")])
assert WIRE_MSG_DT.itemsize == 64 and WIRE_ENT_DT.itemsize == 32 and WAL_REC_DT.itemsize == 48

F_MALFORMED, F_SNAPSHOT, F_GROUP = 1, 2, 4
WAL_METADATA, WAL_ENTRY, WAL_STATE, WAL_CRC, WAL_SNAPSHOT = 1, 2, 3, 4, 5
WAL_F_MALFORMED, WAL_F_BADCRC, WAL_F_GROUP = 1, 2, 4

_ready = False


def _lib() -> C.CDLL:
    global _ready
    L = pyoracle.lib()
    if not _ready:
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        for name in ("rq_crc32c_update", "rq_crc32c_update_table"):
            f = getattr(L, name)
            f.restype, f.argtypes = u32, [u32, vp, C.c_size_t]
        L.rq_crc32c_mulmod.restype, L.rq_crc32c_mulmod.argtypes = u32, [u32, u32]
        L.rq_crc32c_xpow8.restype, L.rq_crc32c_xpow8.argtypes = u32, [u64]
        L.rq_crc32c_combine.restype, L.rq_crc32c_combine.argtypes = u32, [u32, u32, u64]
        L.rq_wire_encode.restype, L.rq_wire_encode.argtypes = u64, [vp, u64, vp, vp, vp, u64, vp]
        L.rq_wire_decode.restype, L.rq_wire_decode.argtypes = C.c_int, [vp, u64, vp, u64, vp, vp, u64, vp, vp]
        L.rq_wire_scan_frames.restype, L.rq_wire_scan_frames.argtypes = C.c_int, [vp, u64, C.c_int, vp, u64, vp, vp]
        L.rq_wal_encode.restype, L.rq_wal_encode.argtypes = u64, [vp, u64, vp, u32, vp, u64, vp, vp]
        L.rq_wal_decode.restype, L.rq_wal_decode.argtypes = C.c_int, [vp, u64, vp, u64, u32, vp, vp, vp]
        L.rq_wire_set_fast_crc.restype, L.rq_wire_set_fast_crc.argtypes = None, [C.c_int]
        _ready = True
    return L


def set_fast_crc(on: bool) -> None:
    """table-driven instead of bitwise CRC inside wal_encode / wal_decode (bench cpu baseline only)"""
    _lib().rq_wire_set_fast_crc(int(on))


def _bytes_arr(b) -> np.ndarray:
    a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, np.uint8)
    return a if len(a) else np.zeros(1, np.uint8)  # a valid pointer even when empty


def crc32c(data, crc: int = 0, table: bool = False) -> int:
    a = _bytes_arr(data)
    n = len(data)
    f = _lib().rq_crc32c_update_table if table else _lib().rq_crc32c_update
    return int(f(crc, a.ctypes.data, n))


def mulmod(a: int, b: int) -> int:
    return int(_lib().rq_crc32c_mulmod(a, b))


def xpow8(n: int) -> int:
    return int(_lib().rq_crc32c_xpow8(n))


def combine(crc_a: int, crc_b: int, len_b: int) -> int:
    return int(_lib().rq_crc32c_combine(crc_a, crc_b, len_b))


def wire_encode(msgs: np.ndarray, ents: np.ndarray | None = None, pool=b""):
    """-> (stream bytes as uint8 array, frame_off uint64[n+1])"""
    m = np.ascontiguousarray(msgs, dtype=WIRE_MSG_DT)
    e = np.ascontiguousarray(ents if ents is not None else np.zeros(0, WIRE_ENT_DT), dtype=WIRE_ENT_DT)
    e_ptr = e.ctypes.data if len(e) else None
    p = _bytes_arr(pool)
    off = np.zeros(len(m) + 1, np.uint64)
    need = _lib().rq_wire_encode(m.ctypes.data, len(m), e_ptr, p.ctypes.data, None, 0, off.ctypes.data)
    out = np.zeros(max(int(need), 1), np.uint8)
    got = _lib().rq_wire_encode(m.ctypes.data, len(m), e_ptr, p.ctypes.data, out.ctypes.data, need, off.ctypes.data)
    assert got == need
    return out[:need], off


def wire_decode(stream, frame_off):
    """-> (msgs, ents, n_malformed)"""
    s = _bytes_arr(stream)
    off = np.ascontiguousarray(frame_off, np.uint64)
    n = len(off) - 1
    msgs = np.zeros(n, WIRE_MSG_DT)
    ne, nb = C.c_uint64(0), C.c_uint64(0)
    L = _lib()
    L.rq_wire_decode(s.ctypes.data, len(stream), off.ctypes.data, n, msgs.ctypes.data, None, 0, C.byref(ne), C.byref(nb))
    ents = np.zeros(ne.value, WIRE_ENT_DT)
    if ne.value:
        L.rq_wire_decode(s.ctypes.data, len(stream), off.ctypes.data, n, msgs.ctypes.data, ents.ctypes.data, ne.value,
                         C.byref(ne), C.byref(nb))
    return msgs, ents, int(nb.value)


def scan_frames(buf, big_endian: bool, cap: int | None = None):
    """-> (off uint64[n+1], consumed)"""
    b = _bytes_arr(buf)
    cap = len(buf) // 8 + 1 if cap is None else cap
    off = np.zeros(cap + 1, np.uint64)
    n, used = C.c_uint64(0), C.c_uint64(0)
    _lib().rq_wire_scan_frames(b.ctypes.data, len(buf), int(big_endian), off.ctypes.data, cap, C.byref(n), C.byref(used))
    return off[: n.value + 1].copy(), int(used.value)


def wal_encode(recs: np.ndarray, pool=b"", prev_crc: int = 0):
    """-> (bytes uint8 array, frame_off, last_crc)"""
    r = np.ascontiguousarray(recs, dtype=WAL_REC_DT)
    p = _bytes_arr(pool)
    off = np.zeros(len(r) + 1, np.uint64)
    last = C.c_uint32(0)
    L = _lib()
    need = L.rq_wal_encode(r.ctypes.data, len(r), p.ctypes.data, prev_crc, None, 0, off.ctypes.data, C.byref(last))
    out = np.zeros(max(int(need), 1), np.uint8)
    got = L.rq_wal_encode(r.ctypes.data, len(r), p.ctypes.data, prev_crc, out.ctypes.data, need, off.ctypes.data,
                          C.byref(last))
    assert got == need
    return out[:need], off, int(last.value)


def wal_decode(data, frame_off, prev_crc: int = 0):
    """-> (recs, n_valid, last_crc)"""
    b = _bytes_arr(data)
    off = np.ascontiguousarray(frame_off, np.uint64)
    n = len(off) - 1
    recs = np.zeros(n, WAL_REC_DT)
    nv, last = C.c_uint64(0), C.c_uint32(0)
    _lib().rq_wal_decode(b.ctypes.data, len(data), off.ctypes.data, n, prev_crc, recs.ctypes.data, C.byref(nv),
                         C.byref(last))
    return recs, int(nv.value), int(last.value)
