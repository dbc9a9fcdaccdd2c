"""ctypes binding of libraftq.so (the C-ABI in include/raftq.h).

The library is the product: if it is missing or does not load, importing the
engine fails loudly.  There is no CPU fallback path in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAFTQ_LIB: another build of the SAME library out of raftsql_amd/build.py (the sanitizer builds libraftq_asan.so /
# libraftq_tsan.so, an A/B build with other -D defines).  It must sit in this package's directory: the package has no
# way to be pointed at anything else -- test doubles are the tests' business (tests/conftest.py), not the product's.
LIB_PATH = os.environ.get("RAFTQ_LIB") or os.path.join(_HERE, "libraftq.so")

RAFTQ_OK = 0
RAFTQ_EINVAL, RAFTQ_ENOMEM, RAFTQ_EHIP, RAFTQ_ESTATE, RAFTQ_ENODEV = -1, -2, -3, -4, -5

SWEEP_COMMIT = 0x01
SWEEP_GATED = 0x02
SWEEP_VOTES = 0x04
SWEEP_NO_ADOPT = 0x08
SWEEP_LDS = 0x10
SWEEP_CHANGED = 0x20
SWEEP_STREAM = 0x40
SWEEP_CACHED = 0x80
CYCLE_TRUSTED = 0x100
TICK_BEAT_BITMAP = 1  # raftq_tick_collect_lists: the MsgBeat groups as a group-order bitmap instead of a list
CYCLE_SEGMENTED = 0x200  # raftq_cycle_packed: the advance list may be left in segments (raftq_last_advance_segments)
SET_GRID, SET_PERSISTENT = 0, 1

MAX_PEERS = 9


class Counts(C.Structure):
    _fields_ = [("n_changed", C.c_uint64), ("n_won", C.c_uint64), ("n_lost", C.c_uint64)]


class TickCounts(C.Structure):
    _fields_ = [("n_hup", C.c_uint64), ("n_beat", C.c_uint64)]


class Delta(C.Structure):
    _fields_ = [("group", C.c_uint64), ("match", C.c_uint64), ("peer", C.c_uint32), ("_pad", C.c_uint32)]


class VoteDelta(C.Structure):
    _fields_ = [("group", C.c_uint64), ("peer", C.c_uint32), ("vote", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class Advance(C.Structure):
    _fields_ = [("group", C.c_uint64), ("old_commit", C.c_uint64), ("new_commit", C.c_uint64)]


# every symbol include/raftq.h declares: (name, restype, argtypes)
_H = C.c_void_p
_SIGS = [
    ("raftq_abi_version", C.c_int, []),
    ("raftq_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("raftq_quorum", C.c_uint32, [C.c_uint32]),
    ("raftq_create", C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(_H)]),
    ("raftq_destroy", None, [_H]),
    ("raftq_groups", C.c_uint64, [_H]),
    ("raftq_peers", C.c_uint32, [_H]),
    ("raftq_last_error", C.c_char_p, [_H]),
    ("raftq_set_stream", C.c_int, [_H, C.c_void_p]),
    ("raftq_get_stream", C.c_void_p, [_H]),
    ("raftq_load_match", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("raftq_load_terms", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("raftq_load_votes", C.c_int, [_H, C.c_void_p]),
    ("raftq_apply_deltas", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("raftq_apply_vote_deltas", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("raftq_apply_term_deltas", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("raftq_step_async", C.c_int, [_H, C.c_uint]),
    ("raftq_wait", C.c_int, [_H, C.POINTER(Counts)]),
    ("raftq_commit_advance", C.c_int, [_H, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]),
    ("raftq_vote_tally", C.c_int, [_H, C.c_void_p, C.POINTER(Counts)]),
    ("raftq_read_committed", C.c_int, [_H, C.c_void_p]),
    ("raftq_read_outcome", C.c_int, [_H, C.c_void_p]),
    ("raftq_read_match", C.c_int, [_H, C.c_void_p]),
    ("raftq_read_votes", C.c_int, [_H, C.c_void_p]),
    ("raftq_collect_changed", C.c_int, [_H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_set_timers", C.c_int, [_H, C.c_uint32, C.c_uint32, C.c_uint64]),
    ("raftq_load_roles", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("raftq_tick", C.c_int, [_H, C.POINTER(TickCounts)]),
    ("raftq_read_tick", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("raftq_collect_hups", C.c_int, [_H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_collect_beats", C.c_int, [_H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_tick_collect", C.c_int, [_H, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("raftq_tick_collect_lists", C.c_int, [_H, C.c_uint, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("raftq_last_tick_lists", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_campaign", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_uint32]),
    ("raftq_cycle", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint, C.c_void_p, C.c_uint64,
                              C.POINTER(C.c_uint64), C.POINTER(Counts)]),
    ("raftq_stage", C.c_int, [_H, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("raftq_last_advances", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_cycle_packed", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint, C.c_void_p, C.c_uint64,
                                     C.POINTER(C.c_uint64), C.POINTER(Counts)]),
    ("raftq_stage_packed", C.c_int, [_H, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("raftq_last_advances_packed", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_last_advance_segments", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("raftq_set_create", C.c_int, [C.POINTER(_H), C.c_uint32, C.POINTER(_H)]),
    ("raftq_set_destroy", None, [_H]),
    ("raftq_set_size", C.c_uint32, [_H]),
    ("raftq_set_last_error", C.c_char_p, [_H]),
    ("raftq_set_get_stream", C.c_void_p, [_H]),
    ("raftq_set_mode", C.c_int, [_H, C.c_int, C.c_uint32]),
    ("raftq_set_sweep_async", C.c_int, [_H, C.c_uint]),
    ("raftq_set_wait", C.c_int, [_H, C.c_void_p, C.POINTER(Counts)]),
    ("raftq_set_tick", C.c_int, [_H]),
    ("raftq_set_timer_begin", C.c_int, [_H]),
    ("raftq_set_timer_end", C.c_int, [_H, C.POINTER(C.c_float)]),
    ("raftq_sweep_many_async", C.c_int, [C.POINTER(_H), C.c_uint32, C.c_uint]),
    ("raftq_clone_state", C.c_int, [_H, _H]),
    ("raftq_timer_begin", C.c_int, [_H]),
    ("raftq_timer_end", C.c_int, [_H, C.POINTER(C.c_float)]),
    ("raftq_host_alloc", C.c_int, [C.POINTER(C.c_void_p), C.c_uint64]),
    ("raftq_host_free", None, [C.c_void_p]),
]
EXPORTS = [s[0] for s in _SIGS]


class StepCounts(C.Structure):
    _fields_ = [("n_msgs", C.c_uint64), ("n_groups_touched", C.c_uint64)]


# every symbol include/raftq_step.h declares
_STEP_SIGS = [
    ("raftq_set_self", C.c_int, [_H, C.c_uint32]),
    ("raftq_load_node", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("raftq_read_node", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("raftq_step_batch", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(StepCounts)]),
    ("raftq_apply_log_deltas", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("raftq_apply_log_deltas_nowait", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("raftq_step_submit", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("raftq_step_collect", C.c_int, [_H, C.c_void_p, C.POINTER(StepCounts)]),
    ("raftq_step_stage", C.c_int, [_H, C.c_uint64, C.POINTER(C.c_void_p)]),
    ("raftq_step_stage_packed", C.c_int, [_H, C.c_uint64, C.POINTER(C.c_void_p)]),
    ("raftq_step_submit_packed", C.c_int, [_H, C.c_void_p, C.c_uint64]),
    ("raftq_step_results", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_step_set_compact", C.c_int, [_H, C.c_int]),
    ("raftq_step_set_msg_flags", C.c_int, [_H, C.c_int]),
    ("raftq_step_results_c", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_step_results_s", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
]
STEP_EXPORTS = [s[0] for s in _STEP_SIGS]



class WireCounts(C.Structure):
    _fields_ = [("n_msgs", C.c_uint64), ("n_ents", C.c_uint64), ("n_malformed", C.c_uint64), ("bytes", C.c_uint64)]


class WalCounts(C.Structure):
    _fields_ = [("n_recs", C.c_uint64), ("n_valid", C.c_uint64), ("bytes", C.c_uint64), ("last_crc", C.c_uint32),
                ("_pad", C.c_uint32)]


# every symbol include/raftq_wire.h declares
_WIRE_SIGS = [
    ("raftq_wire_encode", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                    C.c_uint64, C.c_void_p, C.POINTER(WireCounts)]),
    ("raftq_wire_decode", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.POINTER(WireCounts)]),
    ("raftq_wire_scan_frames", C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    ("raftq_step_submit_wire", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    ("raftq_step_frames", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.c_void_p]),
    ("raftq_propose_frames", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(WireCounts)]),
    ("raftq_step_stage_wire", C.c_int, [_H, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("raftq_step_wire_msgs", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_step_wire_entries", C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    ("raftq_wal_encode", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                   C.c_void_p, C.POINTER(WalCounts)]),
    ("raftq_wal_encode_begin", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                                         C.c_void_p]),
    ("raftq_wal_encode_end", C.c_int, [_H, C.POINTER(WalCounts)]),
    ("raftq_wal_decode", C.c_int, [_H, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                   C.POINTER(WalCounts)]),
]
WIRE_EXPORTS = [s[0] for s in _WIRE_SIGS]

_lib = None


class RaftqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"raftq error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """dlopen libraftq.so.  torch (when present) is imported first so that the
    process ends up with ONE HIP runtime: torch bundles its own libamdhip64
    under the same soname, and whichever is mapped first serves both."""
    global _lib
    if _lib is not None:
        return _lib
    env_lib = os.environ.get("RAFTQ_LIB")
    if env_lib and LIB_PATH == env_lib and os.path.dirname(os.path.abspath(env_lib)) != _HERE:
        raise ImportError(f"RAFTQ_LIB={env_lib}: only builds of libraftq inside {_HERE} are loaded (raftsql_amd/build.py puts "
                          "them there).  raftsql_amd has no CPU path and no test double.")
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -m raftsql_amd.build` (needs hipcc). "
            "raftsql_amd has no CPU fallback for the quorum sweep.")
    try:
        import torch  # noqa: F401  (HIP runtime unification, see docstring)
    except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in _SIGS + _STEP_SIGS + _WIRE_SIGS:
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
