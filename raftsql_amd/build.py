"""Build the in-tree native pieces: libraftq.so (HIP, gfx950) and the tuner.

Explicit hipcc invocations so the built .so sits next to the package and
travels to the GPU box with the source snapshot (a JIT cache would not).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raftsql_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libraftq.so")
TUNER = os.path.join(PKG, "raftq_tune")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libraftq.so (no CPU fallback exists)")


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


N_UNITS = 5


def lib_sources() -> list[str]:
    """[translation units..., headers...] -- the first N_UNITS are compiled."""
    return [
        os.path.join(CSRC, "raftq_capi.hip"),
        os.path.join(CSRC, "raftq_step.hip"),
        os.path.join(CSRC, "raftq_wire.hip"),
        os.path.join(CSRC, "raftq_pipe.cpp"),
        os.path.join(CSRC, "raftq_node.cpp"),
        os.path.join(CSRC, "raftq_kernels.hpp"),
        os.path.join(CSRC, "raftq_step_kernels.hpp"),
        os.path.join(CSRC, "raftq_internal.hpp"),
        os.path.join(CSRC, "raftq_wire_kernels.hpp"),
        os.path.join(ROOT, "include", "raftq.h"),
        os.path.join(ROOT, "include", "raftq_step.h"),
        os.path.join(ROOT, "include", "raftq_pipe.h"),
        os.path.join(ROOT, "include", "raftq_node.h"),
        os.path.join(ROOT, "include", "raftq_wire.h"),
    ]


def build_lib(force: bool = False, defines: dict | None = None, verbose: bool = False) -> str:
    srcs = lib_sources()
    if not force and not _stale(LIB, srcs):
        return LIB
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    for k, v in (defines or {}).items():
        cmd.append(f"-D{k}={v}")
    cmd += ["-o", LIB] + srcs[:N_UNITS]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_tuner(force: bool = False) -> str | None:
    src = os.path.join(CSRC, "raftq_tune.hip")
    if not os.path.exists(src):
        return None
    if not force and not _stale(TUNER, [src, os.path.join(CSRC, "raftq_kernels.hpp")]):
        return TUNER
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-o", TUNER, src]
    subprocess.check_call(cmd)
    return TUNER


def build_tuner2(force: bool = False) -> str | None:
    """Second tuner: layout / instruction-count A/B (profiles/r01/tune2_*.jsonl)."""
    src = os.path.join(CSRC, "raftq_tune2.hip")
    out = TUNER + "2"
    if not os.path.exists(src):
        return None
    if not force and not _stale(out, [src, os.path.join(CSRC, "raftq_kernels.hpp")]):
        return out
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-o", out, src]
    subprocess.check_call(cmd)
    return out


def build_all(force: bool = False) -> None:
    build_lib(force=force)
    build_tuner(force=force)
    build_tuner2(force=force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(LIB)
