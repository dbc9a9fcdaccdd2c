"""Build the in-tree native pieces: libraftq.so (HIP, gfx950) and the standalone measurement tools.

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
        os.path.join(CSRC, "raftq_wire_parse.hpp"),
        os.path.join(CSRC, "raftq_propose_kernels.hpp"),
        os.path.join(ROOT, "include", "raftq.h"),
        os.path.join(ROOT, "include", "raftq_step.h"),
        os.path.join(ROOT, "include", "raftq_pipe.h"),
        os.path.join(ROOT, "include", "raftq_node.h"),
        os.path.join(ROOT, "include", "raftq_wire.h"),
    ]


def _flags() -> list[str]:
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def build_lib(force: bool = False, defines: dict | None = None, verbose: bool = False, log: list | None = None,
              variant: str | None = None) -> str:
    """libraftq.so from its five translation units: compiled concurrently (one hipcc per unit), then linked.
    `log` (a list) receives the exact command lines.  `variant` + `defines`: an A/B or measurement build beside the
    product library (raftsql_amd/libraftq_<variant>.so, loaded with RAFTQ_LIB=<that path>)."""
    srcs = lib_sources()
    LIB = globals()["LIB"] if not variant else os.path.join(PKG, f"libraftq_{variant}.so")
    if not force and not _stale(LIB, srcs):
        return LIB
    objdir = os.path.join(PKG, "build", variant) if variant else os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    defs = [f"-D{k}={v}" for k, v in (defines or {}).items()]
    procs, objs = [], []
    for src in srcs[:N_UNITS]:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [_hipcc()] + _flags() + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if log is not None:
            log.append(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    if log is not None:
        log.append(" ".join(link))
    subprocess.check_call(link)
    os.replace(LIB + ".tmp", LIB)
    return LIB


SANITIZERS = {"asan": "address", "ubsan": "undefined", "tsan": "thread"}


def build_sanitized(kind: str, verbose: bool = False) -> str:
    """libraftq_<kind>.so: the HOST side of every translation unit under AddressSanitizer + UBSan ("asan") or
    ThreadSanitizer ("tsan") -- raftq_pipe.cpp / raftq_node.cpp (mutexes, condition variables, a background thread) and the
    host halves of the .hip units; the device code is compiled as always (-fno-gpu-sanitize).  Loaded instead of the
    product library with RAFTQ_LIB=<path> and the matching clang runtime preloaded (tools/gpurun_trip.sh sanitize)."""
    san = SANITIZERS[kind]
    out = os.path.join(PKG, f"libraftq_{kind}.so")
    objdir = os.path.join(PKG, "build", kind)
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O1", "-g", "-std=c++17", "-fPIC", "-fno-omit-frame-pointer", f"-fsanitize={san}",
             "-fno-gpu-sanitize", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    procs, objs = [], []
    for src in lib_sources()[:N_UNITS]:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-fsanitize={san}", "-shared-libsan", "-o", out] + objs)
    return out


def sanitizer_runtime(kind: str) -> str:
    """The clang runtime to LD_PRELOAD for a sanitized build (it must come first in the process)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
    name = {"asan": "libclang_rt.asan-x86_64.so", "tsan": "libclang_rt.tsan-x86_64.so",
            "ubsan": "libclang_rt.ubsan_standalone-x86_64.so"}[kind]
    return subprocess.run([clang, "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()


TUNE_DIR = os.path.join(ROOT, "tools", "tune")


def build_tool(src: str, out: str | None = None, force: bool = False, deps: list[str] | None = None) -> str | None:
    """A standalone measurement / probe binary (tools/tune/*.hip, tools/probe/*.hip): NOT part of libraftq.so.  Built
    in-tree beside its source so that it travels to the GPU box, which needs no compiler for it."""
    if not os.path.exists(src):
        return None
    out = out or os.path.splitext(src)[0]
    if not force and not _stale(out, [src] + (deps or [os.path.join(CSRC, "raftq_kernels.hpp")])):
        return out
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-o", out, src]
    subprocess.check_call(cmd)
    return out


def build_c_tool(src: str, force: bool = False) -> str | None:
    """A plain-C99 caller of the C-ABI (tools/tune/turn_latency.c: the batching turn as cgo would make it), gcc, linked against
    the in-tree libraftq.so through an $ORIGIN-relative rpath so that it runs wherever the tree is copied."""
    if not os.path.exists(src):
        return None
    out = os.path.splitext(src)[0]
    if not force and not _stale(out, [src, os.path.join(ROOT, "include", "raftq.h")]):
        return out
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), src, "-L" + PKG, "-lraftq",
                           "-Wl,-rpath,$ORIGIN/../../raftsql_amd", "-o", out])
    return out


def tool_sources() -> list[str]:
    """Every standalone HIP tool of the tree: the sweep tuners (tools/tune/: raftq_tune = round 1's policy / tile A/B,
    raftq_tune2 = layout / instruction count, raftq_tune3 = launch shapes + the PMC calibration copy, single_launch_ab =
    round 4's one-launch-at-a-time A/B) and the link / BAR probes (tools/probe/)."""
    import glob

    return sorted(glob.glob(os.path.join(TUNE_DIR, "*.hip")) + glob.glob(os.path.join(ROOT, "tools", "probe", "*.hip")))


SORT_CHECK = os.path.join(PKG, "raftq_sort_check")


def build_sort_check(force: bool = False) -> str:
    """tests/c/sort_check.hip: the radix sort of raftq_sort_kernels.hpp against std::stable_sort (tests/test_sort_gpu.py).
    Built in-tree with everything else so that the GPU box does not need a compiler for it."""
    src = os.path.join(ROOT, "tests", "c", "sort_check.hip")
    deps = [src, os.path.join(CSRC, "raftq_sort_kernels.hpp"), os.path.join(CSRC, "raftq_wire_kernels.hpp"),
            os.path.join(CSRC, "raftq_kernels.hpp")]
    if not force and not _stale(SORT_CHECK, deps):
        return SORT_CHECK
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-o", SORT_CHECK, src]
    subprocess.check_call(cmd)
    return SORT_CHECK


def build_all(force: bool = False, log: list | None = None) -> None:
    """The library, the standalone tools and the sort checker, concurrently (every piece is its own hipcc process)."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(6) as ex:
        futs = [ex.submit(build_lib, force, None, False, log), ex.submit(build_sort_check, force)]
        futs += [ex.submit(build_tool, src, None, force) for src in tool_sources()]
        for f in futs:
            f.result()
    build_c_tool(os.path.join(TUNE_DIR, "turn_latency.c"), force)  # (links against the library built above)


def toolchain() -> dict:
    """What built the binaries: recorded beside the ISA report (profiles/rNN/isa_sweep.txt)."""
    v = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip().splitlines()
    return {"hipcc": _hipcc(), "version": v, "flags": _flags()}


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(LIB)
