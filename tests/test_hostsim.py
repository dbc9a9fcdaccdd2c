"""CPU: the host C++ of the library -- raftq_node.cpp and raftq_pipe.cpp, 1,900 lines of mutexes, condition variables, a
background thread, arenas and queues -- under AddressSanitizer + UBSan, driven by the very suites the GPU box runs against
libraftq.so (tests/test_node_gpu.py, tests/test_node_scenarios_gpu.py, tests/test_pipe_gpu.py).

ASan cannot run beside the HIP runtime on this image (profiles/r03/sanitizers_host_cpp.txt: its HSA interceptors abort),
so here the two translation units are linked with tests/c/engine_sim.cpp, which answers the engine's C-ABI on the CPU with
the oracle -- test infrastructure, not a CPU path of the product: the library exists only under tests/c/ and the product's
loader cannot be pointed at it (tests/conftest.py swaps the path for the sub-run, RAFTQ_TEST_ENGINE_DOUBLE).  On the GPU box the same suites run against the real engine (and under UBSan
and TSan, tools/gpurun_trip.sh sanitize).  Besides the sanitizer, this gives the CPU suite the node's and the pipe's behaviour
tests: election safety, log matching, replay + sentinel, chaos, WAL restart, etcd's network scenarios as recalled."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c")
LIB = os.path.join(CDIR, "libraftq_hostsim.so")
CSRC = os.path.join(ROOT, "raftsql_amd", "csrc")
SOURCES = [os.path.join(CSRC, "raftq_node.cpp"), os.path.join(CSRC, "raftq_pipe.cpp"), os.path.join(CDIR, "engine_sim.cpp")]
ORACLE = [os.path.join(ROOT, "oracle", f) for f in ("raftq_oracle.c", "raftq_step_oracle.c", "raftq_wire_oracle.c")]
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-I" + CSRC]


def _build(san=None, lib=None, bdir_name="build_hostsim") -> str:
    """san / lib / bdir_name: tools/hostsim_opt.py builds the same library without sanitizers to time the host phases"""
    SAN = globals()["SAN"] if san is None else san
    LIB = globals()["LIB"] if lib is None else lib
    deps = SOURCES + ORACLE + [os.path.join(ROOT, "include", h) for h in os.listdir(os.path.join(ROOT, "include"))]
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps):
        return LIB
    bdir = os.path.join(CDIR, bdir_name)
    os.makedirs(bdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        o = os.path.join(bdir, os.path.basename(src) + ".o")
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-Wall", "-Wextra"] + SAN + INC + ["-c", src, "-o", o])
        objs.append(o)
    for src in ORACLE:
        o = os.path.join(bdir, os.path.basename(src) + ".o")
        subprocess.check_call(["gcc", "-std=c11", "-fPIC", "-pthread"] + SAN + INC + ["-c", src, "-o", o])
        objs.append(o)
    # every other name of the ABI: present (the loader binds the whole table) and inert
    sys.path.insert(0, ROOT)
    from raftsql_amd import _lib

    first = os.path.join(bdir, "first.so")
    subprocess.check_call(["g++", "-shared", "-pthread"] + SAN + ["-o", first] + objs)
    have = set(re.findall(r" T (\w+)", subprocess.run(["nm", "-D", "--defined-only", first], capture_output=True, text=True).stdout))
    names = [s[0] for s in _lib._SIGS + _lib._STEP_SIGS + _lib._WIRE_SIGS]
    stubs = os.path.join(bdir, "stubs.c")
    with open(stubs, "w") as f:
        f.write("/* generated: entry points of the ABI that the node / pipe suites never reach -- RAFTQ_ENODEV */\n")
        for nme in sorted(set(names) - have):
            f.write("int %s(void) { return -5; }\n" % nme)
    so = os.path.join(bdir, "stubs.o")
    subprocess.check_call(["gcc", "-std=c11", "-fPIC", "-c", stubs, "-o", so])
    subprocess.check_call(["g++", "-shared", "-pthread"] + SAN + ["-o", LIB] + objs + [so])
    return LIB


def test_node_threads_under_tsan():
    """the crank (raftq_crank_step: every node's turn and the transport on the library's own threads), the threaded
    cluster and raftq_node_forward under ThreadSanitizer, the oracle as the engine"""
    import shutil

    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip() if shutil.which("gcc") else ""
    if not (shutil.which("g++") and os.path.isabs(tsan) and os.path.exists(tsan)):
        pytest.skip("no g++ / libtsan here")
    lib = _build(san=["-fsanitize=thread", "-O1", "-g"], lib=os.path.join(CDIR, "libraftq_hostsim_tsan.so"), bdir_name="build_hostsim_tsan")
    env = dict(os.environ, RAFTQ_TEST_ENGINE_DOUBLE=lib, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "tests/test_node_gpu.py", "-k",
                        "threaded or forward or (chaos and True)"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in out, out[out.index("WARNING: ThreadSanitizer"):][:4000]
    assert r.returncode == 0, out[-4000:]
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) >= 3, out[-2000:]


def _run(tests, extra_env=None, timeout=1500):
    import shutil

    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip() if shutil.which("gcc") else ""
    if not (shutil.which("g++") and os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("no g++ / libasan here: the host C++ cannot be built under AddressSanitizer")
    lib = _build()
    env = dict(os.environ, RAFTQ_TEST_ENGINE_DOUBLE=lib, LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               **(extra_env or {}))
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0, out[-6000:]
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) > 0, out[-2000:]
    return int(m.group(1))


def test_pipe_suite_under_asan_ubsan():
    assert _run(["tests/test_pipe_gpu.py", "-k", "not host-memory"]) >= 5


def test_node_suite_under_asan_ubsan():
    assert _run(["tests/test_node_gpu.py"]) >= 15


def test_node_scenarios_under_asan_ubsan():
    assert _run(["tests/test_node_scenarios_gpu.py"]) >= 20
