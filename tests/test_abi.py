"""CPU: the C-ABI shared library loads and exports every symbol that
include/raftq.h declares; argument validation and the no-GPU error path work
without touching a device.  No compute is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from raftsql_amd import _lib, build

    build.build_lib()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _declared(header="raftq.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(raftq_[a-z_]+)\s*\(", hdr)))


def test_header_and_binding_list_the_same_symbols():
    from raftsql_amd import _lib, node, pipe

    assert _declared("raftq_node.h") == sorted(node.EXPORTS)
    assert _declared() == sorted(_lib.EXPORTS)
    assert _declared("raftq_pipe.h") == sorted(pipe.EXPORTS)
    assert _declared("raftq_step.h") == sorted(_lib.STEP_EXPORTS)
    assert _declared("raftq_wire.h") == sorted(_lib.WIRE_EXPORTS)


def test_every_declared_symbol_is_exported(lib):
    for name in _declared() + _declared("raftq_pipe.h") + _declared("raftq_step.h") + _declared("raftq_node.h") + \
            _declared("raftq_wire.h"):
        assert hasattr(lib, name), name


def test_pipe_refuses_without_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from raftsql_amd.engine import RaftqError
    from raftsql_amd.pipe import MultiRaftPipe

    with pytest.raises(RaftqError) as ei:
        MultiRaftPipe(8, 3)
    assert ei.value.code in (-5, -3)


def test_abi_version_and_quorum(lib):
    assert lib.raftq_abi_version() == 1
    assert [lib.raftq_quorum(n) for n in range(1, 10)] == [1, 2, 2, 3, 3, 4, 4, 5, 5]


def test_argument_validation_without_device(lib):
    from raftsql_amd import _lib

    h = C.c_void_p(None)
    assert lib.raftq_create(0, 0, 3, C.byref(h)) == _lib.RAFTQ_EINVAL
    assert lib.raftq_create(0, 1024, 0, C.byref(h)) == _lib.RAFTQ_EINVAL
    assert lib.raftq_create(0, 1024, 10, C.byref(h)) == _lib.RAFTQ_EINVAL
    assert lib.raftq_create(0, 1024, 3, None) == _lib.RAFTQ_EINVAL
    assert b"n_peers" in lib.raftq_last_error(None) or b"null" in lib.raftq_last_error(None)
    assert lib.raftq_step_async(None, 1) == _lib.RAFTQ_EINVAL
    assert lib.raftq_wait(None, None) == _lib.RAFTQ_EINVAL
    lib.raftq_destroy(None)  # must be a no-op
    assert lib.raftq_step_batch(None, None, 0, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_set_self(None, 0) == _lib.RAFTQ_EINVAL
    assert lib.raftq_apply_log_deltas(None, None, 0, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_wire_encode(None, None, 0, None, 0, None, 0, None, 0, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_wire_decode(None, None, 0, None, 0, None, None, 0, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_wal_encode(None, None, 0, None, 0, 0, None, 0, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_wal_decode(None, None, 0, None, 0, 0, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_step_submit_wire(None, None, 0, None, 0) == _lib.RAFTQ_EINVAL
    assert lib.raftq_wire_scan_frames(None, 8, 1, None, 0, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_step_set_compact(None, 1) == _lib.RAFTQ_EINVAL
    assert lib.raftq_step_results_c(None, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_set_create(None, 0, C.byref(h)) == _lib.RAFTQ_EINVAL and lib.raftq_set_create(None, 3, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_set_sweep_async(None, 1) == _lib.RAFTQ_EINVAL and lib.raftq_set_wait(None, None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_set_mode(None, 0, 0) == _lib.RAFTQ_EINVAL and lib.raftq_set_size(None) == 0
    assert lib.raftq_set_timer_begin(None) == _lib.RAFTQ_EINVAL and lib.raftq_set_timer_end(None, None) == _lib.RAFTQ_EINVAL
    assert lib.raftq_sweep_many_async(None, 2, 1) == _lib.RAFTQ_EINVAL and lib.raftq_sweep_many_async(None, 0, 1) == _lib.RAFTQ_OK
    assert lib.raftq_clone_state(None, None) == _lib.RAFTQ_EINVAL
    lib.raftq_set_destroy(None)  # no-op
    assert b"set" in lib.raftq_set_last_error(None)
    p = C.c_void_p(None)
    assert lib.raftq_host_alloc(None, 64) == _lib.RAFTQ_EINVAL and lib.raftq_host_alloc(C.byref(p), 0) == _lib.RAFTQ_EINVAL
    lib.raftq_host_free(None)  # no-op
    from raftsql_amd import node

    nl = node._load()
    assert nl.raftq_node_wal_enable(None) == _lib.RAFTQ_EINVAL
    assert nl.raftq_node_wal_poll(None, None, 0, None) == _lib.RAFTQ_EINVAL
    assert nl.raftq_node_replay_wal(None, None, 0, 0, None) == _lib.RAFTQ_EINVAL


def test_no_silent_cpu_fallback(lib):
    """Without a GPU the product must refuse, not compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible; the refusal path is for GPU-less hosts")
    from raftsql_amd.engine import QuorumEngine, RaftqError

    with pytest.raises(RaftqError) as ei:
        QuorumEngine(4096, 5)
    assert ei.value.code in (-5, -3)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's CPU legs may execute it.
    Nothing under raftsql_amd/, include/, go/ or tools/ may name it; bench.py and __graft_entry__.py import it inside
    functions only (never at module level, so importing either never loads the oracle)."""
    import re

    for top in ("raftsql_amd", "include", "go", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".go", ".sh")):
                    path = os.path.join(dirpath, f)
                    txt = open(path).read()
                    if f.endswith((".py", ".sh")):
                        assert "pyoracle" not in txt and "raftq_oracle" not in txt and "pywire" not in txt, path
                        assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), path
                    else:  # compiled sources may mention the oracle in a comment, never include or link it
                        assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', txt), path
                        assert "rq_oracle_" not in txt and "liboracle" not in txt, path
    for f in ("bench.py", "__graft_entry__.py"):
        txt = open(os.path.join(ROOT, f)).read()
        assert not re.search(r"^(from|import)\s+oracle\b", txt, flags=re.M), f  # module level


def test_package_has_no_switch_to_a_test_double():
    """the oracle-backed engine of tests/test_hostsim.py is the tests' business.  Nothing under
    raftsql_amd/ names it, and the loader refuses any RAFTQ_LIB outside the package directory."""
    import subprocess
    import sys

    for dirpath, _, files in os.walk(os.path.join(ROOT, "raftsql_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                assert "hostsim" not in open(os.path.join(dirpath, f)).read().lower(), os.path.join(dirpath, f)
    r = subprocess.run([sys.executable, "-c", "from raftsql_amd import _lib; _lib.load()"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, RAFTQ_LIB=os.path.join(ROOT, "tests", "c", "libraftq_hostsim.so")))
    assert r.returncode != 0 and "only builds of libraftq inside" in r.stderr, r.stderr[-600:]


def test_group_bound_is_the_tested_one(lib):
    """raftq_create accepts what one handle has been compared at (tests/test_envelope_gpu.py: 2^29 + 70001), no more."""
    import ctypes as C
    import re

    hdr = open(os.path.join(ROOT, "include", "raftq.h")).read()
    assert re.search(r"#define RAFTQ_MAX_GROUPS \(1ull << 30\)", hdr)
    env = open(os.path.join(ROOT, "tests", "test_envelope_gpu.py")).read()
    assert "(1 << 29) + 70001" in env
    h = C.c_void_p()
    assert lib.raftq_create(0, (1 << 30) + 1, 3, C.byref(h)) == -1 and not h.value  # refused before any device is touched


def test_headers_are_plain_c99_and_link_from_c(lib, tmp_path):
    """cgo compiles its preamble as C: all public headers must pass a pedantic C99 compiler and the
    library must link and answer from a C program (no C++ runtime needed by the caller)."""
    import subprocess

    from raftsql_amd import _lib

    exe = str(tmp_path / "abi_c99")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_c99.c"), "-o", exe, _lib.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "C99-ABI-OK" in out.stdout, (out.returncode, out.stdout, out.stderr)
