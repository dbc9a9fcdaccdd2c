"""GPU: the hand-written stable radix sort behind Step's sorted walk (raftsql_amd/csrc/raftq_sort_kernels.hpp) against
std::stable_sort -- tests/c/sort_check.hip, built in-tree by raftsql_amd.build (hipcc, like the library itself; rebuilt
here only if its sources are newer than the binary)."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_radix_sort_matches_stable_sort(gpu_engine_cls):
    from raftsql_amd import build

    exe = build.build_sort_check()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "ALL OK" in p.stdout and "MISMATCH" not in p.stdout
    assert p.stdout.count(" ok") >= 99
