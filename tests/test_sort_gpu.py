"""GPU: the hand-written stable radix sort behind Step's sorted walk (raftsql_amd/csrc/raftq_sort_kernels.hpp) against
std::stable_sort -- tests/c/sort_check.hip, compiled here with hipcc like the library itself."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_radix_sort_matches_stable_sort(gpu_engine_cls, tmp_path):
    from raftsql_amd import build

    exe = str(tmp_path / "sort_check")
    cmd = [build._hipcc(), f"--offload-arch={build.ARCH}", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + build.CSRC, os.path.join(ROOT, "tests", "c", "sort_check.hip"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "ALL OK" in p.stdout and "MISMATCH" not in p.stdout
    assert p.stdout.count(" ok") >= 99
