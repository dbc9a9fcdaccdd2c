"""GPU: the boundary driven by a plain C host (tests/c/drive_gpu.c, gcc -std=c99) -- what a cgo caller is --
through sweep, ingest + advance list, batched Step, stream frames and a WAL segment, against hand-derived answers."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_drives_the_whole_boundary(gpu_engine_cls, tmp_path):
    from raftsql_amd import _lib

    exe = str(tmp_path / "drive_gpu")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "drive_gpu.c"), "-o", exe, _lib.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C-HOST-GPU-OK" in out.stdout, (out.returncode, out.stdout, out.stderr[-2000:])
