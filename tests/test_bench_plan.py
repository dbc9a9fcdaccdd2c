"""CPU: how bench.py maps `--gpus N` onto devices (it must launch N GPUs' worth of work
itself or refuse -- never print n_gpus: 1 for --gpus 8), and that the GPU-less refusal is loud."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_plan_devices_direct_launch_drives_all_gpus_or_refuses():
    import bench

    assert bench.plan_devices(1, None, 1, 1, 0) == [0]
    assert bench.plan_devices(4, None, 8, 1, 0) == [0, 1, 2, 3]
    assert bench.plan_devices(8, None, 8, 1, 0) == list(range(8))
    with pytest.raises(SystemExit) as ei:
        bench.plan_devices(8, None, 1, 1, 0)
    assert "refusing" in str(ei.value) and "8" in str(ei.value)
    with pytest.raises(SystemExit):
        bench.plan_devices(2, None, 0, 1, 0)
    # testing override: every "GPU" of the job mapped onto one device
    assert bench.plan_devices(2, 0, 1, 1, 0) == [0, 0]
    with pytest.raises(SystemExit):
        bench.plan_devices(2, 3, 1, 1, 0)


def test_plan_devices_under_torchrun_is_one_device_per_rank():
    import bench

    assert bench.plan_devices(8, None, 8, 8, 5) == [5]
    assert bench.plan_devices(2, 0, 1, 2, 1) == [0]  # --device: both ranks on GPU 0 (gloo test on a 1-GPU box)
    with pytest.raises(SystemExit) as ei:
        bench.plan_devices(8, None, 8, 4, 0)  # launcher and flag disagree
    assert "WORLD_SIZE" in str(ei.value)
    with pytest.raises(SystemExit):
        bench.plan_devices(8, None, 4, 8, 6)  # local rank 6 has no GPU


def test_bytes_per_decision_is_the_design_table():
    import bench

    want = {2: (32, 8), 3: (50, 8.25), 4: (66, 8.25), 5: (56, 8)}  # packed votes: 2 B in, 0.25 B out (r01: N in, 1 out)
    for c, rw in want.items():
        assert bench.bytes_per_decision(bench.CONFIGS[c]) == rw


def test_bench_without_a_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    for n in ("1", "8"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n, "--steps", "2"], capture_output=True,
                           text=True, timeout=300, cwd=ROOT)
        assert p.returncode != 0 and "needs a GPU" in p.stderr and "{" not in p.stdout


def test_numa_pinning_helper_degrades_quietly():
    """bench.gpu_numa_cpus: None when there is no GPU / no sysfs entry -- the bench then simply does not pin."""
    import bench

    assert bench.gpu_numa_cpus(0) is None or len(bench.gpu_numa_cpus(0)) > 0


def test_pack_msgs40_layout_and_ranges():
    """raftq_msg40_t as the host mirror packs it: aux is the RejectHint on MsgAppResp and the LogTerm otherwise; ids that
    do not fit are refused."""
    import numpy as np
    import pytest

    from raftsql_amd import step as S

    m = S.pack_msgs(np.array([7, 8, 9], np.uint64), np.array([S.MSG_APP_RESP, S.MSG_VOTE, S.MSG_HEARTBEAT], np.uint8), term=5,
                    frm=2, index=11, log_term=4, commit=3, reject=1, reject_hint=10)
    p = S.pack_msgs40(m)
    assert p.dtype.itemsize == 40 and p.tobytes()[:8] == bytes([7, 0, 0, 0, 2, S.MSG_APP_RESP, 1, 0])
    assert p["aux"].tolist() == [10, 4, 4] and p["term"].tolist() == [5, 5, 5] and p["commit"].tolist() == [3, 3, 3]
    out = np.zeros(3, S.MSG40_DT)
    assert S.pack_msgs40(m, out=out) is out and out.tobytes() == p.tobytes()
    m["group"][1] = 1 << 32
    with pytest.raises(ValueError):
        S.pack_msgs40(m)


def test_bench_py_names_resolve():
    """bench.py only runs on the GPU box: a name that no longer exists (a function lost in an edit) would show up there, as the
    driver's empty record.  Every name loaded anywhere in the file is a builtin, an import, a module-level definition or a local."""
    import ast
    import builtins

    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    known = set(dir(builtins)) | {"__file__", "__name__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            known.add(n.name)
        elif isinstance(n, ast.Import):
            known |= {a.asname or a.name.split(".")[0] for a in n.names}
        elif isinstance(n, ast.ImportFrom):
            known |= {a.asname or a.name for a in n.names}
        elif isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            known.add(n.id)
        elif isinstance(n, ast.arg):
            known.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            known.add(n.name)
    missing = sorted({n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known})
    assert not missing, missing
