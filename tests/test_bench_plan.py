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
