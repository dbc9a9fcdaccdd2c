"""GPU: bench.py keeps the driver's contract -- one JSON line on stdout with the agreed keys, the
metric BASELINE.json names, a roofline object whose numbers are self-consistent, and a CPU baseline
that was really timed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_keys_and_consistency(gpu_engine_cls):
    d = run_bench("--steps", "300", "--warmup", "30", "--no-extras")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(" across")[0] in base["metric"]
    assert d["unit"] == "decisions/s" and d["n_gpus"] == 1 and d["steps"] == 300 and d["warmup"] == 30
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "1M groups x 5 peers" in cfg["workload"] and cfg["groups_per_gpu"] == 1 << 20 and cfg["peers"] == 5
    assert cfg["rotating_bytes_per_gpu"] > 4 * 256 * 2**20  # the working set is >> the 256 MiB Infinity Cache
    # value = groups * steps / wall time
    assert abs(d["value"] - cfg["groups_per_gpu"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["bytes_per_decision"] == {"read": 53, "write": 9} and r["bytes_per_launch"] == 62 * (1 << 20)
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6
    assert 0.3 < r["frac"] < 1.0
    # the event-derived launch time and the wall clock tell the same story (back-to-back launches)
    assert 0.7 < r["launch_us"] / (d["ms_per_step"] * 1e3) < 1.1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "decisions/s" and c["cores"] >= 1 and c["value"] > 1e6
    assert d["value"] > 50 * c["value"]  # sanity: orders of magnitude, not a precision claim


def test_bench_other_configs_run(gpu_engine_cls):
    for cfg in (2, 5):
        d = run_bench("--steps", "200", "--warmup", "20", "--config", str(cfg), "--no-extras", "--no-cpu-baseline")
        assert d["cpu_baseline"] is None and d["roofline"]["frac"] > 0.3
        assert f"config{cfg}" in d["config"]["workload"]
