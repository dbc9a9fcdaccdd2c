"""GPU: bench.py keeps the driver's contract -- one JSON line on stdout with the agreed keys, the
metric BASELINE.json names, a roofline object whose numbers are self-consistent, a CPU baseline
that was really timed -- at the DRIVER'S OWN arguments (--steps 20 --warmup 5), where the timed
region must be kernel time, not host launch time."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 4096  # round 4's line was 24.9 KB and the driver's record came back `parsed: null`
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def read_line(stdout, legs_path):
    """What the driver does: the LAST stdout line, parsed on its own.  It must be the only JSON line, at most 4 KB, carry every
    contract key with roofline / cpu_baseline as flat objects, and agree with the full record written beside it; -> the full
    record (every side leg's objects) with the line's own objects in place of the full ones, the line under "_line"."""
    out = stdout.rstrip("\n").splitlines()
    assert out, "nothing on stdout"
    line_text = out[-1]
    assert [ln for ln in out if ln.lstrip().startswith("{")] == [line_text], stdout[-2000:]
    assert len(line_text.encode()) <= LINE_LIMIT, len(line_text)
    line = json.loads(line_text)
    for k in CONTRACT:
        assert k in line, k
    for k, v in line["roofline"].items():  # scalars, the two byte counts and one entry per GPU: nothing nested
        assert not isinstance(v, dict) or k == "bytes_per_decision", k
    assert all(isinstance(v, (int, float)) or k == "errors" for k, v in line.get("legs", {}).items())
    full = json.load(open(legs_path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data"):
        assert full[k] == line[k], k
    assert full["roofline"]["frac"] == line["roofline"]["frac"] and full["roofline"]["launch_us"] == line["roofline"]["launch_us"]
    if full["cpu_baseline"] is not None:
        assert full["cpu_baseline"]["value"] == line["cpu_baseline"]["value"]
    merged = dict(full)
    merged["_line"] = line
    return merged


def run_bench(*args, expect_rc=0):
    import tempfile

    env = {k: v for k, v in os.environ.items() if k != "RAFTQ_CYCLE_CHECK"}  # the bench measures the turn as shipped
    legs = os.path.join(tempfile.mkdtemp(prefix="raftq_bench_"), "legs.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--legs-out", legs], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    if expect_rc != 0:
        assert p.returncode != 0, p.stdout[-500:]
        return p
    assert p.returncode == 0, p.stderr[-2000:]
    return read_line(p.stdout, legs)


def check_line(d, n_gpus, steps, warmup, sharing=1):
    """sharing: how many of the job's "GPUs" were mapped onto one device (testing only): each dispatch then runs that much slower"""
    for k in CONTRACT:
        assert k in d, k
    if "_line" in d:  # the same checks hold for the line the driver parses (its objects are trimmed, never recomputed)
        check_line(d["_line"], n_gpus, steps, warmup, sharing)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].split(" across")[0] in base["metric"]
    assert d["unit"] == "decisions/s" and d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "1M groups x 5 peers" in cfg["workload"] and cfg["groups_per_batch"] == 1 << 20 and cfg["peers"] == 5
    assert cfg["groups_per_gpu"] == cfg["groups_per_batch"] * cfg["batches_per_gpu"] and cfg["launches_per_step"] == 1
    assert cfg["resident_bytes_per_gpu"] > 4 * 256 * 2**20  # the working set is >> the 256 MiB Infinity Cache
    # value = groups * steps / wall time, whole job
    assert abs(d["value"] - cfg["groups_per_gpu"] * n_gpus * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["bytes_per_decision"] == {"read": 50, "write": 8.25} and r["bytes_per_launch"] == 58.25 * cfg["groups_per_gpu"]
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-6
    assert abs(r["achieved_read_GBps"] - r["achieved"] * 50 / 58.25) / r["achieved"] < 1e-6
    assert "sweep_set_kernel<5, 8, true, false, true, 3, true, 256>" in r["kernel"]
    assert 0.3 / sharing < r["frac"] < 1.0
    return cfg, r


def test_bench_at_the_drivers_arguments(gpu_engine_cls):
    d = run_bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras")
    cfg, r = check_line(d, 1, 20, 5)
    # the timed region is the kernel: one launch per step, issued from C, events on the set's own stream
    assert abs(r["kernel_time_over_wall"] - r["launch_us"] / (d["ms_per_step"] * 1e3 / cfg["launches_per_step"])) < 1e-9
    assert r["kernel_time_over_wall"] >= 0.95, r
    assert d["value"] >= 8.3e10, d["value"]  # VERDICT r01 "done" line; round 1 printed 6.5e10 at these arguments
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "decisions/s" and c["cores"] >= 1 and c["value"] > 1e6
    assert d["value"] > 50 * c["value"]  # sanity: orders of magnitude, not a precision claim


def test_bench_extras_and_other_configs(gpu_engine_cls):
    d = run_bench("--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--batches", "20")
    check_line(d, 1, 10, 2)
    assert d["cpu_baseline"] is None
    assert d["single_launch"]["launches_per_step"] == 20 and d["single_launch"]["launch_us"] > d["roofline"]["per_batch_us"] * 0.9
    assert d["other_dispatch"]["dispatch"] == "persistent" and d["l3_resident"]["launch_us"] > 0
    assert [c["batches"] for c in d["footprint_curve"]] == [64, 128, 240]
    for c in ("config2", "config4", "config5"):
        assert d["other_configs"][c]["frac"] > 0.3, d["other_configs"][c]
    for leg in ("pipeline", "tick", "step", "wire", "node"):
        assert "error" not in d[leg], d[leg]
    # every side leg carries a roofline object of the headline's shape ...
    legs = [d["tick"]["roofline"], d["step"]["roofline"],
            d["step"]["pipelined"]["roofline"], d["step"]["pipelined"]["compact_results"]["roofline"],
            d["step"]["pipelined"]["producer_included"]["roofline_40B"], d["wire"]["message_frames"]["pinned"]["roofline_decode"],
            d["wire"]["step_from_frames"]["staged_in_device_memory"]["roofline_compact"],
            d["wire"]["wal_frames"]["pinned"]["roofline_encode"], d["tick"]["tick_and_lists"]["roofline"],
            d["tick"]["tick_and_lists"]["beat_bitmap"]["roofline"], d["tick"]["set_dispatch"]["steady_state"]["roofline"]]
    for r in legs:
        assert r["bound"] in ("pcie", "hbm") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert 0.0 < r["frac"] < 1.0, r
    # ... except the latency-bound ones (the batching turn): microseconds, launches and bytes, no fraction of a peak they are not
    # bound by; the shipped form (segments) leads, on the C caller's clock
    pl = d["pipeline"]
    for r in (pl["roofline"], pl["segmented_list"]["roofline"], pl["packed_records"]["roofline"], pl["wide_records"]["roofline"]):
        assert r["bound"] == "latency" and r["wall_us"] > 5 and "frac" not in r and r["bytes_per_turn"]["in"] > 0
    assert pl["roofline"] is pl["segmented_list"]["roofline"] or pl["roofline"] == pl["segmented_list"]["roofline"]
    assert pl["clock"].startswith("plain-C caller") and pl["us_per_cycle"] == pl["c_caller"]["us_per_turn_segmented_list"]
    assert pl["us_per_cycle"] <= pl["python_loop_us"]["segmented_list"] * 1.1
    # no leg quotes a measured traffic below what its arrays add up to, and none a partial sum
    def walk(o, path=""):
        if isinstance(o, dict):
            if "traffic_over_algorithmic" in o:
                assert o["traffic_over_algorithmic"] >= 0.95, (path, o["traffic_over_algorithmic"])
            if o.get("traffic") is None and "traffic_kernels" in o:
                raise AssertionError(path)
            for k, v in o.items():
                walk(v, path + "/" + k)
    walk({k: d[k] for k in ("pipeline", "tick", "step", "wire")})
    # the Tick in every launch shape, side by side
    assert set(d["tick"]["set_dispatch"]["shapes"]) == {"narrow", "wide1", "wide2", "wide4"}
    # the line the driver parses carries a few scalars per leg
    for k in ("turn_segmented_c_us", "tick_set_frac", "tick_lists_us", "step_msgs_per_s", "frames_decode_us", "node_proposals_per_s"):
        assert k in d["_line"]["legs"], (k, d["_line"]["legs"])
    assert 0.3 < d["single_launch"]["frac_read_of_peak"] < 1.0


def test_bench_gated_config_as_headline(gpu_engine_cls):
    d = run_bench("--steps", "10", "--warmup", "2", "--config", "5", "--no-extras", "--no-cpu-baseline", "--batches", "20")
    assert "config5" in d["config"]["workload"] and d["roofline"]["bytes_per_decision"] == {"read": 56, "write": 8}
    assert d["roofline"]["frac"] > 0.3 and "true, true, false" in d["roofline"]["kernel"]


def test_bench_four_gpus_worth_from_one_process_and_refusal(gpu_engine_cls):
    """`--gpus N` launched directly drives N devices itself (one launch thread per device); with fewer than N visible it
    must fail loudly instead of printing n_gpus: 1.  --device maps all four onto GPU 0 (testing only)."""
    import torch

    d = run_bench("--gpus", "4", "--device", "0", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--batches", "20")
    cfg, r = check_line(d, 4, 6, 2, sharing=4)
    assert "one process" in cfg["parallelism"] and "config4_whole_job" in d
    assert d["config4_whole_job"]["decisions_per_s"] > 1e9
    assert len(r["launch_us_per_gpu"]) == 4 and len(r["frac_per_gpu"]) == 4 and all(f > 0.05 for f in r["frac_per_gpu"])
    assert d["gate"]["sets_gated"] == 4 and d["gate"]["groups_advanced_per_step_whole_job"] > 4 * 20 * (1 << 18)
    assert cfg["ranks_seen"] == [0] and cfg["rendezvous"]["backend"] == "none"
    if torch.cuda.device_count() < 8:
        p = run_bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extras", expect_rc=1)
        assert "refusing" in (p.stderr + p.stdout)


def _free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _torchrun_bench(nproc, _port_hint, *args):
    port = _free_port()  # (a fixed port would collide with whatever else runs on the box)
    import tempfile

    env = dict({k: v for k, v in os.environ.items() if k != "RAFTQ_CYCLE_CHECK"}, MASTER_ADDR="127.0.0.1")
    legs = os.path.join(tempfile.mkdtemp(prefix="raftq_bench_"), "legs.json")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc),
                        *args, "--legs-out", legs], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    return read_line(p.stdout, legs)


def test_bench_four_ranks_under_torchrun_on_one_gpu(gpu_engine_cls):
    """The driver's N > 1 launch line (torch.distributed.run, one process per GPU) at its default arguments for the
    rendezvous: gloo group + shared-memory barrier, no RCCL anywhere.  All four ranks on GPU 0 (testing only)."""
    d = _torchrun_bench(4, 29533, "--steps", "6", "--warmup", "2", "--device", "0", "--no-extras", "--batches", "20")
    cfg, r = check_line(d, 4, 6, 2, sharing=4)
    assert "torchrun" in cfg["parallelism"] and cfg["ranks_seen"] == [0, 1, 2, 3]
    assert cfg["rendezvous"] == {"backend": "gloo", "barrier": "shm", "note": ""}
    assert len(r["launch_us_per_gpu"]) == 4 and len(r["wall_ms_per_rank"]) == 4 and d["gate"]["sets_gated"] == 4


def test_bench_two_ranks_under_torchrun_with_the_side_legs(gpu_engine_cls):
    """The driver's N > 1 command line has no --no-extras: rank 0 then measures its side legs while the other ranks wait at
    the next barrier, and every rank takes part in config 4's whole job.  Two ranks on GPU 0, a small resident set."""
    d = _torchrun_bench(2, 29536, "--steps", "4", "--warmup", "1", "--device", "0", "--batches", "10", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == [0, 1] and d["gate"]["sets_gated"] == 2
    assert d["config4_whole_job"]["decisions_per_s"] > 1e9 and "2 GPU(s)" in d["config4_whole_job"]["workload"]
    assert d["single_launch"]["launches_per_step"] == 10 and d["other_dispatch"]["dispatch"] == "persistent"
    assert [c["batches"] for c in d["footprint_curve"]] == [64, 128, 240]


def test_bench_rccl_asked_for_where_it_cannot_work(gpu_engine_cls):
    """`--backend nccl` with two ranks mapped onto ONE device: RCCL refuses that.  Every rank must agree -- before any
    RCCL rendezvous -- to stay on gloo, finish the job, and say why (a communicator failure must not
    sink a run whose data path needs no collective)."""
    d = _torchrun_bench(2, 29534, "--steps", "4", "--warmup", "1", "--device", "0", "--backend", "nccl", "--no-extras",
                        "--batches", "20")
    cfg, _ = check_line(d, 2, 4, 1, sharing=2)
    assert cfg["rendezvous"]["backend"] == "gloo" and "nccl asked for, gloo used" in cfg["rendezvous"]["note"]
    assert "shares GPU 0" in cfg["rendezvous"]["note"]


def test_bench_rccl_leg_on_hardware_with_one_rank(gpu_engine_cls):
    """The opt-in RCCL leg on this box's one GPU: torchrun with one rank and `--backend nccl` walks the whole rendezvous
    (gloo group, pre-checks, RCCL group beside it, warm-up all-reduce, RCCL barriers around the timed region, RCCL
    reductions) -- a one-rank communicator, so no xGMI traffic, but the RCCL code path on real hardware."""
    d = _torchrun_bench(1, 29535, "--steps", "4", "--warmup", "1", "--backend", "nccl", "--no-extras", "--batches", "20",
                        "--no-cpu-baseline")
    cfg, _ = check_line(d, 1, 4, 1)
    assert cfg["rendezvous"] == {"backend": "nccl", "barrier": "nccl", "note": ""}, cfg["rendezvous"]
    assert cfg["ranks_seen"] == [0]
