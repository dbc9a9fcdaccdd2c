#!/bin/bash
# One parameterised GPU-box script (replaces round 3's eighteen tools/gpu_r03_*.sh): run as
#   gpurun --timeout T -- 'bash tools/gpurun_trip.sh <step> [<step> ...]'
# Every step writes under gpurun_out/$ROUND/; the summaries worth keeping are copied to profiles/$ROUND/ by hand.
set -u
ROUND=${ROUND:-r04}
P=gpurun_out/$ROUND; mkdir -p $P; export TMPDIR=/tmp
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5"
for step in "$@"; do
  echo "== $step ($(date +%T))"
  case $step in
    probe)     # what the PCIe link does under workgroup copies / SDMA, one direction and both (tools/probe/pcie_duplex_probe.hip)
      timeout 300 tools/probe/pcie_duplex_probe > $P/pcie_duplex_probe.jsonl 2> $P/pcie_duplex_probe.err; echo "rc=$? $(wc -l < $P/pcie_duplex_probe.jsonl) lines" ;;
    single)    # north_star's literal shape, one 1M x 5 launch at a time: tiles / streams / graphs (tools/tune/single_launch_ab.hip)
      timeout 300 tools/tune/single_launch_ab > $P/single_launch_ab.jsonl 2> $P/single_launch_ab.err; echo "rc=$?"; cat $P/single_launch_ab.jsonl ;;
    pmc)       # FETCH / WRITE / fabric-request counters of the kernels either side of the sweep (tools/pmc_legs.py)
      timeout 900 python tools/pmc_legs.py collect /tmp/pmc_legs > $P/pmc_legs_collect.log 2>&1
      python tools/pmc_legs.py summarise /tmp/pmc_legs $P/pmc_traffic_legs.json > $P/pmc_legs_summary.txt 2>&1; tail -5 $P/pmc_legs_collect.log ;;
    legs)      # the side legs of the bench, plain (no profiler): wire, step, cycle, tick
      for l in wire step cycle tick; do CPU=0 timeout 300 python tools/profile_$l.py > $P/leg_$l.txt 2>&1; echo "$l rc=$?"; done ;;
    legstats)  # rocprofv3 kernel stats of the same legs
      for l in wire step cycle tick; do
        CPU=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$l -o $l -- python tools/profile_$l.py > /dev/null 2>&1
        cp $(find /tmp/ks_$l -name "*kernel_stats.csv" | head -1) $P/${l}_kernel_stats.csv 2>/dev/null; done ;;
    tests)     timeout 1500 python -m pytest tests -m gpu -x -q > $P/gpu_tests.log 2>&1; echo "rc=$? $(tail -1 $P/gpu_tests.log)" ;;
    wiretests) timeout 900 python -m pytest tests/test_wire_gpu.py -m gpu -x -q > $P/gpu_tests_wire.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_wire.log)" ;;
    steptests) timeout 900 python -m pytest tests/test_step_gpu.py tests/test_envelope_gpu.py tests/test_parity_gpu.py -m gpu -x -q > $P/gpu_tests_step.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_step.log)" ;;
    nodetests) timeout 900 python -m pytest tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_pipe_gpu.py -m gpu -x -q > $P/gpu_tests_node.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_node.log)" ;;
    bench)     $BENCH > $P/bench_n1.json 2> $P/bench_n1.err; echo "rc=$?"; python - <<PY
import json; d = json.load(open("$P/bench_n1.json")); print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
PY
      ;;
    benchstats) # the driver's command under --kernel-trace --stats (the roofline's average launch duration must agree)
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -o bench -- $BENCH --no-extras --no-cpu-baseline > $P/bench_under_rocprof.json 2> $P/bench_stats.err
      cp $(find /tmp/bs -name "*kernel_stats.csv" | head -1) $P/bench_kernel_stats.csv ;;
    smoke)     python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.out 2>&1; echo "rc=$? $(tail -1 $P/smoke.out)" ;;
    node)      NODE_THREADS=1 timeout 600 python tools/node_profile.py > $P/node_profile.txt 2>&1; echo "rc=$?"; tail -5 $P/node_profile.txt ;;
    nodestats)
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -o node -- python tools/profile_node.py > $P/node_under_rocprof.txt 2>&1
      cp $(find /tmp/np -name "*kernel_stats.csv" | head -1) $P/node_kernel_stats.csv ;;
    soak)      RAFTQ_CYCLE_CHECK=1 timeout 900 python tests/soak/soak.py > $P/soak.txt 2>&1; echo "rc=$?"; tail -5 $P/soak.txt ;;
    *)         if [ -f "$step" ]; then bash "$step"; else echo "unknown step $step"; fi ;;
  esac
done
du -sh $P
