#!/bin/bash
# One parameterised GPU-box script (replaces round 3's eighteen tools/gpu_r03_*.sh, profile_r0N.sh and sanitize_r03.sh): run as
#   tools/gpu.sh T 'bash tools/gpurun_trip.sh <step> [<step> ...]'      (gpu.sh stamps the shipped tree into .git_head first)
# Every step writes under gpurun_out/$ROUND/; the summaries worth keeping are copied to profiles/$ROUND/ by hand.
set -u
ROUND=${ROUND:-r06}
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
    pmchead)   # HBM traffic of the headline kernel, every BASELINE config: FETCH_SIZE / WRITE_SIZE in separate passes + the calibration
               # copy (tools/pmc_traffic.py turns the directory into profiles/pmc_traffic.json)
      H=$P/pmc_head; mkdir -p $H
      python -c "import json,datetime; json.dump({'commit': '$(cat .git_head 2>/dev/null)', 'dirty': '$(cat .git_dirty 2>/dev/null)' == '1', 'date': datetime.datetime.utcnow().isoformat()+'Z'}, open('$H/meta.json','w'))"
      for c in 3 2 4 5; do
        $BENCH --no-extras --no-cpu-baseline --config $c > $H/bench_config$c.json 2> $H/bench_config$c.err
        for ctr in FETCH_SIZE WRITE_SIZE; do
          rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $H -o config${c}_$ctr -- python bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --config $c > /dev/null 2> $H/pmc_config${c}_$ctr.err
        done
      done
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $H -o calib_$ctr -- tools/tune/raftq_tune3 1 calib > /dev/null 2> $H/pmc_calib_$ctr.err
      done
      python tools/pmc_traffic.py $H $P/pmc_traffic.json | grep traffic_over ;;
    sanitize)  # the library's host C++ under UBSan / TSan / ASan, driven by the GPU suites that exercise it (libraries built in-tree by
               # raftsql_amd/build.py build_sanitized; KINDS="ubsan tsan"; the GPU pool refuses AddressSanitizer runs: ASan runs on the CPU build, tests/test_hostsim.py)
      cat > /tmp/tsan.supp <<'SUPP'
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
called_from_lib:libtorch_hip.so
called_from_lib:libtorch_cpu.so
called_from_lib:libc10.so
race:libamdhip64.so
race:libhsa-runtime64.so
SUPP
      TESTS="tests/test_pipe_gpu.py tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_parity_gpu.py::test_cycle_is_all_or_nothing_across_both_kinds tests/test_parity_gpu.py::test_cycle_zero_copy_staging tests/test_step_gpu.py::test_step_pipelined_submit_collect tests/test_wire_gpu.py::test_step_from_frames_staged_in_place"
      for kind in ${KINDS:-ubsan tsan}; do
        RT=$(python -c "from raftsql_amd import build as b; print(b.sanitizer_runtime('$kind'))"); LOG=$P/sanitize_$kind.log; rm -f $P/${kind}_report*
        echo "== $kind: LD_PRELOAD=$RT RAFTQ_LIB=raftsql_amd/libraftq_$kind.so ==" > $LOG
        case $kind in
          ubsan) env LD_PRELOAD=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$P/ubsan_report RAFTQ_LIB=$PWD/raftsql_amd/libraftq_ubsan.so \
                   timeout 1500 python -m pytest $TESTS -m gpu -q -p no:cacheprovider >> $LOG 2>&1 ;;
          tsan)  env LD_PRELOAD=$RT TSAN_OPTIONS=halt_on_error=0:suppressions=/tmp/tsan.supp:log_path=$P/tsan_report:report_signal_unsafe=0 RAFTQ_LIB=$PWD/raftsql_amd/libraftq_tsan.so \
                   timeout 1500 python -m pytest tests/test_pipe_gpu.py tests/test_node_gpu.py -m gpu -q -p no:cacheprovider -k "not host-memory" >> $LOG 2>&1 ;;
        esac
        echo "rc=$?" >> $LOG; ls $P/${kind}_report* >> $LOG 2>&1 || echo "no $kind report files: clean" >> $LOG; tail -3 $LOG
      done ;;
    legs)      # the side legs of the bench, plain (no profiler): wire, step, cycle, tick
      for l in wire step cycle tick frames; do CPU=0 timeout 300 python tools/profile_$l.py > $P/leg_$l.txt 2>&1; echo "$l rc=$?"; done ;;
    legstats)  # rocprofv3 kernel stats of the same legs
      for l in wire step cycle tick frames; do
        CPU=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$l -o $l -- python tools/profile_$l.py > /dev/null 2>&1
        cp $(find /tmp/ks_$l -name "*kernel_stats.csv" | head -1) $P/${l}_kernel_stats.csv 2>/dev/null; done ;;
    tests)     timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $P/gpu_tests.log 2>&1; echo "rc=$? $(tail -1 $P/gpu_tests.log)" ;;
    wiretests) timeout 900 python -m pytest tests/test_wire_gpu.py -m gpu -x -q --timeout 240 > $P/gpu_tests_wire.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_wire.log)" ;;
    ticktests) timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_set_gpu.py tests/test_abi_gpu.py -m gpu -x -q -k "tick or election or abi or drive" > $P/gpu_tests_tick.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_tick.log)" ;;
    tileab)    # the streaming decoder's launch shapes, one process: tile 128 / 256, readers, chunk, no readers at all, the SDMA-reader form
      timeout 600 python tools/probe/wire_tile_ab.py > $P/wire_tile_ab.jsonl 2> $P/wire_tile_ab.err; echo "rc=$?"; cut -c1-220 $P/wire_tile_ab.jsonl ;;
    soak400)   # the bench's node legs 400 times, a fresh process each, 0 give-ups wanted
      timeout 3000 python tools/probe/node_legs_soak.py ${SOAK_RUNS:-400} > $P/node_legs_soak.txt 2>&1; echo "rc=$?"; tail -3 $P/node_legs_soak.txt ;;
    soakstatic) # ... and with round 4's chunk ownership by position (libraftq_static_chunks.so, built by raftsql_amd/build.py build_lib(variant=)):
               # the soak must still be able to see the bug it guards against
      RAFTQ_LIB=$PWD/raftsql_amd/libraftq_static_chunks.so timeout 1500 python tools/probe/node_legs_soak.py ${SOAK_STATIC_RUNS:-120} > $P/node_legs_soak_static_chunks.txt 2>&1; echo "rc=$?"; grep -c FAILED $P/node_legs_soak_static_chunks.txt; tail -2 $P/node_legs_soak_static_chunks.txt ;;
    soakpad)   # the same soak under ROUND 5's residency (decoder workgroups padded back to 102 KB of LDS: one per CU), where round 4's chunk
               # ownership starved: libraftq_pad_static.so (pad + RAFTQ_WIRE_STATIC_CHUNKS: must FAIL some runs -- the soak sees the bug)
               # and libraftq_pad.so (pad only: tickets + workers that serve themselves -- must not)
      RAFTQ_LIB=$PWD/raftsql_amd/libraftq_pad_static.so timeout 1500 python tools/probe/node_legs_soak.py ${SOAK_PAD_RUNS:-100} > $P/node_legs_soak_pad_static_chunks.txt 2>&1; echo "static rc=$? failed: $(grep -c FAILED $P/node_legs_soak_pad_static_chunks.txt)"; tail -1 $P/node_legs_soak_pad_static_chunks.txt
      RAFTQ_LIB=$PWD/raftsql_amd/libraftq_pad.so timeout 1500 python tools/probe/node_legs_soak.py ${SOAK_PAD_RUNS:-100} > $P/node_legs_soak_pad.txt 2>&1; echo "tickets + self-serve rc=$? failed: $(grep -c FAILED $P/node_legs_soak_pad.txt)"; tail -1 $P/node_legs_soak_pad.txt ;;
    flagab)    # the completion word three ways (one-thread kernel | write-value packet | last workgroup to arrive), checked and timed
      timeout 1500 bash tools/probe/flag_ab.sh > $P/flag_ab.txt 2>&1; echo "rc=$?"; cat $P/flag_ab.txt ;;
    steptests) timeout 900 python -m pytest tests/test_step_gpu.py tests/test_envelope_gpu.py tests/test_parity_gpu.py -m gpu -x -q --timeout 300 > $P/gpu_tests_step.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_step.log)" ;;
    nodetests) timeout 900 python -m pytest tests/test_node_gpu.py tests/test_node_scenarios_gpu.py tests/test_pipe_gpu.py -m gpu -x -q > $P/gpu_tests_node.log 2>&1; echo "rc=$? $(tail -3 $P/gpu_tests_node.log)" ;;
    bench)     # the driver's command: stdout = the ONE contract line (<= 4 KB), the full record (every side leg) beside it
      $BENCH --legs-out $P/bench_legs.json > $P/bench_n1.json 2> $P/bench_n1.err; echo "rc=$? $(wc -c < $P/bench_n1.json) bytes on stdout"; python - <<PY
import json; d = json.loads(open("$P/bench_n1.json").read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["value"] if d["cpu_baseline"] else None); print(d.get("legs"))
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
    onenode)   # one node on one GPU, scripted peers (bench.one_node_measure), as one handle and as four shard handles; with the phase clock
      SHARDS=1,4 timeout 300 python tools/profile_one_node.py > $P/one_node.json 2> $P/one_node.err; echo "rc=$?"
      RAFTQ_PROFILE=1 SHARDS=1 timeout 300 python tools/profile_one_node.py 2>&1 | grep "advance phases" | cut -c1-400 > $P/one_node_phases.txt; cat $P/one_node_phases.txt ;;
    ablate)    # the streaming decoder with its outputs taken away (measurement build libraftq_wiretrace.so: build_lib(variant="wiretrace", defines={"RAFTQ_WIRE_TRACE": 1}))
      RAFTQ_LIB=$PWD/raftsql_amd/libraftq_wiretrace.so timeout 300 python tools/probe/wire_ablate.py > $P/wire_ablate.jsonl 2> $P/wire_ablate.err; echo "rc=$?"; cut -c1-230 $P/wire_ablate.jsonl; tail -3 $P/wire_ablate.err ;;
    nodeab)    # the three-node leg with the proposals' MsgApps built on the device (shipped) and on the host (round 5's way), alternated
      for v in 1 0 1 0 1 0; do
        echo "RAFTQ_NODE_PROPOSE_DEVICE=$v $(RAFTQ_NODE_PROPOSE_DEVICE=$v timeout 300 python tools/profile_node.py 2>&1 | grep -o "'proposals_committed_everywhere_per_s': [0-9.e+]*" | head -1)"
      done > $P/three_nodes_ab.txt; cat $P/three_nodes_ab.txt ;;
    onenodestats)  # ... under rocprofv3: the device calls of a turn (one handle)
      SHARDS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/on -o node -- python tools/profile_one_node.py > $P/one_node_under_rocprof.txt 2>&1
      cp $(find /tmp/on -name "*kernel_stats.csv" | head -1) $P/node_kernel_stats.csv; head -12 $P/node_kernel_stats.csv | cut -c1-160 ;;
    timeline)  # the batching turn as a timeline of kernels: the last turns of tools/profile_cycle.py (24-byte, packed, segmented: the segmented ones are last)
      timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o cyc -- python tools/profile_cycle.py > /dev/null 2>&1
      python tools/probe/timeline.py /tmp/tl -12 12 > $P/cycle_timeline.txt; python tools/probe/timeline.py /tmp/tl -340 12 >> $P/cycle_timeline.txt; cat $P/cycle_timeline.txt ;;
    turnsoak)  # >= 10^4 batching turns with the completion flag checked against a full synchronisation every turn (RAFTQ_CYCLE_CHECK)
      RAFTQ_CYCLE_CHECK=1 CYCLES=4000 timeout 600 python tools/profile_cycle.py > $P/turn_soak.txt 2>&1; echo "rc=$? (4 x 4000 turns: 24-byte, copying, packed, segmented)"; tail -c 300 $P/turn_soak.txt ;;
    stepsoak)  SECONDS_BUDGET=${SECONDS_BUDGET:-90} timeout 600 python tests/soak/step_stress.py > $P/soak_step.txt 2>&1; echo "rc=$?"; tail -3 $P/soak_step.txt ;;
    soak)      RAFTQ_CYCLE_CHECK=1 timeout 900 python tests/soak/soak.py > $P/soak.txt 2>&1; echo "rc=$?"; tail -5 $P/soak.txt ;;
    *)         if [ -f "$step" ]; then bash "$step"; else echo "unknown step $step"; fi ;;
  esac
done
du -sh $P
