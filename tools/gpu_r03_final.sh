#!/bin/bash
# Round 3, final tree: the whole GPU suite, smoke(), then the full profile set (tools/profile_r03.sh).
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $P/gpu_tests_final.log 2>&1; echo "suite rc=$? $(tail -1 $P/gpu_tests_final.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_r03.sh > $P/profile_r03.log 2>&1
python tools/results_table.py $P/prof/bench_n1.json | tail -6
