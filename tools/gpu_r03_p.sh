#!/bin/bash
# the whole GPU suite + smoke + the bench line on the final tree
P=gpurun_out/r03/final6; mkdir -p $P
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $P/gpu_tests.log 2>&1; echo "suite rc=$? $(tail -n 1 $P/gpu_tests.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
python bench.py --gpus 1 --steps 20 --warmup 5 > $P/bench_n1.json 2> $P/bench_n1.err; echo "bench rc=$?"
python tools/results_table.py $P/bench_n1.json | tail -n 4
