#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_contract_gpu.py -m gpu -x -q > $P/gpu_tests_m.log 2>&1; echo "contract tests rc=$? $(tail -1 $P/gpu_tests_m.log)"
: > $P/bench_repeat.txt
for i in 1 2 3 4 5 6; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('run $i: value %.4e decisions/s  ms_per_step %.4f  launch_us %.2f  frac %.4f  read frac %.4f  kernel/wall %.3f' % (d['value'], d['ms_per_step'], r['launch_us'], r['frac'], r['frac_read_of_peak'], r['kernel_time_over_wall']))" >> $P/bench_repeat.txt
done
cat $P/bench_repeat.txt
