#!/bin/bash
# Round 3, GPU trip D: where does the compaction's time go -- fine-grained vs runtime-default host buffers, LDS-staged
# vs direct stores, in-kernel flag vs packet; with the flag cross-check on (a soak of turns), and the v2 frame decoder.
set -u
P=gpurun_out/r03
mkdir -p $P
export TMPDIR=/tmp
for coh in 1 0; do for stg in 1 0; do
  RAFTQ_HOST_COHERENT=$coh RAFTQ_COMPACT_STAGE=$stg rocprofv3 --kernel-trace --stats --output-format csv -d $P/cycleD_c${coh}_s${stg} -o cycle -- python tools/profile_cycle.py > $P/cycleD_c${coh}_s${stg}.out 2>&1
  RAFTQ_HOST_COHERENT=$coh RAFTQ_COMPACT_STAGE=$stg python tools/profile_cycle.py > $P/cycleD_c${coh}_s${stg}_noprof.out 2>&1
  echo "coherent=$coh staged=$stg: $(grep -h compact_changed_kernel $P/cycleD_c${coh}_s${stg}/cycle_kernel_stats.csv | awk -F, '{print $(NF-4)}' | tr '\n' ' ') | $(grep -ho "'us_per_cycle': [0-9.]*" $P/cycleD_c${coh}_s${stg}_noprof.out | tr '\n' ' ')"
done; done
# the flag cross-check over many turns, dense and sparse, both flag modes
for coh in 1 0; do
  RAFTQ_HOST_COHERENT=$coh RAFTQ_CYCLE_CHECK=1 timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_pipe_gpu.py tests/test_envelope_gpu.py -m gpu -q -k "cycle or pipe or 2_pow_28 or changed_list" > $P/flagcheck_c${coh}.log 2>&1
  echo "flag check coherent=$coh: $(tail -1 $P/flagcheck_c${coh}.log)"
done
RAFTQ_CYCLE_CHECK=1 timeout 300 python tools/profile_cycle.py > $P/cycleD_check.out 2>&1; echo "profile_cycle under check rc=$? $(tail -2 $P/cycleD_check.out | cut -c1-200)"
timeout 600 python -m pytest tests/test_wire_gpu.py tests/test_node_gpu.py -m gpu -x -q > $P/gpu_tests_d_wire.log 2>&1; echo "wire tests: $(tail -1 $P/gpu_tests_d_wire.log)"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire2 -o wire -- python tools/profile_wire.py > $P/wire2.out 2> $P/wire2.err
grep -h "dec_kernel\|dec_ents" $P/wire2/wire_kernel_stats.csv | awk -F'"' '{print substr($2,1,40), $0}' | awk -F, '{print $1, $(NF-6), $(NF-4), $(NF-3), $(NF-2)}' | cut -c1-40,300-
