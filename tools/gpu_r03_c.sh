#!/bin/bash
# Round 3, GPU trip C: wide compaction + in-kernel completion flag, trusted-turn semantics, /proc/self/maps BAR probe.
set -u
P=gpurun_out/r03
mkdir -p $P
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_envelope_gpu.py::test_one_handle_of_2_pow_28_groups > $P/gpu_tests_c.log 2>&1
echo "rc=$?" >> $P/gpu_tests_c.log
timeout 300 python -m pytest tests/test_envelope_gpu.py::test_one_handle_of_2_pow_28_groups -m gpu -x -q >> $P/gpu_tests_c.log 2>&1
echo "rc=$?" >> $P/gpu_tests_c.log
rocprofv3 --kernel-trace --stats --output-format csv -d $P/cycle1 -o cycle -- python tools/profile_cycle.py > $P/cycle1.out 2> $P/cycle1.err
python tools/profile_cycle.py > $P/cycle1_noprof.out 2>&1
RAFTQ_CYCLE_FLAG=packet python tools/profile_cycle.py > $P/cycle1_packet.out 2>&1
RAFTQ_STAGE=host python tools/profile_cycle.py > $P/cycle1_hoststage.out 2>&1
tail -4 $P/gpu_tests_c.log
grep -h "compact\|Name" $P/cycle1/cycle_kernel_stats.csv | cut -c1-60,150-330
grep -ho "'us_per_cycle': [0-9.]*\|us_per_cycle_copying_form': [0-9.]*" $P/cycle1_noprof.out $P/cycle1_packet.out $P/cycle1_hoststage.out
