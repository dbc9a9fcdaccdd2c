#!/bin/bash
# Round 3, GPU trip B: the flat frame decoder -- parity first, then the same kernel-trace + SQ passes as trip A.
set -u
P=gpurun_out/r03
mkdir -p $P
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wire_gpu.py tests/test_node_gpu.py tests/test_bench_contract_gpu.py -m gpu -x -q > $P/gpu_tests_b.log 2>&1
echo "rc=$?" >> $P/gpu_tests_b.log
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire1 -o wire -- python tools/profile_wire.py > $P/wire1.out 2> $P/wire1.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $P/wire1_pmc -o sq1 -- python tools/profile_wire.py > /dev/null 2> $P/wire1_sq1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $P/wire1_pmc -o sq2 -- python tools/profile_wire.py > /dev/null 2> $P/wire1_sq2.err
tail -3 $P/gpu_tests_b.log
grep -h "dec_kernel\|dec_ents" $P/wire1/wire_kernel_stats.csv | cut -d'"' -f2- | cut -c1-40,150-260
