#!/bin/bash
# Round 3, first GPU trip: the envelope + contract tests, the whole suite with durations, and the counter baseline of
# the frame decoders (VERDICT r02 item 3 asks for the SQ pass BEFORE the kernel is changed).
set -u
P=gpurun_out/r03
mkdir -p $P
export TMPDIR=/tmp
free -g > $P/box.txt; nproc >> $P/box.txt; rocm-smi --showmeminfo vram >> $P/box.txt 2>&1
timeout 1500 python -m pytest tests/test_envelope_gpu.py tests/test_bench_contract_gpu.py -m gpu -x -q --durations=15 > $P/gpu_tests_new.log 2>&1
echo "new tests rc=$?" >> $P/gpu_tests_new.log
timeout 1200 python -m pytest tests -m gpu -x -q --durations=30 --deselect tests/test_envelope_gpu.py --deselect tests/test_bench_contract_gpu.py > $P/gpu_tests_rest.log 2>&1
echo "rest rc=$?" >> $P/gpu_tests_rest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire0 -o wire -- python tools/profile_wire.py > $P/wire0.out 2> $P/wire0.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $P/wire0_pmc -o sq1 -- python tools/profile_wire.py > /dev/null 2> $P/wire0_sq1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $P/wire0_pmc -o sq2 -- python tools/profile_wire.py > /dev/null 2> $P/wire0_sq2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/cycle0 -o cycle -- python tools/profile_cycle.py > $P/cycle0.out 2> $P/cycle0.err
du -sh $P
tail -5 $P/gpu_tests_new.log $P/gpu_tests_rest.log
