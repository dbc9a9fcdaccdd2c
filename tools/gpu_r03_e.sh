#!/bin/bash
# Round 3, GPU trip E: whole suite (both staging forms, flag cross-check on), cycle with our flag kernel, decoder v2 with
# 32-bit offsets, the node leg's phase profile, sanitizer runs.
set -u
P=gpurun_out/r03
mkdir -p $P
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $P/gpu_tests_e.log 2>&1; echo "suite rc=$? $(tail -1 $P/gpu_tests_e.log)"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/cycle2 -o cycle -- python tools/profile_cycle.py > $P/cycle2.out 2>&1
python tools/profile_cycle.py > $P/cycle2_noprof.out 2>&1
RAFTQ_CYCLE_FLAG=packet python tools/profile_cycle.py > $P/cycle2_packet.out 2>&1
echo "cycle flag-kernel: $(grep -ho "'us_per_cycle': [0-9.]*" $P/cycle2_noprof.out | tr '\n' ' ') | packet: $(grep -ho "'us_per_cycle': [0-9.]*" $P/cycle2_packet.out | tr '\n' ' ')"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire3 -o wire -- python tools/profile_wire.py > $P/wire3.out 2> $P/wire3.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $P/wire3_pmc -o sq1 -- python tools/profile_wire.py > /dev/null 2> $P/wire3_sq1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $P/wire3_pmc -o sq2 -- python tools/profile_wire.py > /dev/null 2> $P/wire3_sq2.err
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03/wire3/wire_kernel_stats.csv')):
    if 'dec' in r['Name']: print(r['Name'][:40].ljust(40), r['Calls'], round(float(r['AverageNs'])/1000,2), r['MinNs'], r['MaxNs'])
PY
RAFTQ_PROFILE=1 NODE_THREADS=1 python tools/node_profile.py > $P/node_profile_threads.txt 2>&1
RAFTQ_PROFILE=1 python tools/node_profile.py > $P/node_profile_serial.txt 2>&1
tail -15 $P/node_profile_serial.txt
bash tools/sanitize_r03.sh $P
