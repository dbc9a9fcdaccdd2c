#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_wire_gpu.py -m gpu -x -q > $P/gpu_tests_h.log 2>&1; echo "wire tests rc=$? $(tail -1 $P/gpu_tests_h.log)"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wdec -o w -- python tools/profile_wire_dec.py > /dev/null 2> $P/wdec.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/wdec_pmc -o sq1 -- python tools/profile_wire_dec.py > /dev/null 2> $P/wdec_sq1.err
python - <<'PY'
import csv,collections
for r in csv.DictReader(open('gpurun_out/r03/wdec/w_kernel_stats.csv')):
    if 'dec' in r['Name']: print(r['Name'][:40].ljust(40), r['Calls'], round(float(r['AverageNs'])/1000,2), r['MinNs'], r['MaxNs'])
rows=sorted(csv.DictReader(open('gpurun_out/r03/wdec/w_kernel_trace.csv')), key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows if 'wire_dec_kernel' in r['Kernel_Name']]
print('wire_dec_kernel no-entries (first 20):', sorted(d[:20])[10], ' mixed (last 20):', sorted(d[-20:])[10])
agg=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/r03/wdec_pmc/sq1_counter_collection.csv')):
    if 'wire_dec_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k:(round(sum(v[:20])/20/1028), round(sum(v[-20:])/20/1028)) for k,v in agg.items()})
PY
