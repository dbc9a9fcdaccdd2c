#!/bin/bash
# the node leg INSIDE the whole bench line (after the other legs have used and freed hundreds of MB), twice; then alone
P=gpurun_out/r03; mkdir -p $P
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=d['node']
print('in bench:', n['proposals_committed_everywhere_per_s'], n['ms_per_wave_each'], n['waves_in_flight_4']['proposals_committed_everywhere_per_s'])"; done > $P/node_in_bench.txt 2>&1
timeout 300 python tools/profile_node.py 2>&1 | grep -o "'proposals_committed_everywhere_per_s': [0-9.]*, 'msgs_stepped\|'ms_per_wave_each': \[[0-9., ]*\]" | head -2 >> $P/node_in_bench.txt
cat $P/node_in_bench.txt
