#!/bin/bash
set -u
P=gpurun_out/r03; mkdir -p $P; export TMPDIR=/tmp
for lib in libraftq.so libraftq_gpl8.so; do
  for rep in 1 2; do
    RAFTQ_LIB=$PWD/raftsql_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/ab_single_$lib.$rep.json 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open('$P/ab_single_$lib.$rep.json') if l.startswith('{')][0])
s=d['single_launch']; o=d['other_dispatch']
print('$lib', 'single launch us %.2f read frac %.3f | headline %.3e frac %.3f | persistent per-batch %.2f' % (s['launch_us'], s['frac_read_of_peak'], d['value'], d['roofline']['frac'], o['per_batch_us']))
PY
  done
done
