#!/bin/bash
# Round 3, final tree (after the node work): the whole GPU suite, smoke(), the full bench line, the Step profile again (its
# kernel gained the tail append), the node leg three times + round 2's loop on round 3's runner, sanitizers, soaks.
set -u
P=gpurun_out/r03/final2; mkdir -p $P; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $P/gpu_tests.log 2>&1; echo "suite rc=$? $(tail -1 $P/gpu_tests.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $P/bench_n1.json 2> $P/bench_n1.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/step -o step -- python tools/profile_step.py > $P/step.out 2> $P/step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire -o wire -- python tools/profile_wire.py > $P/wire.out 2> $P/wire.err
for k in 1 0; do echo "== RAFTQ_WIRE_KERNEL_COPIES=$k"; RAFTQ_WIRE_KERNEL_COPIES=$k python tools/probe/codec_call_probe.py 2>&1 | grep -v amdgpu.ids; done > $P/codec_call_probe.txt
for i in 1 2 3; do
  RAFTQ_PROFILE=1 RAFTQ_PROFILE_EVERY=118 timeout 300 python tools/profile_node.py 2>&1 | grep -v "amdgpu.ids\|^wall" | cut -c1-1400
done > $P/node_leg.txt 2>&1
{ NODE_CRANK=1 python tools/node_profile.py; NODE_THREADS=1 python tools/node_profile.py; } 2>&1 | grep -v amdgpu.ids > $P/node_r02_loop.txt
bash tools/sanitize_r03.sh $P ubsan tsan > $P/sanitize.out 2>&1; tail -n 3 $P/sanitize_ubsan.log; tail -n 3 $P/sanitize_tsan.log
SECONDS_BUDGET=100 python tests/soak/soak.py > $P/soak_node.txt 2>&1; tail -1 $P/soak_node.txt
SECONDS_BUDGET=60 python tests/soak/step_stress.py > $P/soak_step.txt 2>&1; tail -1 $P/soak_step.txt
find $P -name "*trace.csv" -size +2M -delete
python tools/results_table.py $P/bench_n1.json | tail -8
