#!/bin/bash
# Round-2 profile set, run ON THE GPU BOX (gpurun): kernel-trace stats of the driver's bench command, PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE, separate, --kernel-trace only) for every BASELINE config + the calibration
# copy, SQ / TCC counters of the headline kernel.  Everything lands under gpurun_out/r02/prof/.
set -u
P=gpurun_out/r02/prof
mkdir -p $P
export TMPDIR=/tmp
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
python -c "import json,subprocess,datetime; json.dump({'commit': '$(cat .git_head 2>/dev/null)', 'date': datetime.datetime.utcnow().isoformat()+'Z'}, open('$P/meta.json','w'))"
# 1. the same command under --kernel-trace --stats: per-kernel average duration must agree with the bench's own events
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- $BENCH > $P/bench_under_rocprof.json 2> $P/stats.err
# 2. PMC traffic, one counter per pass
for c in 3 2 4 5; do
  $BENCH --config $c > $P/bench_config$c.json 2> $P/bench_config$c.err
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $P/pmc -o config${c}_$ctr -- python bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --config $c > /dev/null 2> $P/pmc_config${c}_$ctr.err
  done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $P/pmc -o calib_$ctr -- tools/tune/raftq_tune3 1 calib > /dev/null 2> $P/pmc_calib_$ctr.err
done
# 3. where the waves' time goes + L2-side request bytes of the headline kernel
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $P/pmc -o sq1 -- python bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $P/pmc_sq1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $P/pmc -o sq2 -- python bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $P/pmc_sq2.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $P/pmc -o tcc -- python bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $P/pmc_tcc.err
# 4. the rows either side of the sweep: one batching turn, the codecs, Step
rocprofv3 --kernel-trace --stats --output-format csv -d $P/cycle -o cycle -- python tools/profile_cycle.py > $P/cycle.out 2> $P/cycle.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/wire -o wire -- python tools/profile_wire.py > $P/wire.out 2> $P/wire.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/step -o step -- python tools/profile_step.py > $P/step.out 2> $P/step.err
# 5. the full bench line (extras, CPU baseline) and smoke(), not under the profiler
python bench.py --gpus 1 --steps 20 --warmup 5 > $P/bench_n1.json 2> $P/bench_n1.err
python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.out 2>&1
python bench.py --gpus 2 --device 0 --steps 10 --warmup 3 --no-cpu-baseline --batches 20 > $P/bench_2gpus_worth_one_process.json 2> $P/bench_2gpus.err
du -sh $P
