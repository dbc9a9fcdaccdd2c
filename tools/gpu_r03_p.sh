#!/bin/bash
mkdir -p gpurun_out/r03
{
python tools/probe/codec_call_probe.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
REPS=50 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03/codec_call -o cc -- python $GRAFT_REPO_ROOT/tools/probe/codec_call_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
cd $GRAFT_REPO_ROOT
find gpurun_out/r03/codec_call -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | cut -c1-150'
find gpurun_out/r03/codec_call -name "*memory_copy_stats.csv" | head -1 | xargs cat
find gpurun_out/r03/codec_call -name "*trace.csv" -size +3M -delete
} > gpurun_out/r03/codec_call_probe.txt 2>&1
cat gpurun_out/r03/codec_call_probe.txt
