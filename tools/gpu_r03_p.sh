#!/bin/bash
# timeline of the batching turn (kernels + copies)
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/cy -o cy -- python $GRAFT_REPO_ROOT/tools/profile_cycle.py > /tmp/cy.out 2>&1
cd $GRAFT_REPO_ROOT
{ tail -n 3 /tmp/cy.out | cut -c1-600; python tools/probe/timeline.py /tmp/cy -70 70; } > gpurun_out/r03/cycle_timeline.txt 2>&1
cat gpurun_out/r03/cycle_timeline.txt
