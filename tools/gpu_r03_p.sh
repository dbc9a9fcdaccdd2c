#!/bin/bash
# rocprofv3 kernel stats of the bench's node leg (three raftq_nodes on one GPU)
P=gpurun_out/r03/node_prof; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -o node -- python $GRAFT_REPO_ROOT/tools/profile_node.py > /tmp/np.out 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/np -name "*kernel_stats.csv" | head -1) $P/node_kernel_stats.csv
grep -o "'proposals_committed_everywhere_per_s': [0-9.]*" /tmp/np.out | head -1
cut -d, -f1-5 $P/node_kernel_stats.csv | cut -c1-160 | head -24
