#!/bin/bash
# kernel sequence of ONE graphed batch-1 training step -> gpurun_out/<tag>_trace_b1_graphed.txt
tag=${1:-b1g}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/trb1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trb1 -- python3 bench.py --mode train --batch 1 --steps 3 --warmup 2 --no-other-configs > gpurun_out/${tag}_run.log 2>&1
f=$(find /tmp/trb1 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_seq.py $f > gpurun_out/${tag}_trace_b1_graphed.txt 2>&1
wc -l gpurun_out/${tag}_trace_b1_graphed.txt
