#!/bin/bash
OUT=gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/trace_train
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_train -- python3 bench.py --mode train --batch 32 --steps 8 --warmup 3 --no-other-configs > $OUT/r4s_trace.log 2>&1
f=$(find /tmp/trace_train -name "*kernel_trace.csv" | head -1)
python3 tools/trace_seq.py $f > $OUT/r4s_seq.txt
python3 tools/trace_gaps.py $f 5 | head -3
