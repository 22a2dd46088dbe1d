#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/trace_inf
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_inf -- python3 bench.py --steps 60 --warmup 10 --repeats 0 --no-cpu-baseline --no-other-configs > gpurun_out/r5p.log 2>&1
f=$(find /tmp/trace_inf -name "*kernel_trace.csv" | head -1)
python3 tools/trace_infer_seq.py $f | tee gpurun_out/r5p_seq.txt
