#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_losses.py tests/test_gpu_engine.py tests/test_gpu_parallel.py -x -q -m gpu > $OUT/r4v_pytest.log 2>&1; tail -3 $OUT/r4v_pytest.log
for i in 1 2 3; do
python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r4v_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['ms_per_step'], d['value'])"
done
export TMPDIR=/tmp
rm -rf /tmp/trace_train
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_train -- python3 bench.py --mode train --batch 32 --steps 8 --warmup 3 --no-other-configs > $OUT/r4v_trace.log 2>&1
f=$(find /tmp/trace_train -name "*kernel_trace.csv" | head -1)
python3 tools/trace_seq.py $f > $OUT/r4v_seq.txt
python3 tools/trace_gaps.py $f 5 | head -12
