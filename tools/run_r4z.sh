#!/bin/bash
OUT=gpurun_out
python -m pytest tests -x -q -m gpu > $OUT/r4z_pytest.log 2>&1; tail -3 $OUT/r4z_pytest.log
python3 tools/tools_determinism.py 300 > $OUT/r4z_determinism.log 2>&1; tail -8 $OUT/r4z_determinism.log
for k in 1 0 1 0; do
EG_QUEUE_SELF_RESET=$k python3 bench.py --no-other-configs --no-cpu-baseline 2> $OUT/r4z_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EG_QUEUE_SELF_RESET=$k', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])"
done
