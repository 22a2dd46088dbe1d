#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_engine.py tests/test_gpu_parallel.py -x -q -m gpu > $OUT/r5m_pytest.log 2>&1; tail -2 $OUT/r5m_pytest.log
for k in 1 0 1 0; do
EG_COORD_SIDE_STREAM=$k python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r5m_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train EG_COORD_SIDE_STREAM=$k', d['ms_per_step'], d['value'])"
done
python3 tools/tools_determinism.py 30 2>&1 | tail -1
