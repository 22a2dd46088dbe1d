#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_engine.py -x -q -m gpu > $OUT/r5c_pytest.log 2>&1; tail -3 $OUT/r5c_pytest.log
for k in 1 2; do
python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r5c_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['ms_per_step'], d['value'])"
done
bash tools/tools_train_profile.sh 32 2>&1 | head -12
