#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu > $OUT/r5b_pytest.log 2>&1; tail -3 $OUT/r5b_pytest.log
for k in 1 0 1 0; do
EG_ACT_LIN4=$k python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r5b_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train EG_ACT_LIN4=$k', d['ms_per_step'], d['value'])"
done
bash tools/tools_train_profile.sh 32 2>&1 | head -12
