#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu > $OUT/r4u_pytest.log 2>&1; tail -3 $OUT/r4u_pytest.log
for i in 1 2; do
python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r4u_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['ms_per_step'], d['value'])"
done
EG_CLS_MASKED=0 python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r4u_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train EG_CLS_MASKED=0', d['ms_per_step'], d['value'])"
bash tools/tools_train_profile.sh 32 2>&1 | tail -22
