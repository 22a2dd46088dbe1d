#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_engine.py tests/test_gpu_losses.py -x -q -m gpu > $OUT/r5f_pytest.log 2>&1; tail -2 $OUT/r5f_pytest.log
for k in 1 2; do
python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r5f_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['ms_per_step'], d['value'])"
done
bash tools/tools_train_profile.sh 32 2>&1 | grep -i "bce_partial\|cls_out_fwd\|total kernel\|layer_ps<false, false, 1\|hm_partial"
