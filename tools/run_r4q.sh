#!/bin/bash
# round 4: activation pass of the last layer inside the heads' first kernel -- tests, then A/B of the training step
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "folded_in or inside_the_heads or cfg4_train or classifier_train or chained_train" > $OUT/r4q_pytest.log 2>&1
tail -5 $OUT/r4q_pytest.log
for knob in 1 0 1 0; do
  EG_ACT_HEADS=$knob python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r4q_bench_$knob.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EG_ACT_HEADS=$knob', d['ms_per_step'], d['value'])"
done
