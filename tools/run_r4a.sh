#!/bin/bash
# round 4, run A: GPU test suite + training-step A/B (child sums in the train forward, fused coordinate nodes)
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/r4a_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/r4a_pytest.log
tail -15 $OUT/r4a_pytest.log
for cfg in "default" "EG_TRAIN_CHAIN=0" "EG_COORD_FUSED=0" "EG_ACT_TILES=1 EG_TRAIN_CHAIN=0"; do
  echo "== $cfg" >> $OUT/r4a_train.log
  if [ "$cfg" = "default" ]; then
    python bench.py --mode train --batch 32 --steps 10 --warmup 3 >> $OUT/r4a_train.log 2>&1
  else
    env $cfg python bench.py --mode train --batch 32 --steps 10 --warmup 3 >> $OUT/r4a_train.log 2>&1
  fi
done
grep -o '"ms_per_step": [0-9.]*\|^== .*' $OUT/r4a_train.log
