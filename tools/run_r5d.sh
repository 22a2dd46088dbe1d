#!/bin/bash
# every A/B knob of round 4 switched off in turn: the previous forms still pass the training / model tests
OUT=gpurun_out
for knob in EG_ACT_HEADS EG_LAYER_SUMS_IN_HEADS EG_CLS_MASKED EG_ACT_LIN4 EG_QUEUE_SELF_RESET EG_TRAIN_CHAIN EG_COORD_FUSED; do
  env $knob=0 python -m pytest tests/test_gpu_train.py tests/test_gpu_engine.py tests/test_gpu_model.py -x -q -m gpu -k "not inside_the_heads and not chained_train_step" > $OUT/r5d_$knob.log 2>&1
  echo "$knob=0: $(tail -1 $OUT/r5d_$knob.log)"
done
python __graft_entry__.py smoke 2>&1 | tail -3
