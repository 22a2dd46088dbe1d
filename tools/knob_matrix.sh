# the fallback routes behind the run-time knobs must stay correct: the GPU suite once under each of them (usage: gpurun -- bash tools/knob_matrix.sh)
for kv in "EG_LAYER_IMPL=0" "EG_CSR_TILES=0" "EG_CSR_TILES=1" "EG_SEQ_FUSED=0" "EG_QUEUE_SELF_RESET=0" "EG_TRAIN_PS=0" "EG_CHAIN=0 EG_FUSE_CLS=0" "EG_TRAIN_CHAIN=0 EG_ACT_HEADS=0 EG_COORD_FUSED=0" "EG_FB_DIRECT=0 EG_CLS_MASKED=0 EG_LAYER_SUMS_IN_HEADS=0"; do
  echo "== $kv"
  env $kv timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
done
