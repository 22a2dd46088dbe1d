# the fallback routes behind the 7 run-time variables (include/echoglad_hip.h) must stay correct: the GPU suite once under each of them
# usage: gpurun --timeout 2400 -- 'bash tools/knob_matrix.sh > gpurun_out/knob_matrix.txt 2>&1'
for kv in "EG_LAYER_IMPL=0" "EG_CSR_TILES=0" "EG_CSR_TILES=1" "EG_SEQ_FUSED=0" "EG_TRAIN_PS=0" "EG_SUMS_DOWN=0" "EG_POOL_PYRAMID=0" "EG_FUSED_CRITERIA=0"; do
  echo "== $kv"
  env $kv timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
done
