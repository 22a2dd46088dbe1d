#!/bin/bash
for knobs in "" "EG_LAYER_SUMS_IN_HEADS=0" "EG_ACT_HEADS=0" "EG_COORD_FUSED=0" "EG_TRAIN_CHAIN=0" "EG_CLS_MASKED=0" "EG_COORD_MLP_KERNEL=0"; do
  echo "== $knobs"
  env $knobs python3 tools/dbg_train_fp64.py 64 6 1 0 1 3 2>&1 | grep -v amdgpu | grep "logits\|gnn_layers.2.module_0\|gnn_layers.0.module_0\|node_classifiers.0.0.weight"
done
echo "== B=2"
python3 tools/dbg_train_fp64.py 64 6 1 0 2 3 2>&1 | grep -v amdgpu | grep "logits\|gnn_layers.2.module_0\|gnn_layers.0.module_0\|node_classifiers.0.0.weight"
echo "== B=1 no coord"
python3 tools/dbg_train_fp64.py 64 6 0 0 1 3 2>&1 | grep -v amdgpu | grep "logits\|gnn_layers.2.module_0\|gnn_layers.0.module_0\|node_classifiers.0.0.weight"
