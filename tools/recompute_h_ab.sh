#!/bin/bash
# A/B of the "h of the last layer is never written in full" route (nn.ROUTES.heads_recompute_h) on the cfg4_train step; tests first.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "heads or head" > gpurun_out/rh_tests.log 2>&1
tail -5 gpurun_out/rh_tests.log
timeout 600 python tools/recompute_h_ab.py > gpurun_out/rh_ab.txt 2>&1
cat gpurun_out/rh_ab.txt
