#!/bin/bash
# Same-box A/B of library builds on the training step:  bash tools/train_ab.sh <tag> <reps> <variant> [<variant> ...]   ("base" = shipped)
# every variant in turn, <reps> rounds; bench.py --mode train --batch ${EG_B:-32} --steps 30 (eager ms_per_step)  -> gpurun_out/<tag>_train_ab.txt
tag=$1; reps=$2; shift 2
for r in $(seq $reps); do for v in "$@"; do
  lib=echoglad_amd/lib/libechoglad_hip.$v.so; [ "$v" == "base" ] && lib=echoglad_amd/lib/libechoglad_hip.so
  ms=$(ECHOGLAD_LIB=$lib timeout 300 python bench.py --mode train --batch ${EG_B:-32} --steps 30 --warmup 5 --no-other-configs --no-graph-replay 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "round $r  $v  $ms" | tee -a gpurun_out/${tag}_train_ab.txt
done; done
