#!/bin/bash
# Same-box A/B of a run-time knob on the training step:  bash tools/knob_ab.sh <tag> <reps> <ENV_NAME> <value> [<value> ...]
# every value in turn, <reps> rounds; bench.py --mode train --batch ${EG_B:-32} --steps 30 (eager ms_per_step) -> gpurun_out/<tag>_knob_ab.txt
tag=$1; reps=$2; name=$3; shift 3
for r in $(seq $reps); do for v in "$@"; do
  ms=$(env $name=$v timeout 300 python bench.py --mode train --batch ${EG_B:-32} --steps 30 --warmup 5 --no-other-configs --no-graph-replay 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "round $r  $name=$v  $ms" | tee -a gpurun_out/${tag}_knob_ab.txt
done; done
