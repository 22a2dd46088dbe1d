#!/bin/bash
# selected tests + kernel sequence of the graphed batch-1 training step:  gpurun -- 'bash tools/gpu_quick2.sh <tag> "<pytest -k expression>"'
tag=${1:-q}; expr=${2:-coord}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -k "$expr" > gpurun_out/${tag}_tests.log 2>&1
tail -5 gpurun_out/${tag}_tests.log
bash tools/trace_b1_graphed.sh $tag
timeout 240 python bench.py --mode train --batch 1 --steps 50 --warmup 5 --no-other-configs --details inline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=1 train eager ms', d['ms_per_step'], 'graphed', (d.get('hip_graph_replay') or {}).get('ms_per_step'))"
