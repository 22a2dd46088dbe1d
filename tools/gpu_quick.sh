#!/bin/bash
# quick GPU check of a change: selected tests, then the batch-1 graphed training step and the batch-32 step
#   gpurun -- 'bash tools/gpu_quick.sh <tag> "<pytest -k expression>"'
tag=${1:-q}; expr=${2:-coord}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -k "$expr" > gpurun_out/${tag}_tests.log 2>&1
tail -15 gpurun_out/${tag}_tests.log
timeout 240 python bench.py --mode train --batch 1 --steps 50 --warmup 5 --no-other-configs --details inline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=1 train eager ms', d['ms_per_step'], 'graphed', (d.get('hip_graph_replay') or {}).get('ms_per_step'))"
timeout 240 python bench.py --mode train --batch 32 --steps 30 --warmup 5 --no-other-configs --details inline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=32 train eager ms', d['ms_per_step'], 'graphed', (d.get('hip_graph_replay') or {}).get('ms_per_step'))"
