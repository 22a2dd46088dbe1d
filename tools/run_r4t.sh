#!/bin/bash
OUT=gpurun_out
python3 tools/tools_hosttime.py 32 2>&1 | grep -v amdgpu.ids | tee $OUT/r4t_hosttime.txt
for i in 1 2; do
python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r4t_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', d['ms_per_step'], d['value'])"
done
