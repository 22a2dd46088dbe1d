#!/bin/bash
OUT=gpurun_out
for v in base abl base abl; do
  if [ $v = abl ]; then export ECHOGLAD_LIB=echoglad_amd/lib/libechoglad_hip.ablhash.so; else unset ECHOGLAD_LIB; fi
  python3 bench.py --mode train --batch 32 --steps 20 --warmup 5 --no-other-configs 2> $OUT/r5a_bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
done
