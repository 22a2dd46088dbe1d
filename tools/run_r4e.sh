#!/bin/bash
OUT=gpurun_out
EG_PARITY_FACTOR=1000 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "fp64" -s 2>&1 | tail -22
python bench.py --mode train --batch 32 --steps 10 --warmup 3 > $OUT/r4e_train.json 2>$OUT/r4e_train.err; tail -c 1800 $OUT/r4e_train.json
python bench.py --steps 30 --warmup 5 > $OUT/r4e_bench.json 2>$OUT/r4e_bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4e_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"])
print(json.dumps(d["other_configs"], indent=0)[:3000])
PY
