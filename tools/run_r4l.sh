#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_diag.py tests/test_gpu_pyg_surface.py tests/test_gpu_pack.py -m gpu -x -q > $OUT/r4l_diag.log 2>&1; tail -15 $OUT/r4l_diag.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --repeats 1 > $OUT/r4l_bench.json 2>$OUT/r4l_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4l_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["other_configs"].items(): print(k, {a:b for a,b in v.items() if a not in ("workload","roofline","traffic_source","floor")})
PY
