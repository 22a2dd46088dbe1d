#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_diag.py -m gpu -x -q > $OUT/r4j_diag.log 2>&1; tail -15 $OUT/r4j_diag.log
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_diag.py > $OUT/r4j_pytest.log 2>&1; tail -5 $OUT/r4j_pytest.log
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --repeats 2 > $OUT/r4j_bench.json 2>$OUT/r4j_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4j_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["other_configs"].items(): print(k, {a:b for a,b in v.items() if a not in ("workload","roofline")})
PY
