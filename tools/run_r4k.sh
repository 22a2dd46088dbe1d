#!/bin/bash
OUT=gpurun_out
python -m pytest tests -m gpu -x -q > $OUT/r4k_pytest.log 2>&1; tail -4 $OUT/r4k_pytest.log
python bench.py --steps 50 --warmup 10 --no-other-configs > $OUT/r4k_bench.json 2>$OUT/r4k_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4k_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("parity"), d.get("cpu_baseline",{}).get("value"))
print(json.dumps(d["before_path"], indent=0)); print(d["after_path"])
PY
