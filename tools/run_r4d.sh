#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_engine.py tests/test_gpu_pyg_surface.py -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
rm -rf /tmp/prof_r4d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4d -- python3 tools/tools_heads.py > $OUT/r4d_heads.log 2>&1
cat $OUT/r4d_heads.log | grep heads
python3 - <<'PY'
import csv, glob
f=glob.glob("/tmp/prof_r4d/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-70s calls %4d avg %8.1f us" % (r["Name"].replace("void ","").replace("eg::","")[:70], int(r["Calls"]), float(r["AverageNs"])/1e3))
PY
python bench.py --mode train --batch 32 --steps 10 --warmup 3 2>&1 | grep -o '"ms_per_step": [0-9.]*'
