#!/bin/bash
OUT=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_diag.py tests/test_gpu_pyg_surface.py -x -q -m gpu > $OUT/r5k_pytest.log 2>&1; tail -2 $OUT/r5k_pytest.log
rm -rf /tmp/prof_cfg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg -- python3 tools/tools_cfg_profile.py conn 30 > $OUT/r5k_conn.log 2>&1
tail -1 $OUT/r5k_conn.log
f=$(find /tmp/prof_cfg -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print("%-70s calls %5s avg %8.1f us total %8.3f ms" % (r["Name"].replace("void ","").replace("eg::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
