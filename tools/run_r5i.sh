#!/bin/bash
OUT=gpurun_out
export TMPDIR=/tmp
for w in conn; do
rm -rf /tmp/prof_cfg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg -- python3 tools/tools_cfg_profile.py $w 30 > $OUT/r5i_$w.log 2>&1
tail -1 $OUT/r5i_$w.log
f=$(find /tmp/prof_cfg -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %5s avg %8.1f us total %8.3f ms" % (r["Name"].replace("void ","").replace("eg::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
