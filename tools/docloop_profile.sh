#!/bin/bash
# per-kernel table of INTEGRATION.md E's graphed loop at batch 1 -> gpurun_out/<tag>_docloop_stats.txt
tag=${1:-dl}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/dl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dl -- python3 tools/tools_docloop.py 1 > gpurun_out/${tag}_docloop_run.txt 2>&1
f=$(find /tmp/dl -name "*kernel_stats.csv" | head -1)
python3 - $f > gpurun_out/${tag}_docloop_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cat gpurun_out/${tag}_docloop_run.txt | head -6; cat gpurun_out/${tag}_docloop_stats.txt
