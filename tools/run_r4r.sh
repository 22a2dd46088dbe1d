#!/bin/bash
OUT=gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "folded_in" > $OUT/r4r_pytest.log 2>&1; tail -2 $OUT/r4r_pytest.log
rm -rf /tmp/trace_train
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_train -- python3 bench.py --mode train --batch 32 --steps 8 --warmup 3 --no-other-configs > $OUT/r4r_trace.log 2>&1
f=$(find /tmp/trace_train -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f 5 | tee $OUT/r4r_gaps.txt
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"].split("(")[0][:70]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:16]:
    print("%-72s n=%4d avg %8.1f us" % (k,len(v),sum(v)/len(v)/1e3))
PY
