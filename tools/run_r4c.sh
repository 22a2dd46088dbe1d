#!/bin/bash
# round 4, run C: GPU tests + the training step with the fused heads backward, and its kernel trace
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/r4c_pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/r4c_pytest.log
tail -12 $OUT/r4c_pytest.log
rm -f $OUT/r4c_train.log
for cfg in "default" "EG_TRAIN_CHAIN=0"; do
  echo "== $cfg" >> $OUT/r4c_train.log
  if [ "$cfg" = "default" ]; then python bench.py --mode train --batch 32 --steps 10 --warmup 3 >> $OUT/r4c_train.log 2>&1
  else env $cfg python bench.py --mode train --batch 32 --steps 10 --warmup 3 >> $OUT/r4c_train.log 2>&1; fi
done
grep -o '"ms_per_step": [0-9.]*\|^== .*' $OUT/r4c_train.log
export TMPDIR=/tmp
rm -rf /tmp/prof_r4c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4c -- python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $OUT/r4c_prof_run.log 2>&1
cp $(find /tmp/prof_r4c -name "*kernel_stats.csv" | head -1) $OUT/r4c_train_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4c_train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step: %.3f" % (tot/7/1e6))
for r in rows[:16]:
    print("%-60s calls/step %5.1f  avg %8.1f us  per-step %6.3f ms" % (r["Name"].replace("void ","").replace("eg::","")[:60], int(r["Calls"])/7, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/7/1e6))
PY
