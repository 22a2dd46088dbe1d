#!/bin/bash
# round-4 profiles: inference step (kernel stats + PMC), training step (kernel stats + PMC)
OUT=gpurun_out
bash tools/profile_round.sh r04 > $OUT/r4n_profile_round.log 2>&1
bash tools/profile_train_pmc.sh r04 32 > $OUT/r4n_profile_train_pmc.log 2>&1
export TMPDIR=/tmp
rm -rf /tmp/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 --no-other-configs > $OUT/r4n_train_prof.log 2>&1
cp $(find /tmp/prof_train -name "*kernel_stats.csv" | head -1) $OUT/profiles/r04_train_kernel_stats.csv
tail -12 $OUT/r4n_profile_round.log | cut -c1-250
tail -25 $OUT/r4n_profile_train_pmc.log
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/profiles/r04_train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
N=12
print("total kernel ms per step: %.3f" % (tot/N/1e6))
for r in rows[:18]:
    print("%-62s calls/step %5.1f  avg %8.1f us  per-step %6.3f ms" % (r["Name"].replace("void ","").replace("eg::","")[:62], int(r["Calls"])/N, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/N/1e6))
PY
ls -la $OUT/profiles | tail -12
