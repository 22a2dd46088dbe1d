#!/bin/bash
# Diagnostic: kernel-level breakdown of one training step (bench.py --mode train), per-step milliseconds.
# usage (GPU box, repo root): bash tools/tools_train_profile.sh [batch]
export TMPDIR=/tmp
B=${1:-32}
STEPS=5; WARM=2
rm -rf /tmp/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python3 bench.py --mode train --batch $B --steps $STEPS --warmup $WARM --no-other-configs --no-graph-replay > gpurun_out/train_prof.log 2>&1
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
mkdir -p gpurun_out/profiles
cp $f gpurun_out/profiles/train_kernel_stats.csv
python3 - <<PY
import csv
n=$STEPS+$WARM+min($STEPS,5)      # warm-up + timed + the steps on the rank's own clock
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/n/1e6:8.3f} ms/step {float(r["Percentage"]):6.2f}% calls/step {int(r["Calls"])/n:6.1f}  {r["Name"][:100]}')
print("total kernel ms/step", tot/n/1e6)
PY
tail -1 gpurun_out/train_prof.log | cut -c1-200
