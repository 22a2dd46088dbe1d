#!/bin/bash
# Where a batch-1 inference step goes: stamp build (cycles inside the tile loop of each of the step's three launch
# forms) + kernel trace of the replayed step (durations, gaps).  -> gpurun_out/<tag>_b1_breakdown.txt
tag=${1:-b1}; out=gpurun_out/${tag}_b1_breakdown.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out
for B in 1 8; do for form in kout kin+kout kin+cls; do
  echo "### stamp build, B = $B, form = $form" >> $out
  EG_STAMP_B=$B EG_STAMP_FORM=$form EG_STAMP_WARM=1 ECHOGLAD_LIB=echoglad_amd/lib/libechoglad_hip.stamp.so timeout 300 python tools/tools_stamp.py >> $out 2>&1
done; done
for B in 1 8; do
  rm -rf /tmp/trb_$B
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trb_$B -- python3 bench.py --batch $B --steps 200 --warmup 20 --repeats 0 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
  f=$(find /tmp/trb_$B -name "*kernel_trace.csv" | head -1)
  echo "### kernel trace of the replayed step, B = $B (last 60 steps: mean duration per kernel of the step, mean gap in front of it)" >> $out
  python3 - $f >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = [e for e in ev if "k_gcn_layer_ps" in e[2]]
ev = ev[-180:]
import collections
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i in range(1, len(ev)):
    key = (i % 3, "cls" if "<true" in ev[i][2] else "plain")
    dur[key].append(ev[i][1] - ev[i][0]); gap[key].append(ev[i][0] - ev[i - 1][1])
for k in sorted(dur):
    print(f"  launch {k}: duration {sum(dur[k]) / len(dur[k]) / 1e3:7.1f} us   gap in front {sum(gap[k]) / len(gap[k]) / 1e3:6.1f} us")
step = (ev[-1][1] - ev[-178][0]) / 59
print(f"  step (start of a layer-1 launch to the next one's): {step / 1e3:.1f} us")
PY
done
cat $out
