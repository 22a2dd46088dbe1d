#!/bin/bash
# kernel trace of a short bench run: per-kernel durations as they occur inside real steps
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --kernel-iters 3 > /tmp/tr.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/tr/**/*kernel_stats.csv", recursive=True):
    for i,row in enumerate(csv.reader(open(f))):
        if i<6: print(",".join(x[:60] for x in row[:7]))
PY
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    prev=None; n=0
    for r in rows:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        name=r['Kernel_Name'][:32]
        if n>40 and n<75: print(f"{name:34s} dur_us={(e-s)/1e3:8.1f} gap_us={((s-prev)/1e3) if prev else 0:8.1f}")
        prev=e; n+=1
PY
