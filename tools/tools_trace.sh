#!/bin/bash
# kernel timeline of the last steps of a short bench run (gaps between consecutive kernels)
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --kernel-iters 2 "$@" > /tmp/tr.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/tr.log
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:36]))
for f in glob.glob("/tmp/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),"MEMCPY "+r.get('Direction','')[:20]))
rows.sort()
# find the timed region: the last 8*4 layer/classifier kernels before the kernel-iters loop
idx=[i for i,r in enumerate(rows) if 'k_classifier' in r[2]]
lo=idx[-4] if len(idx)>=4 else 0
prev=None
for s,e,n in rows[lo-2: lo+40]:
    print(f"{n:38s} dur_us={(e-s)/1e3:8.1f} gap_us={((s-prev)/1e3) if prev else 0:8.1f}")
    prev=e
PY
