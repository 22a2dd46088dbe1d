#!/bin/bash
# bench line + L2/fabric counters of the fused layer kernel for the current build
export TMPDIR=/tmp
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|\"avg_launch_ms\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr "\n" " "; echo
for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $pmc | tr ' ' '_')
  rocprofv3 --pmc $pmc --output-format csv -d /tmp/p3_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 3 > /tmp/p3_$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/p3_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_gcn_layer" in row["Kernel_Name"]:
            agg[row["Counter_Name"]][0]+=float(row["Counter_Value"]); agg[row["Counter_Name"]][1]+=1
r={k: v/c for k,(v,c) in agg.items()}
print({k: round(v) for k,v in r.items()})
if "FETCH_SIZE" in r: print("fabric read MB (FETCH_SIZE x2 gfx950 correction): %.0f   write MB: %.0f   L2 hit rate: %.2f" % (r["FETCH_SIZE"]*2*1024/1e6, r.get("WRITE_SIZE",0)*1024/1e6, r["TCC_HIT_sum"]/(r["TCC_HIT_sum"]+r["TCC_MISS_sum"])))
PY
