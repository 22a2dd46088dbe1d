#!/bin/bash
# sweep walk modes / grid sizes; print layer-kernel time per setting
export TMPDIR=/tmp
for mode in 0 1 2; do for grid in 512 768 1024; do
  export EG_WALK_MODE=$mode EG_GRID=$grid
  r=$(timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|\"avg_launch_ms\": [0-9.]*" | tr "\n" " ")
  echo "mode=$mode grid=$grid $r"
done; done
for mode in 0 1 2; do
  export EG_WALK_MODE=$mode EG_GRID=768
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 3 > /tmp/pmc_$mode.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/pmc_$mode/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gcn_layer" in row["Kernel_Name"]:
            agg[row["Counter_Name"]][0]+=float(row["Counter_Value"]); agg[row["Counter_Name"]][1]+=1
print("mode=$mode", {k: round(v/c) for k,(v,c) in agg.items()})
PY
done
