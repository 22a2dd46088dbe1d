#!/bin/bash
# run from the repo root on the GPU box
shopt -s globstar
export TMPDIR=/tmp
run() { # tag, env...
  tag=$1; shift
  for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $pmc | tr ' ' '_')
    env "$@" rocprofv3 --pmc $pmc --output-format csv -d /tmp/p2_${tag}_$n -- python3 tools/tools_micro.py > /tmp/p2_${tag}_$n.log 2>&1
    if ! ls /tmp/p2_${tag}_$n/**/*counter_collection.csv /tmp/p2_${tag}_$n/*/*counter_collection.csv >/dev/null 2>&1; then
      echo "tools_pmc2: no counter_collection.csv for $tag / $pmc (run from the repo root); log:" >&2; tail -5 /tmp/p2_${tag}_$n.log >&2; exit 1
    fi
  done
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/p2_${tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        k = "aggregate<2>" if "k_aggregateILi2" in k or "k_aggregate<2>" in k else "aggregate<1>" if "k_aggregate" in k else "layer<0>" if "k_gcn_layer<0>" in k else "layer<1>" if "k_gcn_layer<1>" in k else "layer<2>" if "k_gcn_layer<2>" in k else "copy" if "elementwise" in k or "copy" in k.lower() else None
        if k is None: continue
        grid=row.get("Grid_Size","")
        agg[(k,grid,row["Counter_Name"])][0]+=float(row["Counter_Value"]); agg[(k,grid,row["Counter_Name"])][1]+=1
print("== $tag")
for k,(v,c) in sorted(agg.items()):
    print(k, f"avg={v/c:.4g} n={c}")
PY
}
run g768 EG_GRID=768
run g256 EG_GRID=256
