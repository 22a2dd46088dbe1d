#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]   — kernel trace + PMC passes, summaries into gpurun_out/prof_<tag>/
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-iters 5 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -- python3 $ARGS > $OUT/pmc_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
out="$OUT"
# kernel stats
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for i,row in enumerate(csv.reader(open(f))):
        if i<12: print(",".join(row[:8]))
# pmc: average per kernel name
for d in sorted(glob.glob(out+"/pmc_*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            k=(row.get("Kernel_Name","")[:60], row.get("Counter_Name",""))
            agg[k][0]+=float(row.get("Counter_Value",0)); agg[k][1]+=1
        print("== pmc", os.path.basename(os.path.dirname(d)))
        for (kn,cn),(v,c) in sorted(agg.items()):
            if "gcn_layer" in kn or "classifier" in kn:
                print(f"{kn:60s} {cn:28s} avg={v/c:.4g} n={c}")
PY
