#!/bin/bash
# Round profile: kernel-trace stats of the default bench command + separate PMC passes (HBM traffic counters).
# usage (on the GPU box, from the repo root):  tools/profile_round.sh r01
# Writes gpurun_out/profiles/<tag>_*.{csv,json,txt}; copy what should be judged into profiles/.
TAG=${1:-r01}
OUT=$PWD/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --repeats 0"
cd $PWD
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- python3 $CMD > $OUT/${TAG}_trace_run.log 2>&1
cp $(find /tmp/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" GRBM_GUI_ACTIVE; do
  n=$(echo $pmc | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/prof_$TAG/pmc_$n -- python3 $CMD > $OUT/${TAG}_pmc_${n}_run.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob("/tmp/prof_$TAG/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        # "k_gcn_layer": the layer kernel proper (k_gcn_layer_ps<false> in the chained step, k_gcn_layer<AGG> otherwise);
        # the last layer with the fused classifier heads is reported on its own
        name = "k_gcn_layer_ps_cls" if "k_gcn_layer_ps<true" in k else "k_gcn_layer" if "k_gcn_layer" in k else "k_classifier" if "k_classifier" in k else None
        if not name: continue
        a=agg[name][row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
out={}
for name,cs in agg.items():
    r={c: v/n for c,(v,n) in cs.items()}
    r["launches_averaged"]={c: n for c,(v,n) in cs.items()}
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of a wide coalesced read (MI355X_MICROARCH.md §HBM):
        # calibrated here on a plain copy (144 MB reported for 295 MB read) and on the plain-linear kernel (153 MB).
        r["hbm_read_bytes_per_launch"]=r["FETCH_SIZE"]*1024*2
        r["hbm_write_bytes_per_launch"]=r["WRITE_SIZE"]*1024
        r["hbm_bytes_per_launch"]=r["hbm_read_bytes_per_launch"]+r["hbm_write_bytes_per_launch"]
    out[name]=r
out["command"]="rocprofv3 --pmc <counter set> -- python3 $CMD   (one pass per counter set)"
import sys; sys.path.insert(0, ".")
import bench
out["kernel_source_digest"]=bench.kernel_source_digest()   # bench.py only trusts a summary measured on the current kernel sources
json.dump(out, open("$OUT/${TAG}_pmc.json","w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
head -8 $OUT/${TAG}_kernel_stats.csv | cut -c1-220
