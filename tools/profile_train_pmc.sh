#!/bin/bash
# PMC passes over one training step (bench.py --mode train): per-kernel HBM bytes and MFMA-busy share, averaged per launch.
# usage (GPU box, repo root):  bash tools/profile_train_pmc.sh r02 [batch]
# Writes gpurun_out/profiles/<tag>_train_pmc.json; copy into profiles/ to have it judged.
TAG=${1:-r02}
B=${2:-32}
OUT=$PWD/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --mode train --batch $B --steps 5 --warmup 2 --no-other-configs --no-graph-replay"   # STEPS below = 2 warm-up + 5 timed + 5 on the rank's own clock
rm -rf /tmp/prof_train_pmc
for pmc in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" GRBM_GUI_ACTIVE; do
  n=$(echo $pmc | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d /tmp/prof_train_pmc/pmc_$n -- python3 $CMD > $OUT/${TAG}_train_pmc_${n}_run.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
total = collections.defaultdict(lambda: [0.0, 0])
STEPS = 2 + 5 + 5        # bench.py --mode train runs max(warmup, 2) + steps + min(steps, 5) steps, all of them inside the profiled process
for f in glob.glob("/tmp/prof_train_pmc/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        tot = total[row["Counter_Name"]]
        tot[0] += float(row["Counter_Value"]); tot[1] += 1          # every kernel of the run, ATen ones included
        if "eg::" not in k:
            continue
        name = re.sub(r"\(.*", "", k).replace("void ", "").replace("eg::", "")
        a = agg[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
out = {}
for name, cs in agg.items():
    r = {c: v / n for c, (v, n) in cs.items()}
    r["launches_averaged"] = max(n for _, n in cs.values())
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        # KiB; gfx950 reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section; same correction as profile_round.sh)
        r["hbm_read_bytes_per_launch"] = r["FETCH_SIZE"] * 1024 * 2
        r["hbm_write_bytes_per_launch"] = r["WRITE_SIZE"] * 1024
        r["hbm_bytes_per_launch"] = r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in r and "GRBM_GUI_ACTIVE" in r and r["GRBM_GUI_ACTIVE"] > 0:
        # MFMA-busy share of the launch: busy cycles summed over 1024 SIMDs / (GUI-active cycles of 8 XCDs / 8 * 1024)
        r["mfma_busy_frac"] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * r["GRBM_GUI_ACTIVE"] / 8.0)
    out[name] = r
keep = {k: v for k, v in out.items() if v.get("hbm_bytes_per_launch", 0) > 5e7}      # kernels that move >= 50 MB per launch
# the whole step: every counted byte of the run / the steps it ran (setup kernels, a few MB, are inside: an upper bound)
if "FETCH_SIZE" in total and "WRITE_SIZE" in total:
    keep["hbm_bytes_per_step"] = (total["FETCH_SIZE"][0] * 1024 * 2 + total["WRITE_SIZE"][0] * 1024) / STEPS
    keep["steps_profiled"] = STEPS
import sys; sys.path.insert(0, ".")
import bench
keep["kernel_source_digest"] = bench.kernel_source_digest(bench.TRAIN_KERNEL_SOURCES)
keep["command"] = "rocprofv3 --pmc <counter set> -- python3 $CMD   (one pass per counter set)"
json.dump(keep, open("$OUT/${TAG}_train_pmc.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) if isinstance(kv[1], dict) else 0):
    if isinstance(v, dict) and "launches_averaged" in v:
        print(f'{k[:60]:60s} HBM {v.get("hbm_bytes_per_launch", 0) / 1e9:6.2f} GB/launch  mfma_busy {v.get("mfma_busy_frac", float("nan")):.2f}  x{v["launches_averaged"]}')
PY
