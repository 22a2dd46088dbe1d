#!/bin/bash
L=$PWD/echoglad_amd/lib
export TMPDIR=/tmp
for v in "" fb1 fb2 fb3; do
  if [ -n "$v" ]; then export ECHOGLAD_LIB=$L/libechoglad_hip.$v.so; else unset ECHOGLAD_LIB; fi
  rm -rf /tmp/prof_fb
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fb -- python3 tools/tools_heads.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f=glob.glob("/tmp/prof_fb/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_cls_first_bwd" in r["Name"] or "k_cls_mid_bwd" in r["Name"]: print("${v:-shipped}", r["Name"][:30], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
