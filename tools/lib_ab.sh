#!/bin/bash
# Same-box A/B of library builds with any command:  bash tools/lib_ab.sh <tag> <reps> "<command>" <variant> [<variant> ...]   ("base" = shipped)
tag=$1; reps=$2; cmd=$3; shift 3
for r in $(seq $reps); do for v in "$@"; do
  lib=echoglad_amd/lib/libechoglad_hip.$v.so; [ "$v" == "base" ] && lib=echoglad_amd/lib/libechoglad_hip.so
  echo "== round $r  $v" | tee -a gpurun_out/${tag}_lib_ab.txt
  ECHOGLAD_LIB=$lib timeout 600 $cmd 2>/dev/null | tee -a gpurun_out/${tag}_lib_ab.txt
done; done
