#!/bin/bash
# kernel durations of the coordinate update's launches (rocprofv3 --kernel-trace --stats of tools/coord_kernel_time.py) per library variant
#   gpurun -- 'bash tools/coord_kernel_prof.sh <tag> base cm1 ...'
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=echoglad_amd/lib/libechoglad_hip.$v.so; [ "$v" == "base" ] && lib=echoglad_amd/lib/libechoglad_hip.so
  rm -rf /tmp/ckp_$v
  export ECHOGLAD_LIB=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ckp_$v -- python3 tools/coord_kernel_time.py > /dev/null 2>&1
  f=$(find /tmp/ckp_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v" | tee -a gpurun_out/${tag}_coord_prof.txt
  grep -E "k_coord|k_bilinear" $f | awk -F, '{printf "%-60s calls %s avg_ns %s min_ns %s\n", substr($1,1,60), $2, $4, $6}' | tee -a gpurun_out/${tag}_coord_prof.txt
done
