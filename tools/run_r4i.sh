#!/bin/bash
# Speed-of-light ablations of the chained layer kernel (B = 8, configs[1]) and its stamps
OUT=gpurun_out
L=$PWD/echoglad_amd/lib
{
for v in "" hot nomfma noloads hotnomfma; do
  if [ -n "$v" ]; then export ECHOGLAD_LIB=$L/libechoglad_hip.$v.so; else unset ECHOGLAD_LIB; fi
  echo "== variant ${v:-shipped}"; python tools/tools_chain.py 2>&1 | grep -v amdgpu.ids
done
for v in stamp stamp2 stamphot; do
  export ECHOGLAD_LIB=$L/libechoglad_hip.$v.so
  echo "== $v kin+kout"; EG_STAMP_FORM=kin+kout EG_STAMP_WARM=1 python tools/tools_stamp.py 2>&1 | grep -v amdgpu.ids
done
} > $OUT/r4i_sol.log 2>&1
cat $OUT/r4i_sol.log
