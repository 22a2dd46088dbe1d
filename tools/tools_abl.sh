#!/bin/bash
for v in "" nonbr nomfma nop3 only_mfma u4; do
  if [ -n "$v" ]; then export ECHOGLAD_LIB=$PWD/echoglad_amd/lib/libechoglad_hip.$v.so; else unset ECHOGLAD_LIB; fi
  r=$(timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|\"avg_launch_ms\": [0-9.]*" | tr "\n" " ")
  echo "variant=${v:-default} $r"
done
