#!/bin/bash
L=$PWD/echoglad_amd/lib
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['repeats']['ms_per_step']['median'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for i in 1 2; do
run new
ECHOGLAD_LIB=$L/libechoglad_hip.oldboth.so run oldboth
ECHOGLAD_LIB=$L/libechoglad_hip.oldkout.so run oldkout
ECHOGLAD_LIB=$L/libechoglad_hip.olddiv.so run olddiv
done
