#!/bin/bash
# final pass of round 4: full GPU suite, default bench line, r04 profiles, determinism soak
OUT=gpurun_out
python -m pytest tests -x -q -m gpu > $OUT/r5g_pytest.log 2>&1; tail -2 $OUT/r5g_pytest.log
bash tools/run_r4n.sh > $OUT/r5g_profiles.log 2>&1
tail -22 $OUT/r5g_profiles.log | head -14
python3 tools/tools_determinism.py 200 > $OUT/r5g_determinism.log 2>&1; tail -7 $OUT/r5g_determinism.log
