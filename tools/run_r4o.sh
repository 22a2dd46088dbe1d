#!/bin/bash
OUT=gpurun_out
python -m pytest tests -m gpu -x -q > $OUT/r4o_pytest.log 2>&1; tail -4 $OUT/r4o_pytest.log
python tools/tools_determinism.py 1500 > $OUT/r4o_determinism.log 2>&1; cat $OUT/r4o_determinism.log | grep -v amdgpu
for i in 1 2; do python bench.py --mode train --batch 32 --steps 10 --warmup 3 --no-other-configs 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
