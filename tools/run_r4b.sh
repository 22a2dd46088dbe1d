#!/bin/bash
OUT=gpurun_out
python tools/tools_act.py > $OUT/r4b_act.log 2>&1
for pw in 2 4 5 8; do EG_ACT_PERSIST=$pw python tools/tools_act.py >> $OUT/r4b_act.log 2>&1; done
ECHOGLAD_LIB=$PWD/echoglad_amd/lib/libechoglad_hip.actnt.so python tools/tools_act.py >> $OUT/r4b_act.log 2>&1
ECHOGLAD_LIB=$PWD/echoglad_amd/lib/libechoglad_hip.actnt.so EG_ACT_PERSIST=5 python tools/tools_act.py >> $OUT/r4b_act.log 2>&1
cat $OUT/r4b_act.log
python -m pytest tests/test_gpu_pyg_surface.py tests/test_gpu_train.py tests/test_gpu_engine.py tests/test_gpu_parallel.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -8
