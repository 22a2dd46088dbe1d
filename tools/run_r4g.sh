#!/bin/bash
OUT=gpurun_out
./tools/_bin/micro_coissue2 > $OUT/r4g_coissue.log 2>&1; cat $OUT/r4g_coissue.log
python -m pytest tests/test_gpu_train.py tests/test_gpu_pyg_surface.py tests/test_gpu_model.py -m gpu -x -q -k "fp64 or queue_ring or two_streams or cfg2_default or cfg4_train_step" -s 2>&1 | grep -v "^$" | tail -25
for i in 1 2; do python tools/tools_ring.py 2>&1 | tail -1; EG_RING_GUARD=0 python tools/tools_ring.py 2>&1 | tail -1; done
