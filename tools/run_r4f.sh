#!/bin/bash
python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "fp64" -s 2>&1 | grep -v "^$" | head -24
for i in 1 2; do python tools/tools_ring.py 2>&1 | tail -1; EG_RING_GUARD=0 python tools/tools_ring.py 2>&1 | tail -1; done
