#!/bin/bash
OUT=gpurun_out
python -m pytest tests/test_gpu_diag.py tests/test_gpu_model.py tests/test_gpu_pyg_surface.py tests/test_gpu_layer.py -x -q -m gpu > $OUT/r5j_pytest.log 2>&1; tail -3 $OUT/r5j_pytest.log
python3 tools/tools_cfg_profile.py conn 50 2>&1 | tail -1
python3 tools/tools_cfg_profile.py plain 50 2>&1 | tail -1
