#!/bin/bash
python -m pytest tests/test_gpu_diag.py -m gpu -q -k "model_" 2>&1 | grep "AssertionError: (\|passed\|failed" | head -12
