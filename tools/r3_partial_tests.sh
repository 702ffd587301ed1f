#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_e2e_gpu.py tests/test_segments_gpu.py tests/test_direct_grads_gpu.py tests/test_train_loop_gpu.py -x -q 2>&1 | tail -5
