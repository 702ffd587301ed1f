#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_e2e_gpu.py -x -q -k "bootstrap256_full or celeb1024 or dominant_kernels" 2>&1 | grep -E "passed|failed|Error|assert|E  " | head -20
