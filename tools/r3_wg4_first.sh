#!/bin/bash
# first GPU contact of the F(4x4,3x3) weight-gradient kernel: checks, then per-layer timing against F(2x2,3x3)
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4_wgrad 2>&1 | tail -25
BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 32 wgrad 2>&1 | tail -12
BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 32 wgrad 2>&1 | tail -11
