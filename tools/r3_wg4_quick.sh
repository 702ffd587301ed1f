#!/bin/bash
# quick A/B of the F(4x4,3x3) weight gradient: checks + per-layer timing (plain at B=32, prologue at B=128)
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4_wgrad 2>&1 | grep -v "^ok" | tail -3
BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 32 wgrad 2>&1 | tail -10 | cut -c1-75
BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 128 wgrad 2>&1 | tail -10 | cut -c1-75
