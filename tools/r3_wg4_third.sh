#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4_wgrad 2>&1 | grep -v "^ok" | tail -6
BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 32 wgrad 2>&1 | tail -10
BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 32 wgrad 2>&1 | tail -10
