#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4 2>&1 | grep -v "^ok" | tail -12
echo "== PRO bench B=32"; BENCH_PRO=1 BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py 32 fwd 2>&1 | grep "k3" | cut -c1-150
