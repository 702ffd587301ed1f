#!/bin/bash
# F(4x4,3x3) kernel: parity checks + per-layer micro-benchmark (plain and with the BatchNorm prologue) in one call
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4 2>&1 | grep -v "^ok" | tail -6
B=${1:-32}
echo "== B=$B plain"; BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py $B fwd 2>&1 | grep "F(4,3)" | cut -c1-150
echo "== B=$B prologue"; BENCH_PRO=1 BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py $B fwd 2>&1 | grep "F(4,3)" | cut -c1-150
