#!/bin/bash
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4 2>&1 | tail -3 | tee gpurun_out/r3_w4_checks.txt
BENCH_KS=3 timeout 300 python tools/bench_conv.py ${1:-32} fwd 2>&1 | grep "F(4,3)" | tee gpurun_out/r3_w4_bench.txt
bash tools/pmc_wino4.sh 2>&1 | grep -v "^W2026\|^done" | tee gpurun_out/r3_w4_pmc.txt
