#!/bin/bash
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4 2>&1 | tail -40 | tee gpurun_out/r3_w4_checks.txt
BENCH_KS=3 timeout 300 python tools/bench_conv.py 32 fwd 2>&1 | tail -14 | tee gpurun_out/r3_w4_bench.txt
