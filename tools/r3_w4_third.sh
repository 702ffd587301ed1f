#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for d in 0 4 8 12; do echo "== SIVAE_W4_DEBUG=$d"; SIVAE_W4_DEBUG=$d BENCH_KS=3 timeout 300 python tools/bench_conv.py 32 fwd 2>&1 | grep "F(4,3)" | cut -c1-20,63-140; done
