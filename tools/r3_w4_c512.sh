#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for m in 256 512; do echo "== MAXC=$m B=128"; SIVAE_WINO4_MAXC=$m BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py 128 fwd 2>&1 | grep "k3" | cut -c1-150; done
