#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for sl in 256 512 1024; do echo "== SIVAE_WG4_SLOTS=$sl"; SIVAE_WG4_SLOTS=$sl BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 300 python tools/bench_conv.py 128 wgrad 2>&1 | grep "k3" | cut -c1-20,28-38 | tr '\n' ' '; echo; done
for sl in 256 512; do echo "== B=16 SIVAE_WG4_SLOTS=$sl"; SIVAE_WG4_SLOTS=$sl BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 300 python tools/bench_conv.py 16 wgrad 2>&1 | grep "k3" | cut -c1-20,28-38 | tr '\n' ' '; echo; done
