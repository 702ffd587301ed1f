#!/bin/bash
# 16x16 image-pair mode of conv_wino4: checks, then A/B of the working-tree library against tools/_build/libsivae_head.so
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
timeout 300 python tests/kernel_checks.py wino4_pair wino4_pro "wino4(" 2>&1 | grep -v "^ok" | tail -12
cp $L /tmp/new.so
for which in new head new head; do
if [ $which = head ]; then cp tools/_build/libsivae_head.so $L; else cp /tmp/new.so $L; fi
echo "== $which plain B=${1:-128}";  BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py ${1:-128} fwd 2>&1 | grep "k3" | cut -c1-20,28-40 | tr '\n' ' '; echo
echo "== $which prologue"; BENCH_PRO=1 BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py ${1:-128} fwd 2>&1 | grep "k3" | cut -c1-20,28-40 | tr '\n' ' '; echo
done
cp /tmp/new.so $L
