#!/bin/bash
# split-K form of conv_wino4: checks, shard-size layer timing (F(4,3) incl. split-K vs F(2,3) incl. its split-K), and a
# regression A/B of the plain path at batch 128 against tools/_build/libsivae_head.so
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
timeout 300 python tests/kernel_checks.py wino4 2>&1 | grep -v "^ok" | tail -6
for B in 16 8; do
echo "== B=$B plain"; BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py $B fwd 2>&1 | grep "F(4,3)" | cut -c1-20,29-40,75-110 | tr '\n' ' '; echo
echo "== B=$B prologue"; BENCH_PRO=1 BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py $B fwd 2>&1 | grep "F(4,3)" | cut -c1-20,29-40,75-110 | tr '\n' ' '; echo
done
cp $L /tmp/new.so
for which in new head new head; do
if [ $which = head ]; then cp tools/_build/libsivae_head.so $L; else cp /tmp/new.so $L; fi
echo "== $which plain B=128";  BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py 128 fwd 2>&1 | grep "k3" | cut -c1-20,28-40 | tr '\n' ' '; echo
done
cp /tmp/new.so $L
