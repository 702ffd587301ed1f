#!/bin/bash
# A/B of the working-tree library against tools/_build/libsivae_old.so on the conv micro-benchmark, inside ONE call
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
timeout 300 python tests/kernel_checks.py wino4 2>&1 | grep -v "^ok" | tail -8
cp $L /tmp/new.so
for round in 1 2; do for which in new old; do
if [ $which = old ]; then cp tools/_build/libsivae_old.so $L; else cp /tmp/new.so $L; fi
echo "== $which plain";  BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py ${1:-32} fwd 2>&1 | grep "F(4,3)" | cut -c1-20,29-40 | tr '\n' ' '; echo
echo "== $which prologue"; BENCH_PRO=1 BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py ${1:-32} fwd 2>&1 | grep "F(4,3)" | cut -c1-20,29-40 | tr '\n' ' '; echo
done; done
cp /tmp/new.so $L
