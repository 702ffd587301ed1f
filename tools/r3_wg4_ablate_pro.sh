#!/bin/bash
# where the fused-prologue variant of wino4_wgrad_kernel loses its ~20 %: 8 = no prologue store, 16 = no prologue at all
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
cp $L /tmp/new.so
for v in 0 8 16 0; do
if [ $v = 0 ]; then cp /tmp/new.so $L; else cp tools/_build/libsivae_wg$v.so $L; fi
echo "== ablate $v"; BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 300 python tools/bench_conv.py 128 wgrad 2>&1 | grep "k3" | cut -c1-20,28-38 | tr '\n' ' '; echo
done
cp /tmp/new.so $L
