#!/bin/bash
# timing ablations of wino4_wgrad_kernel (libraries built with -DG4_ABLATE=<bits> by tools/build_wg4_ablate.sh; their
# results are wrong by design): 1 no LDS-direct loads, 2 no transform phase, 4 no MFMAs
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
cp $L /tmp/new.so
for v in 0 1 2 4 3 6 5 7 0; do
if [ $v = 0 ]; then cp /tmp/new.so $L; else cp tools/_build/libsivae_wg$v.so $L; fi
echo "== ablate $v"; SIVAE_WINO4_WGRAD=1 BENCH_KS=3 timeout 300 python tools/bench_conv.py ${1:-32} wgrad 2>&1 | grep "k3" | cut -c1-20,28-38 | grep -v variant | tr '\n' ' '; echo
done
cp /tmp/new.so $L
