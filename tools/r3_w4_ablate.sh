#!/bin/bash
# timing ablations of conv_wino4_kernel (libraries built with -DW4_ABLATE=<bits>; their results are wrong by design):
#   1 no halo loads, 2 no U refills, 4 no output transform / stores, 8 no chunk barriers
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
L=soft-intro-vae-pytorch_amd/sivae_hip/libsivae_hip.so
cp $L /tmp/new.so
for v in 0 1 2 4 8 7 15 0; do
if [ $v = 0 ]; then cp /tmp/new.so $L; else cp tools/_build/libsivae_ab$v.so $L; fi
echo "== ablate $v"; BENCH_KS=3 BENCH_WINO_ONLY=1 timeout 300 python tools/bench_conv.py ${1:-32} fwd 2>&1 | grep "F(4,3)" | cut -c1-20,29-40 | tr '\n' ' '; echo
done
cp /tmp/new.so $L
