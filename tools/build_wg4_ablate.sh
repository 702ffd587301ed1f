#!/bin/bash
# builds tools/_build/libsivae_wg<bits>.so = the current library with conv_wino4_wgrad.hip compiled -DG4_ABLATE=<bits>
set -e
cd "$(dirname "$0")/../soft-intro-vae-pytorch_amd/csrc"
mkdir -p ../../tools/_build
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -DG4_ABLATE=$v -c conv_wino4_wgrad.hip -o /tmp/g4_ab$v.o &
done
wait
objs=$(ls build/*.o | grep -v conv_wino4_wgrad.o)
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_build/libsivae_wg$v.so $objs /tmp/g4_ab$v.o
done
ls -la ../../tools/_build/ | grep wg
