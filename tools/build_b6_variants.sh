#!/bin/bash
# Instrumented / ablated builds of conv_wino4_b6.hip as whole libraries under tools/abx/ (they travel with gpurun):
#   tools/build_b6_variants.sh timing abl1 abl2 ...   (ablN = -DB6_ABLATE=N; results of ablated builds are wrong by design)
set -e
cd "$(dirname "$0")/.."
src=soft-intro-vae-pytorch_amd/csrc; obj=$src/build; mkdir -p tools/abx tools/_build
bash $src/build.sh > /dev/null
objs=$(ls $obj/*.o | grep -v conv_wino4_b6.o)
for v in "$@"; do
  case "$v" in
    timing) def="-DB6_TIMING";;
    abl*) def="-DB6_ABLATE=${v#abl}";;
    *) echo "unknown variant $v"; exit 2;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize $def -c $src/conv_wino4_b6.hip -o tools/_build/b6_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/abx/b6_$v.so $objs tools/_build/b6_$v.o
  echo "built tools/abx/b6_$v.so"
done
