#!/bin/bash
# timing-only ablation builds of the AD form of bf16_conv_kernel (results are WRONG by construction): whole libraries under tools/abx/
#   usage: tools/build_ad_ablations.sh "NOA" "NOA NOX" "NOA NOX NOB" ...   (one library per argument: ad_<flags joined by _>.so)
cd "$(dirname "$0")/../soft-intro-vae-pytorch_amd/csrc" || exit 1
mkdir -p ../../tools/abx /tmp/adabl
for v in "$@"; do
  tag=$(echo $v | tr ' ' '_'); defs=""; for f in $v; do defs="$defs -DAD_ABL_$f"; done
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $defs -c bf16_conv.hip -o /tmp/adabl/bf16_conv_$tag.o &&
    objs=$(ls build/*.o | grep -v "build/bf16_conv.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abx/ad_$tag.so $objs /tmp/adabl/bf16_conv_$tag.o && echo built $tag ) &
done
wait
