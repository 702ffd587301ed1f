#!/bin/bash
# Builds the instrumented library tools/w4_timing.py loads: conv_wino4.hip with -DW4_TIMING (s_memrealtime phase stamps of
# wave 0 per work item) and optional timing ablations, linked with the product objects of csrc/build/.
#   tools/build_w4_timing.sh [ablate-bits ...]     e.g.  tools/build_w4_timing.sh 0 1 2 16
# ablation bits (results are WRONG with any set): 1 no halo loads, 2 no U refills, 8 no chunk barriers, 16 no output stores
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/soft-intro-vae-pytorch_amd/csrc"
bash "$src/build.sh" > /dev/null
mkdir -p "$root/tools/abx"
objs=$(ls "$src"/build/*.o | grep -v "/conv_wino4.o")
for v in "${@:-0}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -DW4_TIMING -DW4_ABLATE="$v" \
    -c "$src/conv_wino4.hip" -o "/tmp/w4_timing_$v.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/abx/w4_timing_$v.so" $objs "/tmp/w4_timing_$v.o"
  echo "built tools/abx/w4_timing_$v.so   (W4_VARIANT=$v python tools/w4_timing.py 128)"
done
