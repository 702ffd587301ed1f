#!/bin/bash
# per-phase timing build of the two one-launch BatchNorm backward kernels (-DB16_TIMING / -DBF_TIMING: s_memrealtime stamps of
# thread 0 at the phase boundaries; tools/b16_bn_timing.py, tools/bn_fused_timing.py) as a whole library: tools/abx/libsivae_b16timing.so
cd "$(dirname "$0")/../soft-intro-vae-pytorch_amd/csrc" || exit 1
mkdir -p ../../tools/abx /tmp/bntiming
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
/opt/rocm/bin/hipcc $F -DB16_TIMING -c bf16_bn_fused.hip -o /tmp/bntiming/bf16_bn_fused.o &
/opt/rocm/bin/hipcc $F -DBF_TIMING -c bn_fused.hip -o /tmp/bntiming/bn_fused.o &
wait
objs=$(ls build/*.o | grep -v "build/bf16_bn_fused.o" | grep -v "build/bn_fused.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abx/libsivae_b16timing.so $objs /tmp/bntiming/bf16_bn_fused.o /tmp/bntiming/bn_fused.o && echo built tools/abx/libsivae_b16timing.so
