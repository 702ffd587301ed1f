#!/bin/bash
# Builds libsivae_hip.so for gfx950 in-tree (next to the Python loader). Usage: csrc/build.sh [-j N]
set -e
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../sivae_hip/libsivae_hip.so"
obj="$here/build"
mkdir -p "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc"
pids=()
for f in abi pack conv_fwd conv_wino conv_wino4 conv_wino4_b6 conv_wino4_wgrad conv_wino_up conv_wino_up_dgrad conv_wino_up_wgrad conv_wino_wgrad conv_wgrad conv5_edge conv5_k75 conv1x1_stream bn bn_fused eltwise loss optim bf16_conv bf16_wgrad bf16_bn bf16_bn_fused linear; do
  if [ ! -f "$obj/$f.o" ] || [ "$here/$f.hip" -nt "$obj/$f.o" ] || [ "$here/common.h" -nt "$obj/$f.o" ] || [ "$here/bf16_common.h" -nt "$obj/$f.o" ] || [ "$here/pack_batch.h" -nt "$obj/$f.o" ] || [ "$here/bn_fused_common.h" -nt "$obj/$f.o" ] || { [ "$f" == conv_wino4 ] && [ "$here/conv_wino4_kernel.inc" -nt "$obj/$f.o" ]; }; then
    # conv_wino4: the SLP vectorizer packs the transform slices between its MFMAs into v_pk_* ops with extra moves —
    # packed fp32 VALU next to MFMAs is an anti-lever on gfx950 (and it breaks the slice-per-MFMA interleave)
    extra=""; case "$f" in conv_wino4|conv_wino4_b6|conv_wino4_wgrad) extra="-fno-slp-vectorize";; esac
    "$HIPCC" $FLAGS $extra -c "$here/$f.hip" -o "$obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj"/abi.o "$obj"/pack.o "$obj"/conv_fwd.o "$obj"/conv_wino.o "$obj"/conv_wino4.o "$obj"/conv_wino4_b6.o "$obj"/conv_wino4_wgrad.o "$obj"/conv_wino_up.o "$obj"/conv_wino_up_dgrad.o "$obj"/conv_wino_up_wgrad.o "$obj"/conv_wino_wgrad.o "$obj"/conv_wgrad.o "$obj"/conv5_edge.o "$obj"/conv5_k75.o "$obj"/conv1x1_stream.o \
  "$obj"/bn.o "$obj"/bn_fused.o "$obj"/eltwise.o "$obj"/loss.o "$obj"/optim.o "$obj"/bf16_conv.o "$obj"/bf16_wgrad.o "$obj"/bf16_bn.o "$obj"/bf16_bn_fused.o "$obj"/linear.o
echo "built $out"
