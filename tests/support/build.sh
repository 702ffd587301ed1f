#!/bin/bash
# Builds the test-support kernels (gfx950) into tests/support/libsivae_testsupport.so.  Separate from the product library on
# purpose: nothing under tests/ is exported by libsivae_hip.so.
set -e
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/libsivae_testsupport.so"
if [ ! -f "$out" ] || [ "$here/squatter.hip" -nt "$out" ]; then
  "${HIPCC:-/opt/rocm/bin/hipcc}" --offload-arch=gfx950 -O2 -std=c++17 -fPIC -fno-gpu-rdc -shared "$here/squatter.hip" -o "$out"
fi
echo "built $out"
