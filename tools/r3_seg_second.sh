#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_segments_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r3_seg2_tests.txt
python -m pytest tests/test_kernels_gpu.py -x -q -k "bn or BN" 2>&1 | tail -4 >> gpurun_out/r3_seg2_tests.txt
cat gpurun_out/r3_seg2_tests.txt
run() { # label, flags
  python bench.py "${@:2}" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"
}
{
run "celeb256 bs128" 
run "celeb256 bs16" --global-batch 16
SIVAE_BN_FUSED_FINALIZE=0 run "celeb256 bs16 nofusedfin" --global-batch 16
SIVAE_WG_SLOTS=512 run "celeb256 bs16 wgslots512" --global-batch 16
run "boot bs64" --bootstrap --global-batch 64
run "boot bs8" --bootstrap --global-batch 8
SIVAE_BN_FUSED_FINALIZE=0 run "boot bs8 nofusedfin" --bootstrap --global-batch 8
SIVAE_WG_SLOTS=512 run "boot bs8 wgslots512" --bootstrap --global-batch 8
SIVAE_WG_SLOTS=512 run "celeb256 bs128 wgslots512"
} 2>&1 | tee gpurun_out/r3_seg2_bench.txt
