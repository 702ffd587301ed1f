#!/bin/bash
# round 3, first GPU pass of the segmented (paired-pass) engine: its own tests, the end-to-end parity suite (the
# fixtures have B = 8: "auto" pairs them), then shard-size benches with the pairs off / on inside ONE call.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_segments_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r3_seg_tests.txt
python -m pytest tests/test_e2e_gpu.py tests/test_train_loop_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -8 >> gpurun_out/r3_seg_tests.txt
cat gpurun_out/r3_seg_tests.txt
for gb in 16 128; do
  for pp in 0 1; do
    SIVAE_PAIR_PASSES=$pp python bench.py --global-batch $gb --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('celeb256 bs$gb pair=$pp', d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r3_seg_bench.txt
for gb in 8 64; do
  for pp in 0 1; do
    SIVAE_PAIR_PASSES=$pp python bench.py --bootstrap --global-batch $gb --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bootstrap256 bs$gb pair=$pp', d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee -a gpurun_out/r3_seg_bench.txt
