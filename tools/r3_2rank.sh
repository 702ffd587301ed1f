#!/bin/bash
# the N = 2 launch shape of the driver on ONE GPU (both ranks on cuda:0, gloo): exercises the data-parallel path of the
# final build — global batch 128 sharded 64 + 64 with pairing, two flat-gradient all-reduces per iteration
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p gpurun_out/r3f
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
  bench.py --gpus 2 --steps 4 --warmup 2 --same-device --backend gloo --scaling strong > gpurun_out/r3f/bench_2rank_gloo_same_device.json 2> gpurun_out/r3f/bench_2rank.err
tail -c 1500 gpurun_out/r3f/bench_2rank_gloo_same_device.json | cut -c1-600; tail -3 gpurun_out/r3f/bench_2rank.err | cut -c1-300
