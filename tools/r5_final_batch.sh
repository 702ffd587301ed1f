export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile.sh r5final smoke bench "stats:celeb256_bs128_fp32:" "stats:celeb256_bs16_fp32:--global-batch 16" "stats:bootstrap256_bs8_fp32:--bootstrap --global-batch 8" "stats:celeb128_bs128_bf16:--config celeb128 --dtype bf16" "stats:cifar10_bs256_fp32:--config cifar10" pmc "pmc:celeb128_bf16:--config celeb128 --dtype bf16" 2>&1 | tail -60
O=gpurun_out/r5final
for sb in "" "--sync-bn"; do
  tag=$([ -z "$sb" ] && echo local || echo syncbn)
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --same-device --steps 4 --warmup 2 --scaling strong $sb > $O/bench_2rank_gloo_same_device_$tag.json 2> $O/bench_2rank_$tag.err
  tail -1 $O/bench_2rank_gloo_same_device_$tag.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2-rank same device', '$tag', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config'].get('batchnorm'))"
done
