# Round-6 evidence batch (one gpurun call): smoke, the driver's bench line, kernel-stat profiles of every BASELINE workload,
# PMC passes (HBM traffic, matrix-pipe busy) of the headline and of the bf16 mode, per-call traces, per-layer times, and the
# 2-rank bench started as plain `python bench.py --gpus 2` (it spawns its own ranks).  Results: gpurun_out/<tag>/.
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0; cd $GRAFT_REPO_ROOT
tag=${1:-r6final}; O=gpurun_out/$tag; mkdir -p $O
bash tools/profile.sh $tag smoke bench "stats:celeb256_bs128_fp32:" "stats:celeb256_bs16_fp32:--global-batch 16" "stats:bootstrap256_bs8_fp32:--bootstrap --global-batch 8" "stats:celeb128_bs128_bf16:--config celeb128 --dtype bf16" "stats:cifar10_bs256_fp32:--config cifar10" pmc "pmc:celeb128_bf16:--config celeb128 --dtype bf16" 2>&1 | tail -60
for sb in "" "--sync-bn"; do
  t=$([ -z "$sb" ] && echo local_bn || echo sync_bn)
  timeout 600 python bench.py --gpus 2 --backend gloo --same-device --steps 4 --warmup 2 --scaling strong $sb > $O/bench_2rank_gloo_same_device_$t.json 2> $O/bench_2rank_$t.err
  tail -1 $O/bench_2rank_gloo_same_device_$t.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2-rank same device (self-spawned)', '$t', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config'].get('batchnorm'))"
done
timeout 600 python tools/layer_times.py --steps 3 --config celeb256 > $O/layer_times_celeb256_bs128_fp32.txt 2>&1
timeout 600 python tools/layer_times.py --steps 3 --config celeb256 --global-batch 16 > $O/layer_times_celeb256_bs16_fp32.txt 2>&1
for cfg in "bs16:--config celeb256 --global-batch 16" "bs128:--config celeb256" "boot8:--config celeb256 --bootstrap --global-batch 8" "boot64:--config celeb256 --bootstrap --global-batch 64"; do
  n="${cfg%%:*}"; f="${cfg#*:}"
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-also $f > $O/tr_$n.log 2>&1
  python tools/trace_calls.py /tmp/tr_$n > $O/calls_$n.csv; wc -l $O/calls_$n.csv
done
