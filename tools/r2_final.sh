# round-2 final pass (one gpurun call): full GPU test suite, smoke, the bench lines, bf16 profiles, 2-rank bench check
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2z; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
python bench.py --config celeb128 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_celeb128_bs128_bf16.json 2>/dev/null; cut -c1-200 $O/bench_celeb128_bs128_bf16.json
python bench.py --config celeb256 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_celeb256_bs128_bf16.json 2>/dev/null; cut -c1-200 $O/bench_celeb256_bs128_bf16.json
python bench.py --config celeb128 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
python bench.py --config celeb256 --dtype bf16 --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
python bench.py --config cifar10 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --same-device --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; cut -c1-400 $O/bench_2rank_gloo.json; tail -2 $O/bench_2rank_gloo.err | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python bench.py --config celeb128 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/prof_bf16_128.log 2>&1
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_celeb128_bs128_bf16.csv \; ; rm -rf $O/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python bench.py --config celeb256 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/prof_bf16_256.log 2>&1
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_celeb256_bs128_bf16.csv \; ; rm -rf $O/st
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python tools/pmc_step.py --config celeb128 --dtype bf16 > $O/pmc_f16.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python tools/pmc_step.py --config celeb128 --dtype bf16 > $O/pmc_w16.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic_celeb128_bs128_bf16.json; rm -rf $O/pmc_f $O/pmc_w
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -- python tools/pmc_step.py --config celeb128 --dtype bf16 > $O/pmc_m16.log 2>&1
python tools/pmc_mfma_busy.py $O/pmc_m > $O/pmc_mfma_busy_celeb128_bs128_bf16.json; rm -rf $O/pmc_m
python -c "
import json
d=json.load(open('$O/pmc_traffic_celeb128_bs128_bf16.json')); print('bf16 traffic GB', d['step_total_hbm_bytes']/1e9, d['calibration'])
d=json.load(open('$O/pmc_mfma_busy_celeb128_bs128_bf16.json')); print('bf16 busy', d['whole_step_mfma_busy_frac'], list(d['kernels'].items())[:2])
"
ls $O
