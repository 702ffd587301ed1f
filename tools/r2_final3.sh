# final records of the round-2 build: full GPU suite, smoke, headline bench (kernel-timed, CPU baseline), kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2x; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 6 --warmup 2 --cpu-iters 2 > $O/bench_celeb256_bs128_fp32.json 2>/dev/null; cut -c1-200 $O/bench_celeb256_bs128_fp32.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-160
python bench.py --global-batch 16 --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-160
python bench.py --config celeb128 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_celeb128_bs128_fp32.json 2>/dev/null; cut -c1-160 $O/bench_celeb128_bs128_fp32.json
python bench.py --config cifar10 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cifar10_bs256_fp32.json 2>/dev/null; cut -c1-160 $O/bench_cifar10_bs256_fp32.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/prof.log 2>&1
find $O/st -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_celeb256_bs128_fp32.csv \; ; rm -rf $O/st
head -5 $O/rocprofv3_kernel_stats_celeb256_bs128_fp32.csv | cut -c1-120
