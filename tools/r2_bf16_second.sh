cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2d
timeout 300 python tests/kernel_checks.py linear > gpurun_out/r2d/kc_linear.txt 2>&1; tail -40 gpurun_out/r2d/kc_linear.txt
timeout 300 python tests/kernel_checks16.py wgrad > gpurun_out/r2d/kc16_wgrad.txt 2>&1; tail -3 gpurun_out/r2d/kc16_wgrad.txt
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/r2d/test_bf16.log 2>&1; tail -5 gpurun_out/r2d/test_bf16.log
python bench.py --config celeb128 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2d/bench_c128_bf16.json 2>/dev/null; cut -c1-200 gpurun_out/r2d/bench_c128_bf16.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2d/stats -- python bench.py --config celeb128 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2d/prof.log 2>&1
find gpurun_out/r2d/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2d/kernel_stats_c128_bf16.csv \;
rm -rf gpurun_out/r2d/stats
python bench.py --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
