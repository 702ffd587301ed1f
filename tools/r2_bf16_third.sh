cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2e
timeout 300 python tests/kernel_checks16.py wgrad bn16 > gpurun_out/r2e/kc16.txt 2>&1; tail -2 gpurun_out/r2e/kc16.txt; grep -c FAIL gpurun_out/r2e/kc16.txt
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/r2e/test_bf16.log 2>&1; tail -3 gpurun_out/r2e/test_bf16.log
python bench.py --config celeb128 --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
SIVAE_BF16_MATERIALIZE_H=0 python bench.py --config celeb128 --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/stats -- python bench.py --config celeb128 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2e/prof.log 2>&1
find gpurun_out/r2e/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2e/kernel_stats_c128_bf16.csv \;
rm -rf gpurun_out/r2e/stats
