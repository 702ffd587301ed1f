cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2k
timeout 600 python tests/kernel_checks16.py > gpurun_out/r2k/kc16.txt 2>&1; tail -1 gpurun_out/r2k/kc16.txt; grep -E "FAIL|EXC" -A6 gpurun_out/r2k/kc16.txt | head -60
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/r2k/test_bf16.log 2>&1; tail -5 gpurun_out/r2k/test_bf16.log
for cfg in celeb128 celeb256; do
for kp in 1 0; do
echo "== $cfg KWPACK=$kp"
SIVAE_BF16_KWPACK=$kp python bench.py --config $cfg --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>gpurun_out/r2k/err_${cfg}_$kp.log | cut -c1-160; tail -3 gpurun_out/r2k/err_${cfg}_$kp.log | cut -c1-300
done; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2k/stats -- python bench.py --config celeb256 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2k/prof.log 2>&1
find gpurun_out/r2k/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2k/kernel_stats_c256_bf16.csv \;
rm -rf gpurun_out/r2k/stats
head -30 gpurun_out/r2k/kernel_stats_c256_bf16.csv | cut -c1-150
