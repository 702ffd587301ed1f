cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2g
timeout 600 python tests/kernel_checks.py splitk wino_fwd wino_dgrad wino_fused > gpurun_out/r2g/kc.txt 2>&1; grep -c "^ok" gpurun_out/r2g/kc.txt; grep -v "^ok" gpurun_out/r2g/kc.txt | tail -12
python bench.py --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
SIVAE_WINO_SPLITK=0 python bench.py --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2g/stats -- python bench.py --global-batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2g/prof.log 2>&1
find gpurun_out/r2g/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2g/kernel_stats_b16.csv \;
rm -rf gpurun_out/r2g/stats
