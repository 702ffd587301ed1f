# round-2 profile A: bs16 kernel stats (the per-GPU shard of config 4) + MFMA-busy counters at bs128
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
python bench.py --steps 10 --warmup 3 --global-batch 16 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2a/bench_b16.json 2> gpurun_out/r2a/bench_b16.err
tail -c 600 gpurun_out/r2a/bench_b16.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a/stats_b16 -- python bench.py --steps 5 --warmup 2 --global-batch 16 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2a/prof_b16.log 2>&1
find gpurun_out/r2a/stats_b16 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2a/kernel_stats_b16.csv \;
find gpurun_out/r2a/stats_b16 -name "*kernel_trace.csv" -delete
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/r2a/pmc_mfma -- python tools/pmc_step.py > gpurun_out/r2a/pmc_mfma.log 2>&1
tail -3 gpurun_out/r2a/pmc_mfma.log | cut -c1-300
python tools/pmc_mfma_busy.py gpurun_out/r2a/pmc_mfma > gpurun_out/r2a/pmc_mfma_busy.json
rm -rf gpurun_out/r2a/pmc_mfma gpurun_out/r2a/stats_b16
head -c 3000 gpurun_out/r2a/pmc_mfma_busy.json
