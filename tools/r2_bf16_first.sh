cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -s > gpurun_out/r2c/test_bf16.log 2>&1; tail -60 gpurun_out/r2c/test_bf16.log
python bench.py --config celeb128 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2c/bench_c128_bf16.json 2> gpurun_out/r2c/bench_c128_bf16.err; tail -3 gpurun_out/r2c/bench_c128_bf16.err; cut -c1-400 gpurun_out/r2c/bench_c128_bf16.json
python bench.py --config celeb128 --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2c/bench_c128_fp32.json 2>/dev/null; cut -c1-300 gpurun_out/r2c/bench_c128_fp32.json
