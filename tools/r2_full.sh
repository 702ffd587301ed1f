cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2f
python bench.py --config celeb128 --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r2f/bench_c128_bf16.json 2> gpurun_out/r2f/bench_c128_bf16.err; tail -3 gpurun_out/r2f/bench_c128_bf16.err; cut -c1-200 gpurun_out/r2f/bench_c128_bf16.json
timeout 1500 python -m pytest tests -q -m gpu -k "fixture or replay or bf16 or kernel16" > gpurun_out/r2f/pytest_gpu.log 2>&1; tail -8 gpurun_out/r2f/pytest_gpu.log
python bench.py --steps 5 --warmup 2 --cpu-iters 2 > gpurun_out/r2f/bench_b128.json 2> gpurun_out/r2f/bench_b128.err; cut -c1-250 gpurun_out/r2f/bench_b128.json
