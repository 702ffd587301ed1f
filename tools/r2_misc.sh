cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2h
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_bf16_gpu.py tests/test_syncbn_gpu.py -x -q > gpurun_out/r2h/t.log 2>&1; tail -5 gpurun_out/r2h/t.log
python bench.py --config celeb128 --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*'
