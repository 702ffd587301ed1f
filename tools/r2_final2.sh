# after the split-K up-dgrad: full GPU suite, smoke, headline + shard bench records
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2y; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 6 --warmup 2 --cpu-iters 2 > $O/bench_celeb256_bs128_fp32.json 2>/dev/null; cut -c1-200 $O/bench_celeb256_bs128_fp32.json
python bench.py --global-batch 16 --steps 12 --warmup 3 --no-cpu-baseline > $O/bench_celeb256_bs16_fp32.json 2>/dev/null; cut -c1-200 $O/bench_celeb256_bs16_fp32.json
python bench.py --global-batch 16 --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-160
python bench.py --bootstrap --global-batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-160
