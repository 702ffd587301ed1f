cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2d; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for d in 1 0 1 0; do
echo "== DIRECT=$d"
SIVAE_DIRECT_GRADS=$d python bench.py --global-batch 16 --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-130
done
for d in 1 0; do
echo "== DIRECT=$d"
SIVAE_DIRECT_GRADS=$d python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-130
SIVAE_DIRECT_GRADS=$d python bench.py --config celeb128 --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-130
done
