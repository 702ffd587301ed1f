cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2c
timeout 600 python tests/kernel_checks16.py > gpurun_out/r2c/kc16.txt 2>&1; tail -1 gpurun_out/r2c/kc16.txt; grep -E "FAIL|EXC" -A6 gpurun_out/r2c/kc16.txt | head -40
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -3
for cfg in celeb128 celeb256; do
for pp in 1 0 1 0; do
echo "== $cfg PERSIST=$pp"
SIVAE_BF16_CONV_PERSIST=$pp python bench.py --config $cfg --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-150
done; done
