cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; O=gpurun_out/r5j; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -k "bf16 or kernels16 or bn" 2>&1 | grep -E "passed|failed|error" | tail -3 ) 2>&1 | tee $O/tests_bn_bf16.txt
for rep in 1 2; do for pf in 1 0; do
  SIVAE_BN_FUSED_PREFETCH=$pf timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-also --config celeb128 --dtype bf16 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 pf=$pf', d['value'], d['ms_per_step'])" | tee -a $O/ab_pf.txt
done; done
for pf in 1 0; do
  SIVAE_BN_FUSED_PREFETCH=$pf timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-also --config cifar10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cifar pf=$pf', d['value'], d['ms_per_step'])" | tee -a $O/ab_pf.txt
  SIVAE_BN_FUSED_PREFETCH=$pf timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-also --bootstrap --global-batch 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('boot8 pf=$pf', d['value'], d['ms_per_step'])" | tee -a $O/ab_pf.txt
done
