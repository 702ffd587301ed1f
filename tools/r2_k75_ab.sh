cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python tests/kernel_checks.py conv5_k75 2>&1 | tail -24 | cut -c1-130
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x 2>&1 | tail -2
for k in 1 0 1 0; do echo "== K75=$k"; SIVAE_CONV5_K75=$k python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | cut -c1-130; done
SIVAE_CONV5_K75=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); pk=d['roofline']['per_kernel']
for k,v in pk.items():
    if 'k75' in k or 'conv_fwd_kernel<5' in k: print(k, v)
"
