#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tests/kernel_checks.py wino4_wgrad 2>&1 | grep -v "^ok" | tail -6
BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 32 wgrad 2>&1 | tail -10
BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 128 wgrad 2>&1 | tail -10
BENCH_PRO=1 BENCH_WINO_ONLY=1 BENCH_KS=3 timeout 200 python tools/bench_conv.py 8 wgrad 2>&1 | tail -10
for v in 0 1; do SIVAE_WINO4_WGRAD=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WINO4_WGRAD=$v', d['value'], d['ms_per_step'])"; done
for v in 0 1; do SIVAE_WINO4_WGRAD=$v python bench.py --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs16 WINO4_WGRAD=$v', d['value'], d['ms_per_step'])"; done
