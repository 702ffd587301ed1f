#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_e2e_gpu.py tests/test_segments_gpu.py tests/test_direct_grads_gpu.py -x -q 2>&1 | tail -2
for args in "" "--global-batch 16" "--bootstrap --global-batch 8" "--bootstrap --global-batch 64"; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', d['value'], d['ms_per_step'])"
done
