#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests/test_e2e_gpu.py tests/test_segments_gpu.py -x -q 2>&1 | tail -3
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'])"
python bench.py --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs16', d['value'], d['ms_per_step'])"
