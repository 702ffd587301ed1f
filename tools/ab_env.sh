#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (boxes differ by a few %):  tools/ab_env.sh VAR "<bench.py flags>" [reps]
#   runs bench.py with VAR=0 and VAR=1 alternately and prints img/s + ms per iteration
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
var="$1"; flags="$2"; reps="${3:-2}"
for r in $(seq 1 $reps); do
  for v in 0 1; do
    env $var=$v timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-also $flags 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v  [$flags]', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
