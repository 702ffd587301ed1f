cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "wino_splitk or wino4_splitk or fixture" 2>&1 | tail -3
for cfg in "bs16:--config celeb256 --global-batch 16" "boot8:--config celeb256 --bootstrap --global-batch 8" "bs128:--config celeb256 --steps 10" "boot64:--config celeb256 --bootstrap --global-batch 64 --steps 10"; do
  n="${cfg%%:*}"; f="${cfg#*:}"
  for rep in 1 2; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-also $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n new', d['value'], d['ms_per_step'])" | tee -a $O/ab4.txt
  done
done
