cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "splitk or fixture or conv1x1" 2>&1 | tail -12
for cfg in "bs16:--config celeb256 --global-batch 16" "boot8:--config celeb256 --bootstrap --global-batch 8"; do
  n="${cfg%%:*}"; f="${cfg#*:}"
  for rep in 1 2; do
   for fold in 1 0; do
    SIVAE_SPLITK_BN_FOLD=$fold timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-also $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n fold=$fold', d['value'], d['ms_per_step'])" | tee -a $O/ab5.txt
   done
  done
done
