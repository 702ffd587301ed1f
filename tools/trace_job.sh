cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; O=gpurun_out/trace; mkdir -p $O
for cfg in "bs16:--config celeb256 --global-batch 16" "bs128:--config celeb256" "boot8:--config celeb256 --bootstrap --global-batch 8" "boot64:--config celeb256 --bootstrap --global-batch 64"; do
  n="${cfg%%:*}"; f="${cfg#*:}"
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-also $f > $O/tr_$n.log 2>&1
  python tools/trace_calls.py /tmp/tr_$n > $O/calls_$n.csv; wc -l $O/calls_$n.csv; tail -1 $O/tr_$n.log | cut -c1-100
done
