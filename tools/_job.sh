cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; O=gpurun_out/r5final; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_invocation.json 2> $O/bench.err
python - $O/bench_default_invocation.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d.get("value_untimed"), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["mfma_busy_pmc"]["stale"], d["roofline"]["traffic_provenance"]["stale"])
print(d.get("zz_shard_summary"))
PY
( time timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) 2>&1 | tee $O/tests_full.txt
