#!/bin/bash
# full GPU test suite + the default bench line (with the `also` object) on one box
mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3_check_tests.txt; cat gpurun_out/r3_check_tests.txt
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_check_bench.json 2> gpurun_out/r3_check_bench.err
echo "bench wall ${SECONDS}s rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_check_bench.json').read().strip().splitlines()[-1])
print("value",d["value"],"untimed",d.get("value_untimed"),"ms",d["ms_per_step"],"frac",d["roofline"]["frac"])
for k,v in d.get("also",{}).items():
    print(k, v["value"], v["ms_per_step"], v["mfma_issued_frac"], {a:b for a,b in v.items() if a.startswith("rate")}, v.get("hbm",{}).get("frac"))
PY
tail -3 gpurun_out/r3_check_bench.err
