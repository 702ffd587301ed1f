#!/bin/bash
# round-3 final records on one box: full GPU test suite, smoke(), the driver's bench invocation, kernel statistics of the
# same workload (headline + the two shards), PMC passes (HBM traffic, matrix-pipe busy).  Results under gpurun_out/r3f.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; O=gpurun_out/r3f; mkdir -p $O
if [ "${1:-all}" != "profiles" ]; then
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/tests.txt; cat $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_invocation.json 2> $O/bench.err; echo "bench wall ${SECONDS}s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f/bench_default_invocation.json').read().strip().splitlines()[-1])
print("value",d["value"],"untimed",d.get("value_untimed"),"ms",d["ms_per_step"],"roofline",{k:d["roofline"].get(k) for k in ("kernel","frac","achieved","avg_ms","launches")})
for k,v in d.get("also",{}).items():
    print(k, v["value"], v["ms_per_step"], v.get("mfma_issued_frac"), {a:b for a,b in v.items() if a.startswith("rate")}, v.get("hbm",{}).get("frac"))
print("cpu", d["cpu_baseline"])
PY
fi
# kernel statistics
for cfg in "head:" "bs16:--global-batch 16" "boot8:--bootstrap --global-batch 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$n -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-also $a > $O/prof_$n.log 2>&1
  find $O/st_$n -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$n.csv \; ; rm -rf $O/st_$n
  tail -1 $O/prof_$n.log | cut -c1-120
done
# PMC passes (separate runs; no trace domains)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python tools/pmc_step.py > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python tools/pmc_step.py > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic_celeb256_bs128_fp32.json; rm -rf $O/pmc_f $O/pmc_w
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_m -- python tools/pmc_step.py > $O/pmc_m.log 2>&1
python tools/pmc_mfma_busy.py $O/pmc_m > $O/pmc_mfma_busy_celeb256_bs128_fp32.json; rm -rf $O/pmc_m
python -c "
import json
d=json.load(open('$O/pmc_traffic_celeb256_bs128_fp32.json')); print('traffic GB', d['step_total_hbm_bytes']/1e9, d['calibration'].get('read_scale'))
d=json.load(open('$O/pmc_mfma_busy_celeb256_bs128_fp32.json')); print('busy', d['whole_step_mfma_busy_frac'])
for k,v in list(d['kernels'].items())[:8]: print(k, v['mfma_busy_frac'], v['share_of_gpu_active'])
"
