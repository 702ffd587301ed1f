#!/bin/bash
# One parametrised measurement script for the GPU box (replaces the per-round tools/rN_*.sh scratch scripts).
#   tools/profile.sh <out-tag> <job> [<job> ...]
# jobs (run in the order given; results land under gpurun_out/<out-tag>/):
#   tests            full `pytest -m gpu` (tail in tests.txt)
#   tests:<expr>     `pytest -m gpu -k <expr>`
#   smoke            __graft_entry__.smoke()
#   bench            the driver's bench invocation (--gpus 1 --steps 20 --warmup 5) -> bench_default_invocation.json
#   bench:<name>:<bench.py flags>     any other bench line -> bench_<name>.json
#   stats:<name>:<bench.py flags>     rocprofv3 --kernel-trace --stats of bench.py <flags> -> kernel_stats_<name>.csv
#   pmc              HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) + matrix-pipe busy of tools/pmc_step.py
#   pmc:<name>:<pmc_step.py flags>    the same for another workload
#   ab:<old.so>:<bench.py flags>      A/B of the in-tree library against another build inside ONE call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
tag="$1"; shift
O="gpurun_out/$tag"; mkdir -p "$O"
# provenance stamp of every PMC summary: sha256 over the kernel sources (sivae_hip.lib.sha256)
sha=$(python -c "import sys; sys.path.insert(0, 'soft-intro-vae-pytorch_amd'); from sivae_hip import lib; print(lib.sha256()[:16])")
echo "csrc sha256 $sha" | tee "$O/lib.txt"
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value", d["value"], "untimed", d.get("value_untimed"), "ms", d["ms_per_step"], "dominant",
      {k: r.get(k) for k in ("kernel", "frac", "avg_ms", "launches")})
for k, v in d.get("also", {}).items():
    print("  ", k, v["value"], v["ms_per_step"], {a: b for a, b in v.items() if a.startswith("rate")})
PY
}
for job in "$@"; do
  kind="${job%%:*}"; rest="${job#*:}"; [ "$rest" == "$job" ] && rest=""
  name="${rest%%:*}"; flags="${rest#*:}"; [ "$flags" == "$rest" ] && flags=""
  case "$kind" in
    tests)
      if [ -n "$rest" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$rest" 2>&1 | tail -15 | tee "$O/tests_k.txt"
      else timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$O/tests.txt"; fi ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$O/smoke.txt" ;;
    bench)
      if [ -z "$name" ]; then
        SECONDS=0
        timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_default_invocation.json" 2> "$O/bench.err"
        echo "bench wall ${SECONDS}s"; line "$O/bench_default_invocation.json"
      else
        timeout 600 python bench.py $flags > "$O/bench_$name.json" 2> "$O/bench_$name.err"; echo "[$name]"; line "$O/bench_$name.json"
      fi ;;
    stats)
      timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/st_$name" -- \
        python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-also $flags > "$O/prof_$name.log" 2>&1
      find "$O/st_$name" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats_$name.csv" \; ; rm -rf "$O/st_$name"
      echo "[stats $name]"; tail -1 "$O/prof_$name.log" | cut -c1-140 ;;
    pmc)
      n="${name:-celeb256_bs128_fp32}"
      timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_f" -- python tools/pmc_step.py $flags > "$O/pmc_f.log" 2>&1
      timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_w" -- python tools/pmc_step.py $flags > "$O/pmc_w.log" 2>&1
      python tools/pmc_traffic.py "$O/pmc_f" "$O/pmc_w" --lib-sha "$sha" > "$O/pmc_traffic_$n.json"; rm -rf "$O/pmc_f" "$O/pmc_w"
      timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 \
        --output-format csv -d "$O/pmc_m" -- python tools/pmc_step.py $flags > "$O/pmc_m.log" 2>&1
      python tools/pmc_mfma_busy.py "$O/pmc_m" --lib-sha "$sha" > "$O/pmc_mfma_busy_$n.json"; rm -rf "$O/pmc_m"
      python - "$O" "$n" <<'PY'
import json, sys
O, n = sys.argv[1:3]
d = json.load(open(f"{O}/pmc_traffic_{n}.json")); print("traffic GB", d["step_total_hbm_bytes"] / 1e9, d["calibration"].get("read_scale"))
d = json.load(open(f"{O}/pmc_mfma_busy_{n}.json")); print("busy", d["whole_step_mfma_busy_frac"])
for k, v in list(d["kernels"].items())[:10]: print(" ", k[:60], v["mfma_busy_frac"], v["share_of_gpu_active"])
PY
      ;;
    ab)
      old="$name"
      for rep in 1 2; do
        for which in new old; do
          if [ $which == old ]; then export SIVAE_LIB="$old" SIVAE_LIB_ALLOW_MISSING=1; else unset SIVAE_LIB SIVAE_LIB_ALLOW_MISSING; fi
          timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-also $flags 2>/dev/null | tail -1 | \
            python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', d['value'], d['ms_per_step'])" | tee -a "$O/ab.txt"
        done
      done; unset SIVAE_LIB SIVAE_LIB_ALLOW_MISSING ;;
    *) echo "unknown job $job"; exit 2 ;;
  esac
done
