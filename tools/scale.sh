#!/bin/bash
# North-star scaling table (BASELINE.json: images/sec at 1/2/4/8 GPUs, absolute, x over 1 GPU, fraction of the conv
# roofline): runs bench.py --gpus N for N in {1,2,4,8} (those the node has) one process per GPU over RCCL and prints
# the table.  Usage: tools/scale.sh [--steps K] [--warmup W] [extra bench.py flags, e.g. --bootstrap --global-batch 64]
# Precedent for the launch shape: style_soft_intro_vae/launcher.py:102-129 (mp.spawn, one process per GPU, NCCL).
set -e
cd "$(dirname "${BASH_SOURCE[0]}")/.."
STEPS=20; WARM=5; EXTRA=()
while [ $# -gt 0 ]; do
  case "$1" in
    --steps) STEPS=$2; shift 2;;
    --warmup) WARM=$2; shift 2;;
    *) EXTRA+=("$1"); shift;;
  esac
done
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
OUT=${SCALE_OUT:-gpurun_out/scale}; mkdir -p "$OUT"
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-also "${EXTRA[@]}" > "$OUT/n$N.json" 2> "$OUT/n$N.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARM "${EXTRA[@]}" > "$OUT/n$N.json" 2> "$OUT/n$N.err"
  fi
done
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
rows = []
for n in (1, 2, 4, 8):
    p = os.path.join(out, "n%d.json" % n)
    if os.path.exists(p) and os.path.getsize(p):
        rows.append((n, json.loads(open(p).read().strip().splitlines()[-1])))
if not rows:
    sys.exit("no bench output")
base = {sc: None for sc in ("strong", "weak")}
print("%-5s %-7s %10s %8s %10s %12s %14s" % ("GPUs", "scaling", "img/s", "x 1GPU", "ms/step", "img/GPU", "conv-roofline"))
for n, d in rows:
    legs = [(d["scaling"], d["value"], d["ms_per_step"], d["config"]["per_gpu_batch"],
             d["roofline"]["step"].get("mfma_issued_frac"), d["roofline"]["step"]["algorithmic_frac"])]
    if "weak" in d and isinstance(d["weak"], dict):
        w = d["weak"]
        legs.append(("weak", w["value"], w["ms_per_step"], w["per_gpu_batch"], None,
                     w["algorithmic_tflops_per_gpu"] / d["roofline"]["peak"]))
    for sc, v, ms, per, issued, alg in legs:
        if n == 1:
            base["strong"] = base["weak"] = v
        x = v / base[sc] if base.get(sc) else float("nan")
        roof = ("issued %.3f / alg %.3f" % (issued, alg)) if issued is not None else ("alg %.3f" % alg)
        print("%-5d %-7s %10.1f %8.2f %10.2f %12d %s" % (n, sc, v, x, ms, per, roof))
PY
