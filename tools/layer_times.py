"""Per-layer attribution of an iteration (GPU box): every C-ABI launch is bracketed by HIP events and keyed by
(entry point, its integer arguments) — i.e. by layer shape.  Prints ms per iteration per key, largest first.
usage: python tools/layer_times.py [--filter wino4] [--steps 3] [bench.py flags ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
import torch
from sivae_hip import lib

flt = None
steps = 3
argv = sys.argv[1:]
if "--filter" in argv:
    i = argv.index("--filter"); flt = argv[i + 1]; del argv[i:i + 2]
if "--steps" in argv:
    i = argv.index("--steps"); steps = int(argv[i + 1]); del argv[i:i + 2]

REC = []
ON = [False]
_call = lib.call


def call(name, *args):
    if not ON[0] or (flt and flt not in name):
        return _call(name, *args)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    rc = _call(name, *args)
    e.record()
    REC.append((name, tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 20)), s, e))
    return rc


lib.call = call
import bench

_step = None
from sivae_hip import engine as _eng
_orig = _eng.SoftIntroEngine.soft_intro_step
count = [0]


def step(self, *a, **k):
    count[0] += 1
    ON[0] = count[0] > WARM
    return _orig(self, *a, **k)


WARM = 2
_eng.SoftIntroEngine.soft_intro_step = step
sys.argv = ["bench.py", "--steps", str(steps), "--warmup", str(WARM), "--no-cpu-baseline", "--no-kernel-timing", "--no-also"] + argv
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
torch.cuda.synchronize()
agg = {}
for name, ints, s, e in REC:
    d = agg.setdefault((name, ints), [0, 0.0])
    d[0] += 1; d[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values()) / steps
print("bracketed launches: %.1f ms per iteration" % tot)
for (name, ints), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%7.2f ms/it %5.1f x %7.3f ms  %s %s" % (ms / steps, n / steps, ms / n, name.replace("sivae_", ""), " ".join(map(str, ints))))
