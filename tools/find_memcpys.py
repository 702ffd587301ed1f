"""Which host-side calls issue the memcpy / memset / fill launches of an iteration?  (GPU box)
usage: python tools/find_memcpys.py [bench.py flags]"""
import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from sivae_hip import engine as _eng

_orig = _eng.SoftIntroEngine.soft_intro_step
count, prof = [0], [None]


def step(self, *a, **k):
    count[0] += 1
    if count[0] == 4:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as p:
            out = _orig(self, *a, **k)
            torch.cuda.synchronize()
        prof[0] = p
        return out
    return _orig(self, *a, **k)


_eng.SoftIntroEngine.soft_intro_step = step
sys.argv = ["bench.py", "--steps", "2", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timing", "--no-also"] + sys.argv[1:]
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
agg = collections.Counter()
dev = collections.Counter()
for ev in prof[0].events():
    ks = getattr(ev, "kernels", None) or []
    for k in ks:
        n = k.name
        dev[n[:50]] += 1
        if "emcpy" in n or "emset" in n or "copyBuffer" in n or "FillFunctor" in n or "elementwise" in n or "CatArray" in n:
            frames = [f for f in (ev.stack or []) if "sivae_hip" in f or "train_soft" in f or "bench.py" in f]
            where = " < ".join(f.split("/")[-1][:48] for f in frames[:2]) if frames else "(no python frame)"
            agg[(n[:40], ev.name, where, str(ev.input_shapes)[:40])] += 1
for (n, name, where, shp), c in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print("%3d  %-40s %-18s %-100s %s" % (c, n, name, where, shp))
