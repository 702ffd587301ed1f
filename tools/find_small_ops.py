"""Which torch-side ops (copies, fills, adds ...) does an iteration launch, from where?  (GPU box)
usage: python tools/find_small_ops.py [bench.py flags]   e.g. --global-batch 16"""
import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from sivae_hip import engine as _eng

_orig = _eng.SoftIntroEngine.soft_intro_step
count = [0]
prof = [None]


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
for ev in prof[0].events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::reshape",
                   "aten::detach", "aten::alias", "aten::slice", "aten::select", "aten::_unsafe_view", "aten::view_as",
                   "aten::contiguous", "aten::expand", "aten::t", "aten::transpose", "aten::narrow", "aten::unsqueeze",
                   "aten::squeeze", "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::to", "aten::_to_copy",
                   "aten::resolve_conj", "aten::resolve_neg", "aten::lift_fresh", "aten::flatten", "aten::unflatten"):
        continue
    frames = [f for f in (ev.stack or []) if "sivae_hip" in f or "train_soft" in f or "bench.py" in f]
    where = frames[0].split("/")[-1] if frames else "(autograd / no python frame)"
    shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
    agg[(ev.name, where, shp)] += 1
for (name, where, shp), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print("%4d  %-22s %-60s %s" % (n, name, where[:60], shp))
