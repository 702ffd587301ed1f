"""conv_wino4 (fp32 MFMA) against conv_wino4_b6 (six bf16 MFMAs per fp32 product) on the headline layer shapes, inside one
process (GPU box).  usage: python tools/bench_wino4_b6.py [B=256] [pro]   -> ms per launch of both + ratio + max |diff| rel."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PRO = len(sys.argv) > 2 and sys.argv[2] == "pro"
SHAPES = [(512, 512, 32), (256, 256, 64), (128, 128, 128), (64, 64, 256), (64, 64, 128), (512, 512, 16), (64, 128, 128),
          (128, 256, 64), (256, 512, 32), (512, 256, 32), (128, 64, 128)]


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for (Ci, Co, H) in SHAPES:
    x = torch.randn(B, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    wq = ops.PackedW(w, 0)
    pro = None
    if PRO:
        pro = (torch.randn(Ci, device="cuda") * 0.1, torch.rand(Ci, device="cuda") + 0.5, torch.rand(Ci, device="cuda") + 0.5,
               torch.randn(Ci, device="cuda") * 0.1, 0.2)
    outs = {}
    times = {}
    for rep in range(2):
        for b6 in (False, True):
            ops.WINO4_B6 = b6
            t = timeit(lambda: ops.conv2d_fwd(x, wq, Co, 3, want_stats=True, pro=pro))
            times.setdefault(b6, []).append(t)
    for b6 in (False, True):
        ops.WINO4_B6 = b6
        outs[b6] = ops.conv2d_fwd(x, wq, Co, 3, want_stats=True, pro=pro)[0]
    torch.cuda.synchronize()
    d = float((outs[True] - outs[False]).abs().max() / outs[False].abs().max())
    t0, t1 = min(times[False]), min(times[True])
    fl = 2.0 * B * H * H * Ci * Co * 9
    print("%4d->%-4d @%-3d B%d%s: fp32 %7.3f ms (%5.1f TF alg)  b6 %7.3f ms (%5.1f TF alg)  x%.2f   max|diff|/max %.2e"
          % (Ci, Co, H, B, " pro" if PRO else "", t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t0 / t1, d), flush=True)
    del x, outs
