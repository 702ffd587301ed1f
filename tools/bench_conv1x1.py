"""Micro-benchmark of the streaming 1x1 kernel at shard-size launches (GPU box).
usage: [SIVAE_CONV1X1_SMALL_TILES=0] python tools/bench_conv1x1.py"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

print("small tiles:", os.environ.get("SIVAE_CONV1X1_SMALL_TILES", "1"), "lib:", os.environ.get("SIVAE_LIB", "in-tree"))
for B in (8, 16, 32, 128):
    for (Ci, Co, H) in [(64, 128, 128), (128, 64, 128), (128, 256, 64), (256, 128, 64), (256, 512, 32), (512, 256, 32)]:
        x = torch.randn(B, Ci, H, H, device="cuda")
        w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
        wp = ops.pack_weight(w, 0)
        y = torch.empty(B, Co, H, H, device="cuda")
        t = timeit(lambda: ops.conv2d_fwd(x, wp, Co, 1, out=y))
        gb = B * (Ci + Co) * H * H * 4 / 1e9
        print("B=%3d %4d->%-4d @%-3d : %7.1f us  %5.2f TB/s  %6.1f TF/s" % (B, Ci, Co, H, t, gb / t * 1e-6 * 1e6 / 1e0 / 1e0 if False else gb / (t * 1e-6) / 1e3, 2.0 * B * H * H * Ci * Co / (t * 1e-6) / 1e12))
