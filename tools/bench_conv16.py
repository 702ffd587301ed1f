"""Per-layer timing of the bf16-mode conv kernels (forward, and weight gradient) on the layer shapes of a config.
usage: bench_conv16.py [celeb128|celeb256] [B] [fwd|wgrad|both]   (A/B: SIVAE_BF16_CONV_TILE=0 in the environment)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops16  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "celeb128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
what = sys.argv[3] if len(sys.argv) > 3 else "both"
chans = {"celeb128": [64, 128, 256, 512, 512], "celeb256": [64, 128, 256, 512, 512, 512]}[cfg]
size = {"celeb128": 128, "celeb256": 256}[cfg]
layers = []
cc, sz = chans[0], size // 2
for ch in chans[1:]:
    layers += [(cc, ch, sz, 3), (ch, ch, sz, 3)]
    if cc != ch:
        layers.append((cc, ch, sz, 1))
    cc, sz = ch, sz // 2
layers += [(cc, cc, sz, 3)]
layers += [(64, 64, size, 3), (3, 64, size, 5), (64, 3, size, 5)]
dev = "cuda"


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("# %s B=%d tile-switch=%s" % (cfg, B, os.environ.get("SIVAE_BF16_CONV_TILE", "1")))
tot_f = tot_w = 0.0
for (ci, co, hw, ks) in layers:
    x = torch.randn(B, ops16.cblocks(ci), hw, hw, 8, device=dev).bfloat16()
    dy = torch.randn(B, ops16.cblocks(co), hw, hw, 8, device=dev).bfloat16()
    w = torch.randn(co, ci, ks, ks, device=dev) / (ci * ks * ks) ** 0.5
    wp = ops16.PackedW16(w, 0)
    fl = 2.0 * B * hw * hw * ci * co * ks * ks
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    row = "%4d->%4d @%3d k%d" % (ci, co, hw, ks)
    if what in ("fwd", "both"):
        t = timeit(lambda: ops16.conv2d(x, wp, ci, co, ks, want_stats=ks == 3))
        tot_f += t
        row += "  fwd %8.1f us %7.1f TF/s %5.2f TB/s" % (t, fl / t / 1e6, gb / t * 1e3)
    if what in ("wgrad", "both"):
        t = timeit(lambda: ops16.conv2d_wgrad(x, dy, ci, co, ks))
        tot_w += t
        row += "  wgrad %8.1f us %7.1f TF/s" % (t, fl / t / 1e6)
    print(row)
print("total fwd %.1f us, wgrad %.1f us" % (tot_f, tot_w))
