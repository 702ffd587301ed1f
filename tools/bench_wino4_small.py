"""F(4x4,3x3) image-grid mode vs F(2x2,3x3) on the 8x8 / 4x4 maps, per (batch, channels) — the numbers behind the policy in
ops.conv2d_fwd / sivae_conv2d_wino4_small_pays (GPU box).  usage: python tools/bench_wino4_small.py"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops
L = ops._lib.load()


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / reps


print("%-28s %6s %3s %9s %9s %6s" % ("B Ci Co HxW pro", "items", "S", "F(4,3) us", "F(2,3) us", "ratio"))
for H in (8, 4):
    for (Ci, Co) in ((512, 512), (256, 256), (128, 256), (128, 128)):
        for B in (32, 64, 128, 256, 512):
            ipi = L.sivae_conv2d_wino4_images_per_item(H, H)
            if B % ipi:
                continue
            x = torch.randn(B, Ci, H, H, device="cuda")
            w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
            wq = ops.PackedW(w, 0)
            for pro in (None, 1):
                p = None
                if pro:
                    p = (torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda"),
                         torch.zeros(Ci, device="cuda"), 0.2)
                ops.WINO4_SMALL = True
                force = ops.WINO4_SMALL_FORCE if hasattr(ops, "WINO4_SMALL_FORCE") else None
                if force is not None:
                    ops.WINO4_SMALL_FORCE = True
                t4 = timeit(lambda: ops.conv2d_fwd(x, wq, Co, 3, want_stats=True, pro=p))
                if force is not None:
                    ops.WINO4_SMALL_FORCE = force
                ops.WINO4_SMALL = False
                t2 = timeit(lambda: ops.conv2d_fwd(x, wq, Co, 3, want_stats=True, pro=p))
                ops.WINO4_SMALL = True
                items = (B // ipi) * ((Co + 63) // 64)
                S = L.sivae_conv2d_wino4_splitk(B, Ci, Co, H, H)
                print("%-28s %6d %3d %9.1f %9.1f %6.2f" % ("%d %d %d %dx%d %s" % (B, Ci, Co, H, H, "pro" if pro else "-"),
                                                          items, S, t4, t2, t2 / t4))

# ---- weight gradient: the image-strip mode of conv_wino4_wgrad.hip vs the kernels these maps used before
print()
print("%-28s %9s %9s %6s   (weight gradient; 'pays' = the library's own policy)" % ("B Ci Co HxW pro", "F(4,3) us", "before us", "ratio"))
for H in (8, 4):
    for (Ci, Co) in ((512, 512), (256, 256), (128, 256), (128, 128)):
        for B in (32, 64, 128, 256, 512):
            x = torch.randn(B, Ci, H, H, device="cuda")
            dy = torch.randn(B, Co, H, H, device="cuda")
            for pro in (None, 1):
                p = None
                if pro:
                    p = (torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda"),
                         torch.zeros(Ci, device="cuda"), 0.2)
                ops.WINO4_SMALL = True
                f0 = ops.WINO4_FORCE
                ops.WINO4_FORCE = True
                t4 = timeit(lambda: ops.conv2d_wgrad(x, dy, 3, pro=p))
                ops.WINO4_FORCE = f0
                ops.WINO4_SMALL = False
                t2 = timeit(lambda: ops.conv2d_wgrad(x, dy, 3, pro=p))
                ops.WINO4_SMALL = True
                print("%-28s %9.1f %9.1f %6.2f  pays=%d" % ("%d %d %d %dx%d %s" % (B, Ci, Co, H, H, "pro" if pro else "-"), t4, t2,
                                                          t2 / t4, L.sivae_conv2d_wino4_wgrad_pays(B, Ci, Co, H, H)))
