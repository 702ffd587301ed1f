"""Micro-benchmark of the MFMA conv kernels on the headline layer shapes (GPU box).
usage: python tools/bench_conv.py [B]   (env SIVAE_FWD3_VARIANT / SIVAE_WGRAD3_VARIANT select tile variants)"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
what = sys.argv[2] if len(sys.argv) > 2 else "fwd,wgrad"
SHAPES = [  # (Ci, Co, H, ks)
    (64, 128, 128, 3), (128, 128, 128, 3), (128, 256, 64, 3), (256, 256, 64, 3), (256, 512, 32, 3),
    (512, 512, 32, 3), (512, 512, 16, 3), (512, 512, 8, 3), (64, 64, 256, 3), (128, 64, 128, 3),
    (64, 128, 128, 1), (128, 64, 128, 1), (256, 128, 64, 1), (512, 256, 32, 1), (3, 64, 256, 5), (64, 3, 256, 5),
]

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

print("variant fwd3=%s wgrad3=%s B=%d" % (os.environ.get("SIVAE_FWD3_VARIANT", "0"), os.environ.get("SIVAE_WGRAD3_VARIANT", "0"), B))
KS_ONLY = int(os.environ.get("BENCH_KS", "0"))
WINO_ONLY = bool(os.environ.get("BENCH_WINO_ONLY"))
for (Ci, Co, H, ks) in SHAPES:
    if KS_ONLY and ks != KS_ONLY:
        continue
    x = torch.randn(B, Ci, H, H, device="cuda")
    dy = torch.randn(B, Co, H, H, device="cuda")
    w = torch.randn(Co, Ci, ks, ks, device="cuda") / (Ci * ks * ks) ** 0.5
    wp = ops.pack_weight(w, 0)
    fl = 2.0 * B * H * H * Ci * Co * ks * ks
    out = "%4d->%-4d @%-3d k%d :" % (Ci, Co, H, ks)
    if "fwd" in what:
        if not WINO_ONLY:
            t = timeit(lambda: ops.conv2d_fwd(x, wp, Co, ks, want_stats=(ks != 1)))  # (conv_expand has no BatchNorm)
            gbs = (B * (Ci + Co) * H * H * 4) / t / 1e6
            out += "  fwd %7.3f ms %6.1f TF %5.0f GB/s" % (t, fl / t / 1e9, gbs)
            if ks == 1:
                yb = torch.randn(B, Co, H, H, device="cuda")
                t = timeit(lambda: ops.conv2d_fwd(x, wp, Co, ks, out=yb, accumulate=True))
                out += "  +acc %7.3f ms %5.0f GB/s" % (t, (B * (Ci + 2 * Co) * H * H * 4) / t / 1e6)
                del yb
        if ks == 3 and H >= 16:
            wq = ops.PackedW(w, 0)
            pro = None
            if os.environ.get("BENCH_PRO"):
                pro = (torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda"),
                       torch.zeros(Ci, device="cuda"), 0.2)
            t = timeit(lambda: ops.conv2d_fwd(x, wq, Co, ks, want_stats=True, pro=pro))
            if ops.WINO4 and max(Ci, Co) <= ops.WINO4_MAXC and (H % 32 == 0 or H == 16):
                ops.WINO4 = False
                t2 = timeit(lambda: ops.conv2d_fwd(x, wq, Co, ks, want_stats=True, pro=pro))
                ops.WINO4 = True
                out += "  F(4,3) %7.3f ms %6.1f TF(alg) %5.1f TF(exec) | F(2,3) %7.3f ms (x%.2f)" % (
                    t, fl / t / 1e9, fl / 4 / t / 1e9, t2, t2 / t)
            else:
                out += "  wino %7.3f ms %6.1f TF(alg) %5.1f TF(exec)" % (t, fl / t / 1e9, fl * 16 / 36 / t / 1e9)
    if "up" in what and ks == 3 and H >= 32:
        xs_ = torch.randn(B, Ci, H // 2, H // 2, device="cuda")
        wq = ops.PackedW(w, 0)
        t = timeit(lambda: ops.conv2d_fwd(xs_, wq, Co, ks, want_stats=True, upsample=True))
        ops.WINO_UP = False
        t2 = timeit(lambda: ops.conv2d_fwd(xs_, wq, Co, ks, want_stats=True, upsample=True))
        ops.WINO_UP = True
        out += "  up-F(2,2) %7.3f ms %6.1f TF(alg) %5.1f TF(exec) | F(2,3)+upsample %7.3f ms %6.1f TF(alg)" % (
            t, fl / t / 1e9, fl * 9 / 36 / t / 1e9, t2, fl / t2 / 1e9)
    if "updg" in what and ks == 3 and H >= 32:
        wq = ops.PackedW(w, 0)
        t = timeit(lambda: ops.conv2d_up_dgrad(dy, wq, Ci))
        wd = ops.PackedW(w, 1)
        t2 = timeit(lambda: ops.upsample2_bwd(ops.conv2d_fwd(dy, wd, Ci, 3)))
        out += "  up-dgrad F(2,2) %7.3f ms %6.1f TF(alg) | F(2,3) + 2x2 sum %7.3f ms %6.1f TF(alg)" % (
            t, fl / t / 1e9, t2, fl / t2 / 1e9)
    if "upwg" in what and ks == 3 and H >= 32:
        xs_ = torch.randn(B, Ci, H // 2, H // 2, device="cuda")
        t = timeit(lambda: ops.conv2d_wgrad(xs_, dy, 3, upsample=True))
        ops.WINO_UP = False
        t2 = timeit(lambda: ops.conv2d_wgrad(xs_, dy, 3, upsample=True))
        ops.WINO_UP = True
        out += "  up-wgrad F(2,2) %7.3f ms %6.1f TF(alg) | F(2,3)+upsample %7.3f ms %6.1f TF(alg)" % (
            t, fl / t / 1e9, t2, fl / t2 / 1e9)
    if "wgrad" in what.split(","):
        prow = None
        if os.environ.get("BENCH_PRO") and ks == 3:
            prow = (torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda"),
                    torch.zeros(Ci, device="cuda"), 0.2)
        t = timeit(lambda: ops.conv2d_wgrad(x, dy, ks, pro=prow))
        out += "  wgrad %7.3f ms %6.1f TF" % (t, fl / t / 1e9)
        if ks == 3 and ops.WINO4_WGRAD and WINO_ONLY:
            ops.WINO4_WGRAD = False
            t2 = timeit(lambda: ops.conv2d_wgrad(x, dy, ks, pro=prow))
            ops.WINO4_WGRAD = True
            out += " (%5.1f TF exec) | F(2,3) %7.3f ms (x%.2f)" % (fl / 4 / t / 1e9, t2, t2 / t)
        if ks == 3 and H >= 16 and ops.WINO_WGRAD and not WINO_ONLY:
            ops.WINO_WGRAD = False
            t = timeit(lambda: ops.conv2d_wgrad(x, dy, ks))
            ops.WINO_WGRAD = True
            out += "  (direct %7.3f ms %6.1f TF)" % (t, fl / t / 1e9)
    if ks == 5 and min(Ci, Co) <= 3:
        if Co <= 3:
            wq = ops.pack5_smallco(w, 0)
            t = timeit(lambda: ops.conv5_smallco_fwd(x, wq, Co))
            out += "  | edge fwd %7.3f ms %6.1f TF" % (t, fl / t / 1e9)
        else:
            wq = ops.pack5_smallco(w, 1)
            t = timeit(lambda: ops.conv5_smallco_fwd(dy, wq, Ci))
            out += "  | edge dgrad %7.3f ms %6.1f TF" % (t, fl / t / 1e9)
        t = timeit(lambda: ops.conv5_edge_wgrad(x, dy))
        out += "  edge wgrad %7.3f ms %6.1f TF" % (t, fl / t / 1e9)
    print(out)
    del x, dy
