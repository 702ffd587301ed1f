"""HBM-bandwidth micro-benchmark of the BatchNorm / elementwise passes on the headline activation shapes (GPU box).
usage: python tools/bench_bn.py [B]      — prints ms and algorithmic GB/s (compulsory reads + writes) per pass"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(64, 256), (128, 128), (256, 64), (512, 32), (512, 16)]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for (C, H) in SHAPES:
    n = B * C * H * H * 4 / 1e9  # GB per full-resolution tensor
    x = torch.randn(B, C, H, H, device="cuda")
    res = torch.randn(B, C, H, H, device="cuda")
    resh = torch.randn(B, C, H // 2, H // 2, device="cuda")
    dy = torch.randn(B, C, H, H, device="cuda")
    dyh = torch.randn(B, C, H // 2, H // 2, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, invstd = ops.bn_stats(x)
    y = ops.bn_apply_act(x, res, mean, invstd, g, b)
    out = "%4d @%-3d (%.2f GB/tensor):" % (C, H, n)
    rows = [
        ("copy", lambda: y.copy_(x), 2),
        ("stats", lambda: ops.bn_stats(x), 1),
        ("apply+res", lambda: ops.bn_apply_act(x, res, mean, invstd, g, b, out=y), 3),
        ("apply+resup", lambda: ops.bn_apply_act(x, resh, mean, invstd, g, b, out=y, res_up=True), 2.25),
        ("apply+res+pool", lambda: ops.bn_apply_act_pool(x, res, mean, invstd, g, b, want_full=True), 3.25),
        ("bwd(dz)", lambda: ops.bn_bwd(dy, y, x, mean, invstd, g, want_dz=True, act_mode=1), 3 + 3 + 2),
        ("bwd(pooled dy, dz)", lambda: ops.bn_bwd(dyh, y, x, mean, invstd, g, want_dz=True, act_mode=1,
                                                  dy_pooled=True), 2.25 + 2.25 + 2),
        ("bwd(dzsum)", lambda: ops.bn_bwd_dzsum(dy, y, x, mean, invstd, g), 3 + 3 + 1.25),
        ("bwd(mode2)", lambda: ops.bn_bwd(dy, None, x, mean, invstd, g, want_dz=False, beta=b, act_mode=2),
         2 + 2 + 1),
    ]
    print(out)
    for name, fn, tensors in rows:
        t = timeit(fn)
        print("    %-20s %7.3f ms  %6.0f GB/s (%.2f tensor passes)" % (name, t, tensors * n / t * 1e3, tensors))
    del x, res, resh, dy, dyh, y
