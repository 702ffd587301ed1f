"""BatchNorm backward: the one-pass persistent kernel (bn_fused.hip) against the three-launch form, per headline layer shape.
usage: python tools/bench_bn_fused.py [B] [nseg]   — ms per call and effective GB/s of the compulsory 3 tensor passes"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SHAPES = [(64, 256), (64, 128), (128, 128), (128, 64), (256, 64), (256, 32), (512, 32), (512, 16), (512, 8), (512, 4)]


def timeit(fn, reps=6):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


L = ops._lib.load()
for (C, H) in SHAPES:
    n = B * C * H * H * 4 / 1e9
    if n > 4.4:
        continue
    x = torch.randn(B, C, H, H, device="cuda"); dy = torch.randn(B, C, H, H, device="cuda")
    res = torch.randn(B, C, H, H, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    Bs = B // nseg
    xs = x.view(nseg, Bs, C, H * H)
    mean = xs.mean((1, 3)).reshape(-1).contiguous(); invstd = (1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).reshape(-1).contiguous()
    sup = L.sivae_bn_bwd_fused_supported(B, C, H, H, Bs)
    mask = None
    if H % 8 == 0:
        _, _, mask = ops.bn_apply_act_signmask(x, res, mean, invstd, g, b, 0.2, nseg=nseg)
    rows = [("mode2", lambda: ops.bn_bwd(dy, None, x, mean, invstd, g, beta=b, act_mode=2, nseg=nseg))]
    if mask is not None:
        rows.append(("mask+dzsum", lambda: ops.bn_bwd_signmask(dy, mask, x, mean, invstd, g, 0.2, dz_sum=True, nseg=nseg)))
    line = "%4d @%-3d B=%d nseg=%d (%.2f GB/tensor) fused_supported=%d" % (C, H, B, nseg, n, sup)
    for name, fn in rows:
        ts = []
        for fused in (False, True):
            ops.BN_FUSED = fused
            ts.append(timeit(fn))
        ops.BN_FUSED = True
        line += "  | %s: 3-launch %.3f ms, fused %.3f ms (%.0f GB/s of 3 passes) x%.2f" % (name, ts[0], ts[1], 3 * n / ts[1] * 1e3, ts[0] / ts[1])
    print(line, flush=True)
    del x, dy, res, mask
