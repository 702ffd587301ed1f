"""Per-phase timing of the one-launch fp32 BatchNorm backward (bn_fused.hip built with -DBF_TIMING into
tools/abx/libsivae_b16timing.so: thread 0 of every block stamps s_memrealtime, 100 MHz, at the phase boundaries of every
group).  usage (GPU box):  SIVAE_LIB=tools/abx/libsivae_b16timing.so python tools/bn_fused_timing.py [B C H W nseg] [pooled] [dzsum]
stamps: 0 group start | 1 raw vectors landed (vmcnt 0) | 2 sums folded + published | 3 arrived | 4 released | 5 coefficients
ready | 6 phase 2 issued | 7 block barrier passed (stores complete: __syncthreads waits vmcnt 0)"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import lib, ops, ops16  # noqa: E402

args = [a for a in sys.argv[1:] if a.isdigit()]
B, C, H, W, nseg = (int(a) for a in args[:5]) if len(args) >= 5 else (256, 64, 256, 256, 2)
pooled, dzsum = "pooled" in sys.argv, "dzsum" in sys.argv
dev = torch.device("cuda:0")
L = lib.load()
L.sivae_debug_bf_timing.restype = None
plan = (ctypes.c_int * 6)()
assert L.sivae_debug_bf_plan(B, C, H, W, B // nseg, 10, plan) == 0
nu, spc, cpg, ngroups, nb_sub, nsub = list(plan)
nblk = nb_sub * nsub
print("shape B=%d C=%d %dx%d nseg=%d%s%s: NU=%d spc=%d cpg=%d groups=%d blocks=%d x %d half-grids"
      % (B, C, H, W, nseg, " pooled-dy" if pooled else "", " dz-sums" if dzsum else "", nu, spc, cpg, ngroups, nb_sub, nsub))
x = torch.randn(B, C, H, W, device=dev)
dy = torch.randn(B, C, H // 2 if pooled else H, W // 2 if pooled else W, device=dev)
mask = torch.randint(0, 256, (L.sivae_bn_signmask_bytes(B, C, H * W),), device=dev, dtype=torch.uint8)
mean, invstd = torch.randn(nseg * C, device=dev) * 0.1, torch.rand(nseg * C, device=dev) + 0.5
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1


def run():
    return ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, dy_pooled=pooled, dz_sum=dzsum, want_dz=dzsum, nseg=nseg)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
gb = (x.numel() * 4 * 2 + dy.numel() * 4 + mask.numel()) / 1e9
print("launch %.1f us, %.2f GB -> %.2f TB/s" % (ms * 1e3, gb, gb / ms))
ts = torch.zeros(nblk * ngroups * 8, dtype=torch.int64, device=dev)
L.sivae_debug_bf_timing(ctypes.c_void_p(ts.data_ptr()))
run()
torch.cuda.synchronize()
L.sivae_debug_bf_timing(None)
t = ts.cpu().numpy().reshape(nblk, ngroups, 8).astype(np.float64) / 100.0  # us
names = ["load (0->1)", "sums+publish (1->2)", "arrive (2->3)", "wait (3->4)", "fold (4->5)", "phase 2 (5->6)",
         "stores done (6->7)"]
for sub in range(nsub):
    blocks = slice(sub * nb_sub, (sub + 1) * nb_sub)
    groups = list(range(sub, ngroups, nsub))
    tt = t[blocks][:, groups]            # [blocks, groups of this half, 8]
    act = tt[:, :, 7] > 0
    t0 = tt[:, 0, 0][tt[:, 0, 0] > 0].min()
    print("half-grid %d: %d groups, span %.1f us" % (sub, len(groups), tt[:, :, 7].max() - t0))
    d = np.diff(tt, axis=2)
    print("  %-22s %8s %8s %8s" % ("interval", "mean", "p50", "max"))
    for k, n in enumerate(names):
        v = d[:, :, k][act]
        print("  %-22s %8.2f %8.2f %8.2f" % (n, v.mean(), np.median(v), v.max()))
    g = (tt[:, :, 7] - tt[:, :, 0])[act]
    print("  %-22s %8.2f %8.2f %8.2f" % ("group (0->7)", g.mean(), np.median(g), g.max()))
    # skew of the arrivals and of the releases inside a group (what the slowest block costs everybody)
    arr = tt[:, :, 3]
    rel = tt[:, :, 4]
    sk = [(arr[:, i][act[:, i]].max() - arr[:, i][act[:, i]].min()) for i in range(len(groups))]
    lat = [(rel[:, i][act[:, i]].min() - arr[:, i][act[:, i]].max()) for i in range(len(groups))]
    rs = [(rel[:, i][act[:, i]].max() - rel[:, i][act[:, i]].min()) for i in range(len(groups))]
    print("  arrival skew (last - first arrival)      mean %.2f us" % np.mean(sk))
    print("  barrier latency (first release - last arrival) mean %.2f us" % np.mean(lat))
    print("  release skew (last - first release)      mean %.2f us" % np.mean(rs))
