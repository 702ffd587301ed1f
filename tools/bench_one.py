"""Run ONE conv shape a few times (for rocprofv3 --pmc). usage: bench_one.py fwd|wino|wgrad B Ci Co H ks [reps]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops
kind, B, Ci, Co, H, ks = sys.argv[1], *[int(v) for v in sys.argv[2:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
x = torch.randn(B, Ci, H, H, device="cuda")
dy = torch.randn(B, Co, H, H, device="cuda")
w = torch.randn(Co, Ci, ks, ks, device="cuda") / (Ci * ks * ks) ** 0.5
wp = ops.PackedW(w, 0) if kind == "wino" else ops.pack_weight(w, 0)
for _ in range(reps):
    if kind in ("fwd", "wino"):
        ops.conv2d_fwd(x, wp, Co, ks, want_stats=True)
    else:
        ops.conv2d_wgrad(x, dy, ks)
torch.cuda.synchronize()
print("done")
