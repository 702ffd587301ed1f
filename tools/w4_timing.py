"""Phase timing of conv_wino4_kernel items (GPU box; needs the -DW4_TIMING build of conv_wino4.hip that
tools/build_w4_timing.sh [ablate-bits ...] makes: tools/abx/w4_timing_<bits>.so, selected by W4_VARIANT=<bits>).
usage: python tools/w4_timing.py [B]     prints per layer the mean microseconds of: K loop | epilogue round 0 write+barrier |
round 0 read/store | rounds 1-3 | statistics tail, and the shader clock seen by clock64."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SIVAE_LIB"] = os.path.join(REPO, "tools", "ab", "w4_timing_%s.so" % os.environ.get("W4_VARIANT", "0"))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
import numpy as np
import torch
from sivae_hip import lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
PRO = bool(os.environ.get("BENCH_PRO"))
L = ctypes.CDLL(os.environ["SIVAE_LIB"])
buf = (ctypes.c_longlong * (256 * 8 * 16))()
for (Ci, Co, H) in [(64, 64, 256), (64, 64, 128), (64, 128, 128), (128, 128, 128), (256, 256, 64), (512, 512, 32)]:
    x = torch.randn(B, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    wq = ops.PackedW(w, 0)
    pro = None
    if PRO:
        pro = (torch.zeros(Ci, device="cuda"), torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda"),
               torch.zeros(Ci, device="cuda"), 0.2)
    for _ in range(int(os.environ.get("W4_REPS", "40"))):  # (the clocks ramp over the first milliseconds of load)
        ops.conv2d_fwd(x, wq, Co, 3, want_stats=True, pro=pro)
    torch.cuda.synchronize()
    assert L.sivae_debug_w4_read(buf) == 0
    d = np.frombuffer(buf, dtype=np.int64).reshape(256, 8, 16).astype(np.float64)
    it = d[:, 1:5, :]  # items 1..4 of every block (all layers here give a block >= 8 items at B = 128)
    us = lambda a, b: float(np.mean(it[:, :, b] - it[:, :, a])) / 100.0
    tot = us(0, 5)
    npair = min(Ci // 16, 8)
    pairs = [us(0, 8)] + [us(8 + k - 1, 8 + k) for k in range(1, npair)]
    gap = float(np.mean(d[:, 2:6, 0] - d[:, 1:5, 5])) / 100.0
    print("%4d->%-4d @%-3d: item %6.2f us = K %6.2f | r0 write %5.2f | r0 read/store %5.2f | r1-3 %5.2f | tail %5.2f | gap %5.2f ; chunk pairs %s"
          % (Ci, Co, H, tot, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), gap, " ".join("%.2f" % v for v in pairs)))
    del x, w, wq
