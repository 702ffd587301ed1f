"""Phase timing of conv_wino4_b6 (a -DB6_TIMING build: SIVAE_LIB=tools/abx/b6_timing.so python tools/b6_timing.py) and the
launch time of ablated builds (SIVAE_LIB=tools/abx/b6_ablN.so python tools/b6_timing.py plain)."""
import ctypes, os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import lib, ops

plain = len(sys.argv) > 1 and sys.argv[1] == "plain"
ops.WINO4_B6 = True
B = 256
shapes = [(512, 512, 32), (256, 256, 64), (128, 128, 128), (64, 64, 256)]
if os.environ.get("B6_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["B6_SHAPES"].split(",")]
for (Ci, Co, H) in shapes:
    x = torch.randn(B, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    wq = ops.PackedW(w, 0)
    for _ in range(12):  # (clocks ramp up over the first tens of milliseconds)
        ops.conv2d_fwd(x, wq, Co, 3, want_stats=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(6):
        ops.conv2d_fwd(x, wq, Co, 3, want_stats=True)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 6
    nsteps = Ci // 16
    items = B * (H // 16) * (H // 32) * (Co // 64)
    line = "%4d->%-4d @%-3d: %7.3f ms, %5.1f us per item, %5.2f us per step" % (Ci, Co, H, ms, ms * 1e3 / (items / 256.0),
                                                                                ms * 1e3 / (items / 256.0) / nsteps)
    if not plain:
        buf = (ctypes.c_longlong * (256 * 8))()
        ctypes.CDLL(lib.LIB_PATH).sivae_debug_b6_read(buf)
        a = np.array(buf, dtype=np.int64).reshape(256, 8).astype(np.float64)
        st, it = a[:, 4], a[:, 5]
        line += " | wave 0 of a block: phases %.2f / %.2f / %.2f us per step; epilogue + set-up %.2f us per item (means over blocks)" % (
            np.mean(a[:, 0] / st) / 100.0, np.mean(a[:, 1] / st) / 100.0, np.mean(a[:, 2] / st) / 100.0,
            np.mean(a[:, 3] / it) / 100.0)
        line += "; shader clock %.2f GHz" % (np.mean(a[:, 6] / a[:, 7]) * 0.1)
        buf2 = (ctypes.c_longlong * (256 * 12 * 4))()
        ctypes.CDLL(lib.LIB_PATH).sivae_debug_b6_read2(buf2)
        w = np.array(buf2, dtype=np.int64).reshape(256, 12, 4).astype(np.float64)
        per = w[:, :, :3] / (3 * w[:, :, 3:4]) / 100.0  # us per phase: [block][wave][stage]
        line += "\n      per wave and phase (split A | MFMA A + split B | MFMA B + transform | sum), us: " + "  ".join(
            "w%d %.2f|%.2f|%.2f|%.2f" % (k, *np.mean(per[:, k, :], 0), np.mean(per[:, k, :].sum(1))) for k in range(12))
    print(line, flush=True)
    del x
