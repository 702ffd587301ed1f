"""CPU study for the next kernel generation: how much fp32 accuracy do the Winograd forms cost
through the WHOLE network?  Every 3x3 convolution of the CPU oracle's forward pass is replaced by an fp32 emulation of
   F(2x2,3x3)  — what conv_wino.hip computes today (transforms and the channel contraction in fp32), or
   F(4x4,3x3)  — a candidate (interpolation points 0, +-1, +-2, inf; transform constants up to 8 and 1/24),
   bf16 pieces — the other candidate: every fp32 operand split into three bf16 pieces, the products of pieces (exact in
                 fp32) summed with fp32 accumulation on the 16x faster bf16 MFMA: 9, 6 or 3 piece pairs per product,
and encoder mu / logvar and the reconstruction are compared with an fp64 run of the plain algorithm.  The direct fp32
convolution's own error against fp64 is printed next to them: that is the noise floor the 1e-4 parity gate sits on.

usage: python tools/wino_numerics.py [cifar|celeb128n|celeb256n] [batch]        (CPU only, a few minutes)
"""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import sivae_oracle as O  # noqa: E402

# F(2x2,3x3)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
# F(4x4,3x3), Lavin & Gray
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                    [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                   [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                   dtype=torch.float64)


def wino_conv3x3(x, w, m):
    """3x3 / pad 1 convolution through F(m x m, 3x3) with every step in x.dtype (fp32): input transform, per-frequency
    channel contraction (the MFMA GEMM), output transform."""
    BT, G, AT = (BT2, G2, AT2) if m == 2 else (BT4, G4, AT4)
    dt = x.dtype
    BT, G, AT = BT.to(dt), G.to(dt), AT.to(dt)
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    t = m + 2
    nh, nw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, 1 + nw * m - W, 1, 1 + nh * m - H))
    tiles = xp.unfold(2, t, m).unfold(3, t, m)                     # [B, Ci, nh, nw, t, t]
    V = torch.einsum("ik,bcxykl,jl->bcxyij", BT, tiles, BT)        # B^T d B
    U = torch.einsum("ik,ockl,jl->ocij", G, w, G)                  # G g G^T
    M = torch.einsum("ocij,bcxyij->boxyij", U, V)                  # sum over input channels, per frequency
    Y = torch.einsum("ik,boxykl,jl->boxyij", AT, M, AT)            # A^T M A -> [B, Co, nh, nw, m, m]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, Co, nh * m, nw * m)
    return Y[:, :, :H, :W].contiguous()


def split_bf16(a, n=3):
    """a (fp32) = sum of n bf16-representable pieces (exact for n = 3 unless the low piece underflows)"""
    out, r = [], a
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def bf16_pieces_conv3x3(x, w, terms):
    """3x3 conv with every fp32 product replaced by products of bf16 pieces (exact in fp32: 8 x 8 mantissa bits),
    accumulated in fp32 — what a bf16 MFMA with fp32 accumulation computes.  terms: 9 = all piece pairs,
    6 = pairs with i + j <= 2 (drops contributions below 2^-24), 3 = i + j <= 1."""
    xs, ws = split_bf16(x), split_bf16(w)
    lim = {9: 4, 6: 2, 3: 1}[terms]
    y = None
    for i in range(3):
        for j in range(3):
            if i + j <= lim:
                t = F_conv2d_orig(xs[i], ws[j], padding=1)
                y = t if y is None else y + t
    return y


F_conv2d_orig = F.conv2d


class patched_conv:
    """route every 3x3 / padding-1 F.conv2d through the Winograd emulation of order m (m = 2, 4) or through the
    bf16-piece products (m = -9, -6, -3)"""

    def __init__(self, m):
        self.m = m

    def __enter__(self):
        self.orig = F.conv2d
        m = self.m

        def conv(x, w, bias=None, stride=1, padding=0, *a, **k):
            if w.shape[2] == 3 and w.shape[3] == 3 and padding == 1 and bias is None and x.shape[2] % 2 == 0:
                return wino_conv3x3(x, w, m) if m > 0 else bf16_pieces_conv3x3(x, w, -m)
            return self.orig(x, w, bias, stride, padding, *a, **k)

        F.conv2d = conv
        return self

    def __exit__(self, *exc):
        F.conv2d = self.orig


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cifar"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfgs = {"cifar": ([64, 128, 256], 32, 128), "celeb128n": ([16, 32, 64, 128, 128], 128, 64),
            "celeb256n": ([8, 16, 32, 64, 64, 64], 256, 64)}
    channels, size, zdim = cfgs[name]
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    P32 = O.init_params(3, zdim, channels, size, seed=0)
    x32 = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(1))

    def run(P, x, m=None):
        P = {k: v.clone() for k, v in P.items()}
        with torch.no_grad():
            if m is None:
                mu, lv = O.encode(P, x, channels, size, training=True)
                rec = O.decode(P, mu, channels, size, training=True)
            else:
                with patched_conv(m):
                    mu, lv = O.encode(P, x, channels, size, training=True)
                    rec = O.decode(P, mu, channels, size, training=True)
        return mu, lv, rec

    P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P32.items()}
    ref = run(P64, x32.double())
    rows = [("direct fp32", run(P32, x32)), ("F(2x2,3x3) fp32", run(P32, x32, 2)), ("F(4x4,3x3) fp32", run(P32, x32, 4)),
            ("bf16 pieces x9", run(P32, x32, -9)), ("bf16 pieces x6", run(P32, x32, -6)),
            ("bf16 pieces x3", run(P32, x32, -3))]
    print("%s  channels=%s  %dx%d  B=%d   (max-norm relative error against the fp64 direct run)" % (name, channels, size, size, B))
    print("%-18s %12s %12s %12s" % ("", "mu", "logvar", "reconstruction"))
    for tag, (mu, lv, rec) in rows:
        print("%-18s %12.3e %12.3e %12.3e" % (tag, rel(mu, ref[0]), rel(lv, ref[1]), rel(rec, ref[2])))
    # one deep layer in isolation (K = 512 channels)
    g = torch.Generator().manual_seed(2)
    xl = torch.randn(4, 512, 16, 16, generator=g)
    wl = torch.randn(512, 512, 3, 3, generator=g) / (512 * 9) ** 0.5
    r64 = F.conv2d(xl.double(), wl.double(), padding=1)
    print("single 512->512 3x3 layer @16x16:  direct %.3e   F(2x2,3x3) %.3e   F(4x4,3x3) %.3e   bf16x9 %.3e   bf16x6 %.3e   "
          "bf16x3 %.3e" % (rel(F.conv2d(xl, wl, padding=1), r64), rel(wino_conv3x3(xl, wl, 2), r64),
                           rel(wino_conv3x3(xl, wl, 4), r64), rel(bf16_pieces_conv3x3(xl, wl, 9), r64),
                           rel(bf16_pieces_conv3x3(xl, wl, 6), r64), rel(bf16_pieces_conv3x3(xl, wl, 3), r64)))


if __name__ == "__main__":
    main()
