"""Stage-wise backward of one residual block vs torch fp64 (diagnostic, GPU box)."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
d = lambda t: t.detach().float().cuda().contiguous()

def run(B, C, H):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, H, dtype=torch.float64, requires_grad=True)
    w1 = (torch.randn(C, C, 3, 3, dtype=torch.float64) / (C * 9) ** 0.5).requires_grad_()
    w2 = (torch.randn(C, C, 3, 3, dtype=torch.float64) / (C * 9) ** 0.5).requires_grad_()
    g1 = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_(); b1 = (torch.rand(C, dtype=torch.float64) - 0.5).requires_grad_()
    g2 = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_(); b2 = (torch.rand(C, dtype=torch.float64) - 0.5).requires_grad_()
    a = F.conv2d(x, w1, padding=1); a.retain_grad()
    z1 = F.batch_norm(a, None, None, g1, b1, training=True)
    h = F.leaky_relu(z1, 0.2); h.retain_grad()
    c = F.conv2d(h, w2, padding=1); c.retain_grad()
    z2 = F.batch_norm(c, None, None, g2, b2, training=True)
    s = z2 + x
    out = F.leaky_relu(s, 0.2)
    g = torch.randn_like(out)
    s.retain_grad()
    out.backward(g)
    print("==== B=%d C=%d H=%d" % (B, C, H))
    mean2 = c.mean((0, 2, 3)); inv2 = 1 / torch.sqrt(c.var((0, 2, 3), unbiased=False) + 1e-5)
    mean1 = a.mean((0, 2, 3)); inv1 = 1 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)
    dc, dz, dg2, db2 = ops.bn_bwd(d(g), d(out), d(c), d(mean2), d(inv2), d(g2), 0.2, want_dz=True, act_mode=1)
    print("bn2: dc %.2e dz %.2e dg2 %.2e db2 %.2e" % (rel(dc, c.grad), rel(dz, s.grad), rel(dg2, g2.grad), rel(db2, b2.grad)))
    pro1 = (d(mean1), d(inv1), d(g1), d(b1), 0.2)
    dw2 = ops.conv2d_wgrad(d(a), d(c.grad), 3, pro=pro1)
    print("wgrad2(pro) %.2e" % rel(dw2, w2.grad))
    dh = ops.conv2d_fwd(d(c.grad), ops.pack_weight(d(w2), 1), C, 3)
    print("dgrad2 %.2e" % rel(dh, h.grad))
    da, _, dg1, db1 = ops.bn_bwd(d(h.grad), None, d(a), d(mean1), d(inv1), d(g1), 0.2, beta=d(b1), act_mode=2)
    print("bn1: da %.2e dg1 %.2e db1 %.2e" % (rel(da, a.grad), rel(dg1, g1.grad), rel(db1, b1.grad)))
    dw1 = ops.conv2d_wgrad(d(x), d(a.grad), 3)
    print("wgrad1 %.2e" % rel(dw1, w1.grad))
    dx = d(s.grad).clone()
    ops.conv2d_fwd(d(a.grad), ops.pack_weight(d(w1), 1), C, 3, out=dx, accumulate=True)
    print("dx(acc) %.2e" % rel(dx, x.grad))

run(16, 64, 32)
run(16, 64, 32)
run(4, 32, 8)
run(16, 128, 32)
