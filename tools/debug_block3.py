"""Composed ResBlockFn backward with every intermediate compared to torch fp64 (diagnostic, GPU box)."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import ops, functional as SF
from sivae_hip.nn import ResidualBlock

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

LOG = {}
for name in ("bn_bwd", "conv2d_fwd", "conv2d_wgrad"):
    orig = getattr(ops, name)
    def wrap(*a, _orig=orig, _name=name, **k):
        r = _orig(*a, **k)
        LOG.setdefault(_name, []).append((a, k, r))
        return r
    setattr(ops, name, wrap)

def run(B, C, H):
    LOG.clear()
    torch.manual_seed(0)
    blk = ResidualBlock(C, C)
    x = torch.randn(B, C, H, H)
    sd = {k: v.detach().double().clone() for k, v in blk.state_dict().items()}
    xr = x.double().requires_grad_()
    w1 = sd['conv1.weight'].requires_grad_(); w2 = sd['conv2.weight'].requires_grad_()
    a = F.conv2d(xr, w1, padding=1); a.retain_grad()
    z1 = F.batch_norm(a, None, None, sd['bn1.weight'], sd['bn1.bias'], training=True)
    h = F.leaky_relu(z1, 0.2); h.retain_grad()
    c = F.conv2d(h, w2, padding=1); c.retain_grad()
    z2 = F.batch_norm(c, None, None, sd['bn2.weight'], sd['bn2.bias'], training=True)
    s = z2 + xr; s.retain_grad()
    out = F.leaky_relu(s, 0.2)
    g = torch.randn_like(out)
    out.backward(g)
    blk = blk.cuda().train()
    xd = x.cuda().requires_grad_()
    yd = blk(xd)
    nf = len(LOG["conv2d_fwd"])
    yd.backward(g.float().cuda())
    print("==== B=%d C=%d H=%d  fwd y %.2e" % (B, C, H, rel(yd, out)))
    (a1, k1, r1), (a2, k2, r2) = LOG["bn_bwd"]
    print("bn2 inputs: dy %.2e out %.2e c %.2e" % (rel(a1[0], g), rel(a1[1], out), rel(a1[2], c)))
    print("bn2 outputs: dc %.2e dz %.2e" % (rel(r1[0], c.grad), rel(r1[1], s.grad)))
    print("bn1 inputs: dh %.2e a %.2e" % (rel(a2[0], h.grad), rel(a2[2], a)))
    print("bn1 outputs: da %.2e" % rel(r2[0], a.grad))
    for (aa, kk, rr) in LOG["conv2d_fwd"][nf:]:
        print("bwd conv2d_fwd call: Co=%s ks=%s acc=%s" % (aa[2], aa[3], kk.get("accumulate")))
    print("dx %.2e" % rel(xd.grad, xr.grad))

run(16, 64, 32)
run(4, 32, 8)
