"""Re-verify every bn_bwd / conv2d_wgrad / dgrad call of a real VAE step against torch fp64 on its ACTUAL inputs."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from sivae_hip import ops
import train_soft_intro_vae as T
from test_e2e_gpu import _engine

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

LOG = []
for name in ("bn_bwd", "conv2d_wgrad"):
    orig = getattr(ops, name)
    def wrap(*a, _orig=orig, _name=name, **k):
        r = _orig(*a, **k)
        LOG.append((_name, [t.clone() if isinstance(t, torch.Tensor) else t for t in a],
                    {kk: (tuple(x.clone() if isinstance(x, torch.Tensor) else x for x in v) if isinstance(v, tuple) else (v.clone() if isinstance(v, torch.Tensor) else v)) for kk, v in k.items()},
                    [t.clone() if isinstance(t, torch.Tensor) else t for t in (r if isinstance(r, tuple) else (r,))]))
        return r
    setattr(ops, name, wrap)

torch.manual_seed(0)
model = T.SoftIntroVAE(cdim=3, zdim=128, channels=[64, 128, 256], image_size=32).cuda().train()
eng, grads = _engine(model, False, dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8), 2e-4)
g = torch.Generator().manual_seed(1234)
real = torch.rand(16, 3, 32, 32, generator=g).cuda()
eng.vae_step(real, eps=torch.randn(16, 128, generator=g).cuda())
torch.cuda.synchronize()
for name, a, k, r in LOG:
    if name == "bn_bwd":
        dy, y, x, mean, invstd, gamma, slope = a[:7]
        beta, act = k.get("beta"), k.get("act_mode")
        dy, x, mean, invstd, gamma = [t.double() for t in (dy, x, mean, invstd, gamma)]
        C = x.shape[1]
        v = lambda t: t.view(1, C, 1, 1)
        xh = (x - v(mean)) * v(invstd)
        if act == 1:
            s = y.double()
        elif act == 2:
            s = xh * v(gamma) + v(beta.double())
        gz = torch.where(s > 0, dy, dy * slope) if act else dy
        N = x.numel() / C
        s1, s2 = gz.sum((0, 2, 3)), (gz * xh).sum((0, 2, 3))
        dx = v(gamma * invstd) * (gz - v(s1 / N) - xh * v(s2 / N))
        near = float((s.abs() < 1e-5).double().mean()) if act else 0.0
        out = "bn_bwd C=%d HW=%d act=%s dx %.2e" % (C, x.shape[2] * x.shape[3], act, rel(r[0], dx))
        if r[2] is not None:
            out += " dgamma %.2e dbeta %.2e" % (rel(r[2], s2), rel(r[3], s1))
        out += " frac|s|<1e-5: %.2e  exact0: %d" % (near, int((s == 0).sum()) if act else 0)
        print(out)
    else:
        x, dy, ks = a[:3]
        pro, up = k.get("pro"), k.get("upsample", False)
        xd = x.double()
        if pro is not None:
            pm, pi, pg, pb, sl = pro
            C = x.shape[1]
            v = lambda t: t.double().view(1, C, 1, 1)
            xd = F.leaky_relu((xd - v(pm)) * (v(pi) * v(pg)) + v(pb), sl)
        ref = torch.nn.grad.conv2d_weight(xd.cpu(), r[0].shape, dy.double().cpu(), padding=ks // 2)
        print("wgrad ks=%d Ci=%d Co=%d HW=%d pro=%s  %.2e" % (ks, x.shape[1], dy.shape[1], dy.shape[2] * dy.shape[3], pro is not None, rel(r[0], ref)))
