"""Single ResidualBlock forward/backward: HIP block Function vs torch fp64 (diagnostic, GPU box)."""
import os, sys
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))
from sivae_hip import functional as SF, ops
from sivae_hip.nn import ResidualBlock

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def run(B, Ci, Co, H, post):
    torch.manual_seed(0)
    blk = ResidualBlock(Ci, Co)
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(B, Ci, H, H)
    ref = {k: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k and 'num_b' not in k) for k, v in blk.state_dict().items()}
    xr = x.double().requires_grad_()
    idt = F.conv2d(xr, ref['conv_expand.weight']) if 'conv_expand.weight' in ref else xr
    a = F.conv2d(xr, ref['conv1.weight'], padding=1)
    z1 = F.batch_norm(a, None, None, ref['bn1.weight'], ref['bn1.bias'], training=True)
    h = F.leaky_relu(z1, 0.2)
    c = F.conv2d(h, ref['conv2.weight'], padding=1)
    z2 = F.batch_norm(c, None, None, ref['bn2.weight'], ref['bn2.bias'], training=True)
    out = F.leaky_relu(z2 + idt, 0.2)
    y = F.avg_pool2d(out, 2) if post == 'pool' else (F.interpolate(out, scale_factor=2, mode='nearest') if post == 'up' else out)
    g = torch.randn_like(y)
    y.backward(g)
    blk = blk.cuda().train()
    xd = x.cuda().requires_grad_()
    yd = blk(xd, post=post)
    yd.backward(g.float().cuda())
    print("---- B=%d %d->%d @%d post=%s" % (B, Ci, Co, H, post))
    print("fwd y", rel(yd, y))
    print("dx", rel(xd.grad, xr.grad))
    for k in ('conv_expand.weight', 'conv1.weight', 'bn1.weight', 'bn1.bias', 'conv2.weight', 'bn2.weight', 'bn2.bias'):
        if k in ref:
            mod, attr = k.split('.')
            print("%-20s %.3e" % (k, rel(getattr(getattr(blk, mod), attr).grad, ref[k].grad)))
    # direct look at the BN1 statistics and the activation sign agreement
    with torch.no_grad():
        wp = ops.pack_weight(blk.conv1.weight.detach(), 0)
        ah, part = ops.conv2d_fwd(x.cuda(), wp, blk.conv1.out_channels, 3, want_stats=True)
        m1, i1 = ops.bn_stats_from_conv(part, B, blk.conv1.out_channels, H * H)
        m2, i2 = ops.bn_stats(ah)
        print("a err", rel(ah, a), " mean(conv-epilogue) err", rel(m1, a.mean((0, 2, 3))), " mean(bn_stats) err", rel(m2, a.mean((0, 2, 3))))
        print("invstd(conv-epilogue) err", rel(i1, 1 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)), " invstd(bn_stats) err", rel(i2, 1 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)))
        zh = (ah - m1.view(1, -1, 1, 1)) * (i1 * blk.bn1.weight).view(1, -1, 1, 1) + blk.bn1.bias.view(1, -1, 1, 1)
        mism = ((zh > 0).cpu() != (z1 > 0)).sum().item()
        print("sign mismatches z1:", mism, "of", zh.numel(), " max|z1 err|", float((zh.double().cpu() - z1).abs().max()))

run(16, 64, 64, 32, None)
run(16, 256, 256, 4, 'up')
run(16, 64, 128, 16, 'pool')
run(4, 32, 32, 8, None)
