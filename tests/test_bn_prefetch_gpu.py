"""The LDS request buffer of the persistent BatchNorm backward (bn_fused.hip / bf16_bn_fused.hip, round 5) moves WHEN x
is loaded, not what is computed: with the buffer on and off (SIVAE_BN_FUSED_PREFETCH, read once per process — hence
subprocesses) every output is bit-identical, for the one-grid form, the two-half-grids form, two segments, pooled dy and
the sign-mask variants, fp32 and bf16 kernels."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r'''
import hashlib, sys
sys.path.insert(0, %r)
import torch
from sivae_hip import ops
L = ops._lib.load()
g = torch.Generator(device="cuda").manual_seed(11)
h = hashlib.sha256()
def add(*ts):
    for t in ts:
        if t is not None:
            h.update(t.detach().contiguous().cpu().numpy().tobytes())
# (B, C, H, W, nseg): one grid with a plane set per group; two half-grids, several groups; two segments; a small map
for (B, C, H, W, nseg) in [(64, 16, 256, 256, 1), (32, 64, 128, 128, 1), (32, 32, 128, 128, 2), (16, 128, 32, 32, 1)]:
    x = torch.randn(B, C, H, W, device="cuda", generator=g) * 1.3 + 0.2
    dy = torch.randn(B, C, H, W, device="cuda", generator=g)
    res = torch.randn(B, C, H, W, device="cuda", generator=g)
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g)
    xs = x.view(nseg, B // nseg, C, H * W)
    mean = xs.mean((1, 3)).reshape(-1).contiguous()
    invstd = (1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).reshape(-1).contiguous()
    assert L.sivae_bn_bwd_fused_supported(B, C, H, W, B // nseg) == 1
    add(*ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg))
    y, _, mask = ops.bn_apply_act_signmask(x, res, mean, invstd, gamma, beta, 0.2, nseg=nseg)
    add(*ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, dz_sum=True, nseg=nseg))
    add(*ops.bn_bwd(dy, y, x, mean, invstd, gamma, 0.2, want_dz=True, act_mode=1, nseg=nseg))
    dyp = torch.randn(B, C, H // 2, W // 2, device="cuda", generator=g)
    add(*ops.bn_bwd(dyp, y, x, mean, invstd, gamma, 0.2, want_dz=True, act_mode=1, dy_pooled=True, nseg=nseg))
# the bf16 kernel through its own op layer
from sivae_hip import ops16
for (B, C, H, W) in [(64, 64, 128, 128), (32, 128, 64, 64)]:
    x = ops16.from_f32(torch.randn(B, C, H, W, device="cuda", generator=g))
    dy = ops16.from_f32(torch.randn(B, C, H, W, device="cuda", generator=g))
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g)
    xf = ops16.to_f32(x, C)
    mean = xf.mean((0, 2, 3)).contiguous()
    invstd = (1.0 / torch.sqrt(xf.var((0, 2, 3), unbiased=False) + 1e-5)).contiguous()
    assert L.sivae_bf16_bn_bwd_fused_supported(B, C, H, W) == 1
    add(*[t.view(torch.int16) if t.dtype == torch.bfloat16 else t
          for t in ops16.bn_bwd(dy, None, x, mean, invstd, gamma, beta, C, want_dz=True) if isinstance(t, torch.Tensor)])
torch.cuda.synchronize()
ops.bn_fused_check()
print("DIGEST", h.hexdigest())
'''


def _digest(prefetch):
    env = dict(os.environ)
    env["SIVAE_BN_FUSED_PREFETCH"] = prefetch
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    out = subprocess.run([sys.executable, "-c", _CODE % os.path.join(REPO, "soft-intro-vae-pytorch_amd")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST ")]
    assert lines, out.stdout[-2000:]
    return lines[-1].split()[1]


def test_bn_fused_request_buffer_does_not_change_a_bit():
    assert _digest("1") == _digest("0")
