"""CPU: the host-side algebra of the kw-packed RGB-side 5x5 layers of the bf16 mode (functional16._virtual / _unpack_dw).

A 5x5 conv over / into C <= 3 channels is run by the HIP path as a 5-tap (5 rows x 1 column) conv over / into 5C channels
(ops16.im2col_kw5 / fold_kw5 + ks code 51).  The kernels are checked on the GPU (tests/kernel_checks16.py::check_kwpack);
here the permutations that build the virtual weights and re-index the weight gradients are checked with stock torch ops
in fp64: im2col and fold are written out from their definitions in include/sivae_hip.h.
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "soft-intro-vae-pytorch_amd"))


def _im2col_kw5(x, sgn):
    """[B, C, H, W] -> [B, 5C, H, W]:  out[kw*C + c][h][w] = x[c][h][w + sgn*(kw-2)], 0 outside"""
    B, C, H, W = x.shape
    out = torch.zeros(B, 5 * C, H, W, dtype=x.dtype)
    for kw in range(5):
        for w in range(W):
            ws = w + sgn * (kw - 2)
            if 0 <= ws < W:
                out[:, kw * C:(kw + 1) * C, :, w] = x[:, :, :, ws]
    return out


def _fold_kw5(g, bias, C, sgn):
    """[B, 5C, H, W] -> [B, C, H, W]:  out[c][h][w] = bias[c] + sum_kw g[kw*C + c][h][w + sgn*(kw-2)]"""
    B, C5, H, W = g.shape
    out = torch.zeros(B, C, H, W, dtype=g.dtype)
    if bias is not None:
        out += bias.view(1, C, 1, 1)
    for kw in range(5):
        for w in range(W):
            ws = w + sgn * (kw - 2)
            if 0 <= ws < W:
                out[:, :, :, w] += g[:, kw * C:(kw + 1) * C, :, ws]
    return out


@pytest.mark.parametrize("C,Cw,H,W", [(3, 8, 9, 11), (1, 4, 6, 7), (2, 5, 5, 5)])
def test_kwpacked_forms_equal_the_5x5_convs(C, Cw, H, W):
    from sivae_hip import functional16 as SF16
    g = torch.Generator().manual_seed(0)
    B = 2
    # ---- narrow INPUT (the encoder stem, C -> Cw)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cw, C, 5, 5, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, padding=2)
    da = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(da)
    xk = _im2col_kw5(x.detach(), +1)
    yk = F.conv2d(xk, SF16._virtual(w.detach(), "in"), padding=(2, 0))
    assert torch.allclose(yk, y.detach(), atol=1e-12)
    dwk = torch.nn.grad.conv2d_weight(xk, (Cw, 5 * C, 5, 1), da, padding=(2, 0))
    assert torch.allclose(SF16._unpack_dw(dwk, Cw, C, "in"), w.grad, atol=1e-12)
    gk = F.conv2d(da, SF16._virtual(w.detach(), "in_d"), padding=(2, 0))
    assert torch.allclose(_fold_kw5(gk, None, C, -1), x.grad, atol=1e-12)
    # ---- narrow OUTPUT (Decoder.predict, Cw -> C, with bias)
    xf = torch.randn(B, Cw, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    wp = torch.randn(C, Cw, 5, 5, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(C, generator=g, dtype=torch.float64)
    yp = F.conv2d(xf, wp, bias, padding=2)
    dy = torch.randn(yp.shape, generator=g, dtype=torch.float64)
    yp.backward(dy)
    ypk = _fold_kw5(F.conv2d(xf.detach(), SF16._virtual(wp.detach(), "out"), padding=(2, 0)), bias, C, +1)
    assert torch.allclose(ypk, yp.detach(), atol=1e-12)
    dyk = _im2col_kw5(dy, -1)
    dwpk = torch.nn.grad.conv2d_weight(xf.detach(), (5 * C, Cw, 5, 1), dyk, padding=(2, 0))
    assert torch.allclose(SF16._unpack_dw(dwpk, C, Cw, "out"), wp.grad, atol=1e-12)
    dxk = F.conv2d(dyk, SF16._virtual(wp.detach(), "out_d"), padding=(2, 0))
    assert torch.allclose(dxk, xf.grad, atol=1e-12)


def test_kwpack_applies_only_to_5x5_layers_with_at_most_three_narrow_channels():
    from sivae_hip import functional16 as SF16
    assert SF16._kwpack_ok(torch.zeros(64, 3, 5, 5), 3)
    assert SF16._kwpack_ok(torch.zeros(3, 64, 5, 5), 3)
    assert not SF16._kwpack_ok(torch.zeros(64, 4, 5, 5), 4)   # 5 * 4 channels do not fit the 16-channel k-step
    assert not SF16._kwpack_ok(torch.zeros(64, 3, 3, 3), 3)
