"""Kernel-level parity checks (T2): every HIP kernel vs the same op in stock torch on the CPU in fp64.

Each check returns (name, err, tol) where err = max|hip - ref| / (max|ref| + 1e-30). Used by
tests/test_kernels_gpu.py (pytest -m gpu) and by `python tests/kernel_checks.py` which prints the
whole table without stopping at the first failure (one GPU call -> the complete picture).
"""
import ctypes
import math
import sys
import traceback

import torch
import torch.nn.functional as F

import support as _support  # tests/support: kernels that must not live in the product library

DEV = "cuda"


def _err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if a.shape != ref.shape:
        return float("inf")
    if not torch.isfinite(a).all():
        return float("inf")
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def _d(t):
    return t.float().to(DEV).contiguous()


# ------------------------------------------------------------------------------------------------ conv
CONV_SHAPES = [
    # (B, Ci, Co, H, W, ks)
    (2, 64, 128, 32, 32, 3),
    (2, 128, 64, 16, 16, 3),
    (3, 64, 64, 64, 64, 3),
    (4, 160, 136, 8, 8, 3),    # non-multiple channels
    (8, 512, 512, 4, 4, 3),
    (5, 16, 32, 8, 8, 3),      # Co <= 32 config
    (2, 8, 8, 7, 7, 3),        # odd spatial
    (2, 64, 64, 28, 28, 3),
    (2, 64, 128, 32, 32, 1),
    (2, 128, 64, 16, 16, 1),
    (3, 100, 40, 8, 8, 1),
    (3, 72, 200, 12, 20, 1),   # ragged output channels, two output-channel tiles
    (9, 40, 72, 64, 64, 1),    # 144 split-K slices -> 16-row slice reduce with a ragged tail
    (5, 64, 96, 48, 48, 1),    # 45 slices -> 4-row slice reduce with a ragged tail
    (3, 3, 64, 32, 32, 5),     # encoder stem
    (2, 64, 3, 32, 32, 5),     # decoder predict
    (2, 1, 64, 28, 28, 5),
    (2, 64, 1, 28, 28, 5),
    (2, 20, 130, 8, 8, 5),
]


def _conv_ref(x, w, b=None):
    return F.conv2d(x, w, b, stride=1, padding=w.shape[-1] // 2)


WINO_SHAPES = [
    # (B, Ci, Co, H, W, 3): maps the Winograd F(2x2,3x3) kernel accepts (even H >= 8, even W >= 16)
    (2, 64, 128, 32, 32, 3),
    (2, 128, 64, 16, 16, 3),
    (3, 64, 64, 64, 64, 3),
    (2, 64, 64, 28, 28, 3),    # partial tile blocks in both directions (8x16 px geometry)
    (1, 20, 70, 8, 16, 3),     # smallest map, non-multiple channels
    (2, 8, 8, 12, 20, 3),
    (2, 72, 130, 16, 48, 3),   # 4x32 px geometry with a partial block
    (1, 512, 512, 16, 16, 3),  # long K loop
    (4, 64, 128, 8, 8, 3),     # 8x8 maps: two images per tile block
    (3, 40, 70, 8, 8, 3),      # ... odd batch (last block has one image), non-multiple channels
    (8, 64, 128, 4, 4, 3),     # 4x4 maps: four images per tile block
    (6, 24, 40, 4, 4, 3),      # ... partial last block
]


def _pack(ops, w, mode, wino):
    return ops.PackedW(w, mode) if wino else ops.pack_weight(w, mode)


def check_conv_fwd(shape, bias=False, stats=False, wino=False):
    from sivae_hip import ops
    B, Ci, Co, H, W, ks = shape
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Co, Ci, ks, ks, seed=2, scale=1.0 / math.sqrt(Ci * ks * ks))
    b = _rand(Co, seed=3) if bias else None
    ref = _conv_ref(x, w, b)
    wp = _pack(ops, _d(w), 0, wino)
    out = ops.conv2d_fwd(_d(x), wp, Co, ks, bias=_d(b) if bias else None, want_stats=stats)
    res = []
    tag = "wino_" if wino else "conv_"
    if stats:
        y, part = out
        s = part.double().sum(0).cpu()
        res.append((tag + "fwd_stats_sum%s" % (shape,), _err(s[:, 0], ref.sum((0, 2, 3))) , 2e-5))
        res.append((tag + "fwd_stats_sq%s" % (shape,), _err(s[:, 1], (ref * ref).sum((0, 2, 3))), 2e-5))
    else:
        y = out
    # Winograd F(2x2,3x3) in fp32: the +-1 / 0.5 / 0.25 transforms cost about one extra bit pair of rounding
    res.append((tag + "fwd%s%s" % (shape, "+bias" if bias else ""), _err(y, ref), WINO_TOL if wino else 1e-5))
    return res


WINO_TOL = 2e-5


def check_conv_dgrad(shape, wino=False):
    from sivae_hip import ops
    B, Ci, Co, H, W, ks = shape
    x = _rand(B, Ci, H, W, seed=1).requires_grad_()
    w = _rand(Co, Ci, ks, ks, seed=2, scale=1.0 / math.sqrt(Ci * ks * ks))
    dy = _rand(B, Co, H, W, seed=4)
    _conv_ref(x, w).backward(dy)
    wpd = _pack(ops, _d(w), 1, wino)
    dx = ops.conv2d_fwd(_d(dy), wpd, Ci, ks)
    return [(("wino_" if wino else "conv_") + "dgrad%s" % (shape,), _err(dx, x.grad), WINO_TOL if wino else 1e-5)]


def check_conv_wgrad(shape, wino=None):
    """wino: None = whatever ops dispatches to (direct for maps the Winograd kernel does not take),
    True / False force the Winograd-domain / direct kernel"""
    from sivae_hip import ops
    B, Ci, Co, H, W, ks = shape
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Co, Ci, ks, ks, seed=2).requires_grad_()
    dy = _rand(B, Co, H, W, seed=4)
    _conv_ref(x, w).backward(dy)
    saved = ops.WINO_WGRAD
    if wino is not None:
        ops.WINO_WGRAD = wino
    try:
        dw = ops.conv2d_wgrad(_d(x), _d(dy), ks)
    finally:
        ops.WINO_WGRAD = saved
    return [(("wino_" if wino else "conv_") + "wgrad%s" % (shape,), _err(dw, w.grad), WINO_TOL if wino else 1e-5)]


def check_conv_fused(shape, wino=False):
    """prologue BN+LeakyReLU, nearest-2x upsample addressing, accumulate — forward and wgrad"""
    from sivae_hip import ops
    tag, tol = ("wino_", WINO_TOL) if wino else ("conv_", 1e-5)
    B, Ci, Co, H, W, ks = shape
    res = []
    xs = _rand(B, Ci, H // 2, W // 2, seed=5)
    w = _rand(Co, Ci, ks, ks, seed=2, scale=1.0 / math.sqrt(Ci * ks * ks)).requires_grad_()
    mean, invstd = _rand(Ci, seed=6), _rand(Ci, seed=7).abs() + 0.5
    gamma, beta = _rand(Ci, seed=8), _rand(Ci, seed=9)
    dy = _rand(B, Co, H, W, seed=4)
    pro = (_d(mean), _d(invstd), _d(gamma), _d(beta), 0.2)

    def bnact(t):
        v = (t - mean.view(1, -1, 1, 1)) * (invstd * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        return F.leaky_relu(v, 0.2)

    # upsample + prologue
    xin = F.interpolate(bnact(xs), scale_factor=2, mode="nearest")
    ref = _conv_ref(xin, w)
    ref.backward(dy)
    wp = _pack(ops, _d(w.detach()), 0, wino)
    y = ops.conv2d_fwd(_d(xs), wp, Co, ks, pro=pro, upsample=True)
    res.append((tag + "fwd_pro_up%s" % (shape,), _err(y, ref), tol))
    saved = ops.WINO_WGRAD
    ops.WINO_WGRAD = bool(wino)
    try:
        dw = ops.conv2d_wgrad(_d(xs), _d(dy), ks, pro=pro, upsample=True)
    finally:
        ops.WINO_WGRAD = saved
    res.append((tag + "wgrad_pro_up%s" % (shape,), _err(dw, w.grad), tol))
    # prologue only, full-res input, accumulate into an existing tensor
    xf = _rand(B, Ci, H, W, seed=11)
    base = _rand(B, Co, H, W, seed=12)
    ref2 = _conv_ref(bnact(xf), w.detach()) + base
    out = _d(base).clone()
    ops.conv2d_fwd(_d(xf), wp, Co, ks, pro=pro, out=out, accumulate=True)
    res.append((tag + "fwd_pro_acc%s" % (shape,), _err(out, ref2), tol))
    return res


def check_conv5_edge():
    """the merged (channel, kw) 5x5 kernels: predict forward, stem data gradient, both weight gradients"""
    from sivae_hip import ops
    res = []
    for (B, Cb, Cs, H, W) in [(2, 64, 3, 32, 32), (3, 64, 3, 64, 64), (2, 64, 1, 28, 28), (2, 40, 3, 12, 20),
                              (1, 64, 3, 256, 256), (2, 16, 2, 130, 136)]:
        tag = "(%d,%d,%d,%d,%d)" % (B, Cb, Cs, H, W)
        # predict-like: Cb -> Cs
        x = _rand(B, Cb, H, W, seed=1).requires_grad_()
        w = _rand(Cs, Cb, 5, 5, seed=2, scale=1.0 / math.sqrt(Cb * 25)).requires_grad_()
        b = _rand(Cs, seed=3)
        dy = _rand(B, Cs, H, W, seed=4)
        ref = F.conv2d(x, w, b, padding=2)
        ref.backward(dy)
        y = ops.conv5_smallco_fwd(_d(x.detach()), ops.pack5_smallco(_d(w.detach()), 0), Cs, bias=_d(b))
        res.append(("conv5_smallco_fwd" + tag, _err(y, ref), 1e-5))
        dw = ops.conv5_edge_wgrad(_d(x.detach()), _d(dy))
        res.append(("conv5_smallco_wgrad" + tag, _err(dw, w.grad), 1e-5))
        # stem-like: Cs -> Cb
        x2 = _rand(B, Cs, H, W, seed=5).requires_grad_()
        w2 = _rand(Cb, Cs, 5, 5, seed=6, scale=1.0 / math.sqrt(Cs * 25)).requires_grad_()
        dy2 = _rand(B, Cb, H, W, seed=7)
        F.conv2d(x2, w2, padding=2).backward(dy2)
        dx2 = ops.conv5_smallco_fwd(_d(dy2), ops.pack5_smallco(_d(w2.detach()), 1), Cs)
        res.append(("conv5_stem_dgrad" + tag, _err(dx2, x2.grad), 1e-5))
        dw2 = ops.conv5_edge_wgrad(_d(x2.detach()), _d(dy2))
        res.append(("conv5_smallci_wgrad" + tag, _err(dw2, w2.grad), 1e-5))
    return res


def check_linear():
    from sivae_hip import ops
    res = []
    for (B, Ci, Co) in [(16, 8192, 1024), (16, 512, 8192), (7, 100, 30), (512, 2, 256)]:
        x = _rand(B, Ci, seed=1).requires_grad_()
        w = _rand(Co, Ci, seed=2, scale=1.0 / math.sqrt(Ci)).requires_grad_()
        b = _rand(Co, seed=3)
        dy = _rand(B, Co, seed=4)
        ref = F.linear(x, w, b)
        ref.backward(dy)
        wp = ops.pack_weight(_d(w.detach()), 0)
        wpd = ops.pack_weight(_d(w.detach()), 1)
        y = ops.conv2d_fwd(_d(x.detach()).view(B, Ci, 1, 1), wp, Co, 1, bias=_d(b))
        res.append(("linear_fwd(%d,%d,%d)" % (B, Ci, Co), _err(y.view(B, Co), ref), 1e-5))
        dx = ops.conv2d_fwd(_d(dy).view(B, Co, 1, 1), wpd, Ci, 1)
        res.append(("linear_dgrad(%d,%d,%d)" % (B, Ci, Co), _err(dx.view(B, Ci), x.grad), 1e-5))
        dw = ops.conv2d_wgrad(_d(x.detach()).view(B, Ci, 1, 1), _d(dy).view(B, Co, 1, 1), 1)
        res.append(("linear_wgrad(%d,%d,%d)" % (B, Ci, Co), _err(dw.view(Co, Ci), w.grad), 1e-5))
    return res


def check_linear_fast():
    """linear.hip (small-batch split-contraction GEMMs) incl. ragged batch / column / contraction sizes, bias + ReLU"""
    from sivae_hip import ops
    res = []
    for (B, K, N) in [(128, 8192, 512), (16, 8192, 1024), (16, 512, 8192), (128, 256, 8192), (7, 100, 36), (33, 4100, 60),
                      (1, 64, 128), (200, 72, 260), (512, 4096, 256), (512, 128, 4096), (300, 64, 64)]:  # (> 256 rows: chunks)
        assert ops.linear_supported(B, K, N)
        x = _rand(B, K, seed=1).requires_grad_()
        w = _rand(N, K, seed=2, scale=1.0 / math.sqrt(K)).requires_grad_()
        b = _rand(N, seed=3)
        dy = _rand(B, N, seed=4)
        ref = F.linear(x, w, b)
        ref.backward(dy)
        tag = "(%d,%d,%d)" % (B, K, N)
        y = ops.linear_fwd(_d(x.detach()), _d(w.detach()), _d(b))
        res.append(("linear_fast_fwd" + tag, _err(y, ref), 1e-5))
        yr = ops.linear_fwd(_d(x.detach()), _d(w.detach()), _d(b), relu=True)
        res.append(("linear_fast_fwd_relu" + tag, _err(yr, ref.detach().clamp(min=0)), 1e-5))
        res.append(("linear_fast_dgrad" + tag, _err(ops.linear_dgrad(_d(dy), _d(w.detach())), x.grad), 1e-5))
        res.append(("linear_fast_wgrad" + tag, _err(ops.linear_wgrad(_d(dy), _d(x.detach())), w.grad), 1e-5))
    res.append(("linear_fast_unsupported", float(ops.linear_supported(20000, 64, 64) or ops.linear_supported(8, 66, 64)), 0.0))
    return res


# ------------------------------------------------------------------------------------------------ BN
BN_SHAPES = [(4, 64, 32, 32), (8, 512, 4, 4), (3, 7, 7, 7), (2, 128, 64, 64), (16, 256, 1, 1)]


def check_bn(shape, residual):
    from sivae_hip import ops
    B, C, H, W = shape
    res = []
    x = (_rand(B, C, H, W, seed=1) * 2.0 + 0.7).requires_grad_()
    r = _rand(B, C, H, W, seed=2).requires_grad_() if residual else None
    gamma = (_rand(C, seed=3) * 0.5 + 1.0).requires_grad_()
    beta = _rand(C, seed=4).requires_grad_()
    rm, rv = _rand(C, seed=5), _rand(C, seed=6).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    dy = _rand(B, C, H, W, seed=7)
    bn = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    yref = F.leaky_relu(bn + r if residual else bn, 0.2)
    yref.backward(dy)
    tag = "%s%s" % (shape, "+res" if residual else "")

    rm_d, rv_d = _d(rm), _d(rv)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    xd = _d(x.detach())
    mean, invstd = ops.bn_stats(xd, rm_d, rv_d, nbt)
    xm = x.detach().mean((0, 2, 3))
    xv = x.detach().var((0, 2, 3), unbiased=False)
    res.append(("bn_mean%s" % tag, _err(mean, xm), 1e-6))
    res.append(("bn_invstd%s" % tag, _err(invstd, 1.0 / torch.sqrt(xv + 1e-5)), 1e-5))
    res.append(("bn_running_mean%s" % tag, _err(rm_d, rm_ref), 1e-6))
    res.append(("bn_running_var%s" % tag, _err(rv_d, rv_ref), 1e-5))
    res.append(("bn_nbt%s" % tag, 0.0 if int(nbt.item()) == 1 else float("inf"), 0.5))
    y = ops.bn_apply_act(xd, _d(r.detach()) if residual else None, mean, invstd, _d(gamma.detach()),
                         _d(beta.detach()), 0.2)
    res.append(("bn_apply%s" % tag, _err(y, yref), 1e-5))
    dx, dz, dgamma, dbeta = ops.bn_bwd(_d(dy), y, xd, mean, invstd, _d(gamma.detach()), 0.2, want_dz=residual)
    res.append(("bn_dx%s" % tag, _err(dx, x.grad), 2e-5))
    res.append(("bn_dgamma%s" % tag, _err(dgamma, gamma.grad), 2e-5))
    res.append(("bn_dbeta%s" % tag, _err(dbeta, beta.grad), 2e-5))
    if residual:
        res.append(("bn_dres%s" % tag, _err(dz, r.grad), 1e-5))
    else:
        # sign of the activation recomputed from x instead of read from the saved output
        dx2, _, dg2, db2 = ops.bn_bwd(_d(dy), None, xd, mean, invstd, _d(gamma.detach()), 0.2,
                                      beta=_d(beta.detach()), act_mode=2)
        res.append(("bn_dx_recompute%s" % tag, _err(dx2, x.grad), 2e-5))
        res.append(("bn_dgamma_recompute%s" % tag, _err(dg2, gamma.grad), 2e-5))
    return res


def check_bn_from_conv():
    """conv epilogue partial statistics -> finalize == stats kernel on the conv output"""
    from sivae_hip import ops
    res = []
    for shape in [(4, 32, 64, 32, 32, 3), (8, 64, 256, 8, 8, 3), (2, 3, 64, 64, 64, 5)]:
        B, Ci, Co, H, W, ks = shape
        x = _rand(B, Ci, H, W, seed=1)
        w = _rand(Co, Ci, ks, ks, seed=2, scale=1.0 / math.sqrt(Ci * ks * ks))
        wp = ops.pack_weight(_d(w), 0)
        y, part = ops.conv2d_fwd(_d(x), wp, Co, ks, want_stats=True)
        m1, i1 = ops.bn_stats_from_conv(part, B, Co, H * W)
        ref = _conv_ref(x, w)
        res.append(("bn_from_conv_mean%s" % (shape,), _err(m1, ref.mean((0, 2, 3))), 1e-5))
        res.append(("bn_from_conv_invstd%s" % (shape,),
                    _err(i1, 1.0 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)), 1e-5))
    # the two-stage form (>= 2048 partial rows per pass) against the one-stage form on synthetic partial rows: statistics,
    # running buffers and the batch counter, one and two segments (seg_rev), ragged channel counts
    L = ops._lib.load()
    for (rows, nseg, C, rev) in [(4096, 1, 64, False), (2 * 3000, 2, 40, True), (2 * 2048, 2, 300, False), (5000, 1, 7, False)]:
        S = rows // nseg
        g = torch.Generator().manual_seed(rows + C)
        part = torch.rand(rows, C, 2, generator=g)
        part[:, :, 1] = part[:, :, 0] ** 2 + 0.5 + part[:, :, 1]  # (sumsq above sum^2 / n: positive variances)
        pd = part.to(DEV)
        assert L.sivae_bn_stats_from_conv_workspace_bytes(rows, nseg, C) > 0
        outs = []
        for two_stage in (False, True):
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            nbt = torch.zeros((), dtype=torch.int64, device=DEV)
            mean = torch.empty(nseg * C, device=DEV)
            invstd = torch.empty(nseg * C, device=DEV)
            if two_stage:
                ws = ops.workspace(L.sivae_bn_stats_from_conv_workspace_bytes(rows, nseg, C), pd.device)
                ops._lib.call("sivae_bn_stats_from_conv_ws", ops._p(pd), rows, nseg, int(rev), 4, C, 64, 1e-5, 0.1, ops._p(rm),
                              ops._p(rv), ops._p(nbt), ops._p(mean), ops._p(invstd), ops._p(ws), ws.numel(), ops._s())
            else:
                ops._lib.call("sivae_bn_stats_from_conv_seg", ops._p(pd), rows, nseg, int(rev), 4, C, 64, 1e-5, 0.1,
                              ops._p(rm), ops._p(rv), ops._p(nbt), ops._p(mean), ops._p(invstd), ops._s())
            outs.append((mean, invstd, rm, rv, nbt))
        tag = "(%d,%d,%d)" % (rows, nseg, C)
        for n_, i_ in (("mean", 0), ("invstd", 1), ("running_mean", 2), ("running_var", 3)):
            res.append(("bn_from_conv_2stage_%s%s" % (n_, tag), _err(outs[1][i_], outs[0][i_]), 1e-6))
        res.append(("bn_from_conv_2stage_nbt" + tag, float(abs(int(outs[1][4]) - int(outs[0][4]))), 0.0))
        ref_mean = part.view(nseg, S, C, 2)[..., 0].double().sum(1) / (4 * 64)
        res.append(("bn_from_conv_2stage_mean_fp64" + tag, _err(outs[1][0].view(nseg, C), ref_mean), 1e-6))
    return res


# ------------------------------------------------------------------------------------------------ eltwise
def check_eltwise():
    from sivae_hip import ops
    res = []
    for shape in [(2, 8, 16, 16), (3, 5, 7, 7), (2, 4, 28, 28), (2, 3, 4, 4)]:
        x = _rand(*shape, seed=1).requires_grad_()
        ref = F.avg_pool2d(x, 2)
        g = _rand(*ref.shape, seed=2)
        ref.backward(g)
        y = ops.avgpool2_fwd(_d(x.detach()))
        res.append(("avgpool_fwd%s" % (shape,), _err(y, ref), 1e-6))
        dx = ops.avgpool2_bwd(_d(g), shape[2], shape[3])
        res.append(("avgpool_bwd%s" % (shape,), _err(dx, x.grad), 1e-6))
        x2 = _rand(*shape, seed=3).requires_grad_()
        ref2 = F.interpolate(x2, scale_factor=2, mode="nearest")
        g2 = _rand(*ref2.shape, seed=4)
        ref2.backward(g2)
        y2 = ops.upsample2_fwd(_d(x2.detach()))
        res.append(("upsample_fwd%s" % (shape,), _err(y2, ref2), 1e-7))
        dx2 = ops.upsample2_bwd(_d(g2))
        res.append(("upsample_bwd%s" % (shape,), _err(dx2, x2.grad), 1e-6))
    x = _rand(1000, seed=5).requires_grad_()
    ref = F.relu(x)
    g = _rand(1000, seed=6)
    ref.backward(g)
    y = ops.relu_fwd(_d(x.detach()))
    res.append(("relu_fwd", _err(y, ref), 1e-7))
    res.append(("relu_bwd", _err(ops.relu_bwd(_d(g), y), x.grad), 1e-7))
    a, b = _rand(1003, seed=7), _rand(1003, seed=8)
    ad = _d(a)
    ops.add_(ad, _d(b))
    res.append(("add_inplace", _err(ad, a + b), 1e-6))
    return res


# ------------------------------------------------------------------------------------------------ losses
def check_losses():
    from sivae_hip import ops
    res = []
    for (B, Z) in [(8, 128), (128, 512), (5, 2), (3, 33)]:
        y = _rand(B, 2 * Z, seed=1)
        yd = _d(y)
        mu_d, lv_d = yd[:, :Z], yd[:, Z:]
        mu = y[:, :Z].clone().requires_grad_()
        lv = y[:, Z:].clone().requires_grad_()
        eps = _rand(B, Z, seed=2)
        zref = mu + eps * torch.exp(0.5 * lv)
        g = _rand(B, Z, seed=3)
        zref.backward(g)
        z = ops.reparam_fwd(mu_d, lv_d, _d(eps))
        res.append(("reparam_fwd(%d,%d)" % (B, Z), _err(z, zref), 1e-6))
        dmu, dlv = ops.reparam_bwd(_d(g), lv_d, _d(eps))
        res.append(("reparam_dmu(%d,%d)" % (B, Z), _err(dmu, mu.grad), 1e-6))
        res.append(("reparam_dlv(%d,%d)" % (B, Z), _err(dlv, lv.grad), 1e-6))
        for (mu_o, lv_o) in [(0.0, 0.0), (0.3, -0.7)]:
            mu.grad = None
            lv.grad = None
            kl = -0.5 * (1 + lv - lv_o - lv.exp() / math.exp(lv_o) - (mu - mu_o).pow(2) / math.exp(lv_o)).sum(1)
            gk = _rand(B, seed=4)
            kl.backward(gk)
            klh = ops.kl_fwd(lv_d, mu_d, mu_o, lv_o)
            res.append(("kl_fwd(%d,%d,%g)" % (B, Z, mu_o), _err(klh, kl), 1e-6))
            dlv2, dmu2 = ops.kl_bwd(_d(gk), True, 1.0, lv_d, mu_d, mu_o, lv_o)
            res.append(("kl_dlv(%d,%d,%g)" % (B, Z, mu_o), _err(dlv2, lv.grad), 1e-6))
            res.append(("kl_dmu(%d,%d,%g)" % (B, Z, mu_o), _err(dmu2, mu.grad), 1e-6))
    for (B, D) in [(8, 3072), (4, 49152), (3, 101), (2, 196608)]:
        for lt in ("mse", "l1", "bce"):
            if lt == "bce":
                x = torch.rand(B, D, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
                r = torch.rand(B, D, generator=torch.Generator().manual_seed(2), dtype=torch.float64) * 0.98 + 0.01
            else:
                x, r = _rand(B, D, seed=1), _rand(B, D, seed=2)
            x = x.requires_grad_()
            r = r.requires_grad_()
            if lt == "mse":
                el = (r - x) ** 2
            elif lt == "l1":
                el = (r - x).abs()
            else:
                el = F.binary_cross_entropy(r, x, reduction="none")
            rows = el.sum(1)
            g = _rand(B, seed=3)
            rows.backward(g)
            out = ops.recon_rowsum_fwd(_d(x.detach()), _d(r.detach()), lt)
            res.append(("recon_rowsum_%s(%d,%d)" % (lt, B, D), _err(out, rows), 2e-6))
            d_r, d_x = ops.recon_bwd(_d(x.detach()), _d(r.detach()), lt, _d(g), 0, 1.0, True, True)
            res.append(("recon_drecon_%s(%d,%d)" % (lt, B, D), _err(d_r, r.grad), 1e-5))
            res.append(("recon_dx_%s(%d,%d)" % (lt, B, D), _err(d_x, x.grad), 1e-5))
            if D <= 3072:
                eo = ops.recon_elem_fwd(_d(x.detach()), _d(r.detach()), lt)
                res.append(("recon_elem_%s(%d,%d)" % (lt, B, D), _err(eo, el), 1e-6))
    for B in (8, 128, 300):
        L = (_rand(B, seed=1).abs() * 1000).requires_grad_()
        KL = (_rand(B, seed=2).abs() * 50).requires_grad_()
        scale, br, bn = 1.0 / 3072, 1.0, 256.0
        ref = (-2 * scale * (br * L + bn * KL)).exp().mean()
        ref.backward()
        out, e = ops.expelbo_fwd(_d(L.detach()), _d(KL.detach()), scale, br, bn)
        res.append(("expelbo_fwd(%d)" % B, _err(out, ref), 1e-6))
        gout = torch.ones((), dtype=torch.float32, device=DEV)
        dL, dKL = ops.expelbo_bwd(gout, e, scale, br, bn)
        res.append(("expelbo_dL(%d)" % B, _err(dL, L.grad), 1e-5))
        res.append(("expelbo_dKL(%d)" % B, _err(dKL, KL.grad), 1e-5))
        res.append(("vec_sum(%d)" % B, _err(ops.vec_sum(_d(L.detach()), 0.5), 0.5 * L.detach().sum()), 1e-6))
    # the loss assembly (lossE: 4 terms, lossD: 5; reference :583-586, :618-620) through the autograd function: value and
    # the gradient of every term against torch's own 0-dim arithmetic in fp64
    from sivae_hip import functional as SF
    for n in (1, 4, 5, 6):
        vals = [float(v) for v in (_rand(n, seed=7) * 10)]
        ws = [1.0 / 196608 * 0.5 * (i + 1) if i % 2 == 0 else 0.25 for i in range(n)]
        ts = [torch.tensor(v, dtype=torch.float32, device=DEV, requires_grad=(i != 1)) for i, v in enumerate(vals)]
        out = SF.lincomb(ts, ws)
        out.backward(torch.tensor(3.0, device=DEV))
        ref = sum(w * float(torch.tensor(v, dtype=torch.float32)) for w, v in zip(ws, vals))
        res.append(("lincomb(%d)" % n, _err(out.detach().reshape(1), torch.tensor([ref], dtype=torch.float64)), 1e-6))
        gerr = max(abs(float(t.grad) - 3.0 * w) / abs(3.0 * w) for i, (t, w) in enumerate(zip(ts, ws)) if i != 1)
        res.append(("lincomb_grad(%d)" % n, gerr, 1e-6))
        res.append(("lincomb_no_grad_for_detached(%d)" % n, float(n > 1 and ts[1].grad is not None), 0.0))
    return res


def _allclose_err(a, ref, rtol, atol_scale=1e-6):
    """element-wise criterion |a - ref| <= rtol*|ref| + atol (atol = atol_scale * max|ref|): returns the worst
    violation ratio (<= 1 passes) — the strict reading of "1e-4 relative" for tensors with small entries"""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if a.shape != ref.shape or not torch.isfinite(a).all():
        return float("inf")
    atol = atol_scale * float(ref.abs().max())
    return float(((a - ref).abs() / (rtol * ref.abs() + atol + 1e-300)).max())


def check_kl_tensor_prior():
    """calc_kl with TENSOR mu_o / logvar_o (reference :231-251 accepts tensors): broadcast shapes, no host sync"""
    from sivae_hip import ops
    res = []
    B, Z = 6, 40
    mu, lv = _rand(B, Z, seed=1).requires_grad_(), _rand(B, Z, seed=2).requires_grad_()
    gk = _rand(B, seed=4)
    for name, mo, lo in [("0d", _rand(1, seed=5)[0], _rand(1, seed=6)[0]), ("[Z]", _rand(Z, seed=5), _rand(Z, seed=6)),
                         ("[B,Z]", _rand(B, Z, seed=5), _rand(B, Z, seed=6)), ("[B,1]", _rand(B, 1, seed=5), _rand(B, 1, seed=6)),
                         ("[1,Z]/0d", _rand(1, Z, seed=5), torch.tensor(0.25, dtype=torch.float64))]:
        mu.grad = lv.grad = None
        kl = -0.5 * (1 + lv - lo - lv.exp() / torch.exp(lo) - (mu - mo).pow(2) / torch.exp(lo)).sum(1)
        kl.backward(gk)
        klh = ops.kl_fwd_t(_d(lv.detach()), _d(mu.detach()), mo.float().to(DEV), lo.float().to(DEV))
        res.append(("kl_tensor_prior_fwd %s" % name, _err(klh, kl), 1e-6))
        dlv, dmu = ops.kl_bwd_t(_d(gk), True, 1.0, _d(lv.detach()), _d(mu.detach()), mo.float().to(DEV), lo.float().to(DEV))
        res.append(("kl_tensor_prior_dlv %s" % name, _err(dlv, lv.grad), 1e-6))
        res.append(("kl_tensor_prior_dmu %s" % name, _err(dmu, mu.grad), 1e-6))
    return res


def check_adversarial():
    """SURVEY T2 adversarial inputs: |logvar| up to 40 (exp range), a zero-variance BatchNorm channel, a constant
    image, single-sample shards (B = 1)"""
    from sivae_hip import ops
    res = []
    # ---- sampler / KL at the edge of the exp range: element-wise rtol 1e-5 (fp32 exp of +-40 is exact to ~1e-6 rel.)
    B, Z = 4, 64
    g = torch.Generator().manual_seed(11)
    lv = (torch.rand(B, Z, generator=g, dtype=torch.float64) * 80.0 - 40.0)
    lv[0, 0], lv[0, 1] = 40.0, -40.0
    mu = _rand(B, Z, seed=12) * 3.0
    eps = _rand(B, Z, seed=13)
    lv32, mu32, eps32 = lv.float().double(), mu.float().double(), eps.float().double()
    z = ops.reparam_fwd(_d(mu), _d(lv), _d(eps))
    res.append(("adv reparam |logvar|<=40 (allclose 1e-5)", _allclose_err(z, mu32 + eps32 * torch.exp(0.5 * lv32), 1e-5), 1.0))
    kl = ops.kl_fwd(_d(lv), _d(mu), 0.0, 0.0)
    klref = -0.5 * (1 + lv32 - lv32.exp() - mu32.pow(2)).sum(1)
    res.append(("adv kl |logvar|<=40 (allclose 1e-5)", _allclose_err(kl, klref, 1e-5), 1.0))
    dlv, dmu = ops.kl_bwd(_d(torch.ones(B, dtype=torch.float64)), True, 1.0, _d(lv), _d(mu), 0.0, 0.0)
    res.append(("adv kl_dlv |logvar|<=40 (allclose 1e-5)", _allclose_err(dlv, -0.5 * (1 - lv32.exp()), 1e-5), 1.0))
    # exp-ELBO with a huge exponent must underflow to 0, not NaN
    L = torch.tensor([0.0, 1e6, 3e7], dtype=torch.float64)
    KLv = torch.tensor([0.0, 1e4, 1e9], dtype=torch.float64)
    out, e = ops.expelbo_fwd(_d(L), _d(KLv), 1.0 / 3072, 1.0, 1024.0)
    ref = (-2 / 3072 * (L + 1024.0 * KLv)).exp()
    res.append(("adv expelbo underflow", _err(e, ref), 1e-6))
    # ---- zero-variance channel: constant channel (2.5) inside a random tensor; invstd = 1/sqrt(eps), output = beta
    Bc, C, H, W = 4, 8, 16, 16
    x = _rand(Bc, C, H, W, seed=1)
    x[:, 3] = 2.5
    x[:, 5] = 0.0
    gamma, beta = _rand(C, seed=3) * 0.5 + 1.0, _rand(C, seed=4)
    xd = _d(x)
    mean, invstd = ops.bn_stats(xd)
    xv = x.float().double().var((0, 2, 3), unbiased=False)
    res.append(("adv zero-variance bn mean", _err(mean, x.float().double().mean((0, 2, 3))), 1e-6))
    res.append(("adv zero-variance bn invstd", _err(invstd, 1.0 / torch.sqrt(xv + 1e-5)), 1e-5))
    y = ops.bn_apply_act(xd, None, mean, invstd, _d(gamma), _d(beta), 0.2)
    yref = F.leaky_relu(F.batch_norm(x.float().double(), None, None, gamma.float().double(), beta.float().double(),
                                     training=True, eps=1e-5), 0.2)
    res.append(("adv zero-variance bn apply", _err(y, yref), 1e-5))
    dy = _rand(Bc, C, H, W, seed=7)
    dx, _, dg, db = ops.bn_bwd(_d(dy), y, xd, mean, invstd, _d(gamma), 0.2)
    ok = bool(torch.isfinite(dx).all() and torch.isfinite(dg).all() and torch.isfinite(db).all())
    res.append(("adv zero-variance bn bwd finite", 0.0 if ok else float("inf"), 0.5))
    # conv epilogue statistics of an all-zero output channel (zero filter) and of a constant image
    w = _rand(16, 3, 3, 3, seed=2)
    w[7] = 0.0
    img = torch.full((2, 3, 16, 16), 0.5, dtype=torch.float64)
    yc, part = ops.conv2d_fwd(_d(img), ops.PackedW(_d(w), 0), 16, 3, want_stats=True)
    m1, i1 = ops.bn_stats_from_conv(part, 2, 16, 256)
    ref = _conv_ref(img, w.float().double())
    res.append(("adv constant image conv", _err(yc, ref), 1e-5))
    res.append(("adv constant image stats mean", _err(m1, ref.mean((0, 2, 3))), 1e-5))
    res.append(("adv constant image stats invstd", _err(i1, 1.0 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5)), 2e-4))
    # constant image through the reconstruction losses: x == recon -> 0; bce at the clamp (r in {0, 1})
    xi = torch.full((2, 300), 0.5, dtype=torch.float64)
    res.append(("adv recon mse x==r", float(ops.recon_rowsum_fwd(_d(xi), _d(xi), "mse").abs().max()), 0.0))
    r01 = torch.tensor([[0.0, 1.0, 0.0, 1.0]], dtype=torch.float64)
    x01 = torch.tensor([[0.0, 1.0, 1.0, 0.0]], dtype=torch.float64)
    bref = F.binary_cross_entropy(r01, x01, reduction="none").sum(1)
    res.append(("adv bce clamp", _err(ops.recon_rowsum_fwd(_d(x01), _d(r01), "bce"), bref), 1e-6))
    # ---- B = 1 shards
    for shape in [(1, 64, 128, 16, 16, 3), (1, 512, 512, 4, 4, 3), (1, 3, 64, 32, 32, 5), (1, 64, 128, 8, 8, 1)]:
        res += check_conv_fwd(shape, stats=shape[5] == 3, wino=shape[5] == 3)
        res += check_conv_dgrad(shape, wino=shape[5] == 3)
        res += check_conv_wgrad(shape)
    res += check_bn((1, 32, 8, 8), True)
    return res


def check_wino_splitk():
    """split-K Winograd forward / data gradient (deep small maps at small batch): same numbers as the single-pass
    kernel, incl. the fused prologue, accumulate and the per-image BatchNorm partial sums"""
    from sivae_hip import lib, ops
    L = lib.load()
    res = []
    # (planes of 16 / 64 / 256 pixels are summed by the vector reducer, the 32 x 32 map by the scalar one)
    for shape in [(16, 512, 512, 4, 4, 3), (16, 512, 512, 8, 8, 3), (8, 256, 384, 4, 4, 3), (3, 200, 72, 8, 8, 3),
                  (2, 512, 512, 16, 16, 3), (1, 512, 128, 32, 32, 3)]:
        B, Ci, Co, H, W, ks = shape
        S = L.sivae_conv2d_wino_splitk(B, Ci, Co, H, W)
        res.append(("splitk%s slices=%d" % (shape, S), 0.0 if S > 1 else float("inf"), 0.5))
        res += check_conv_fwd(shape, stats=True, wino=True)
        res += check_conv_dgrad(shape, wino=True)
        res += check_conv_fused(shape, wino=True)
    # accumulate through the split path
    B, Ci, Co, H, W = 8, 256, 128, 8, 8
    x, w, y0 = _rand(B, Ci, H, W, seed=1), _rand(Co, Ci, 3, 3, seed=2, scale=0.02), _rand(B, Co, H, W, seed=3)
    y = _d(y0)
    ops.conv2d_fwd(_d(x), ops.PackedW(_d(w), 0), Co, 3, out=y, accumulate=True)
    res.append(("splitk accumulate", _err(y, y0 + _conv_ref(x, w)), WINO_TOL))
    # a launch that already gives every CU more than one block must NOT split
    res.append(("splitk off on large grids", float(L.sivae_conv2d_wino_splitk(128, 512, 512, 16, 16) != 1), 0.0))
    return res


def check_up_dgrad_splitk():
    """split-K form of the upsample-conv data gradient (small shards): same numbers as the single-pass kernel, plain
    and accumulating; shape order as check_conv_up_dgrad: (B, Ci = low-res channels, Co = dy channels, H, W = dy size)"""
    from sivae_hip import lib
    L = lib.load()
    res = []
    for shape in [(4, 512, 512, 32, 32, 3), (2, 256, 512, 16, 32, 3), (16, 512, 512, 32, 32, 3), (3, 200, 136, 16, 32, 3)]:
        B, Ci, Co, H, W, ks = shape
        S = L.sivae_conv2d_wino_up_dgrad_splitk(B, Co, Ci, H // 2, W // 2)
        res.append(("up_dgrad_splitk%s slices=%d" % (shape, S), 0.0 if S > 1 else float("inf"), 0.5))
        res += check_conv_up_dgrad(shape) + check_conv_up_dgrad(shape, True)
    res.append(("up_dgrad_splitk off on large grids",
                float(L.sivae_conv2d_wino_up_dgrad_splitk(128, 512, 512, 16, 32) != 1), 0.0))
    res.append(("up_dgrad_splitk off for N <= 64", float(L.sivae_conv2d_wino_up_dgrad_splitk(2, 512, 64, 16, 32) != 1), 0.0))
    return res


def check_conv1x1_stream():
    """streaming 1x1 kernel: forward / data gradient / accumulate, ragged pixel tiles (HW not a multiple of 128), output
    channel counts that are not multiples of 64, batch sizes that do not fill a block's four pixel tiles"""
    from sivae_hip import lib, ops
    L = lib.load()
    res = [("conv1x1_stream supported", float(L.sivae_conv1x1_stream_supported(2, 64, 128, 1024) != 1
                                              or L.sivae_conv1x1_stream_supported(2, 64, 128, 49) != 0
                                              or L.sivae_conv1x1_stream_supported(2, 63, 128, 64) != 0
                                              or L.sivae_conv1x1_stream_supported(2, 512, 128, 64) != 1
                                              or L.sivae_conv1x1_stream_supported(2, 514, 128, 64) != 0), 0.0)]
    for shape in [(2, 64, 128, 32, 32, 1), (5, 128, 64, 16, 16, 1), (3, 100, 200, 6, 10, 1), (1, 256, 72, 20, 36, 1),
                  (7, 2, 3, 4, 8, 1), (2, 64, 128, 128, 128, 1),
                  # the three wave tiles: 64 x 128 (>= 512 block items), 32 x 128, 32 x 64 (round 5: the shard-size launches)
                  (16, 64, 128, 128, 128, 1), (16, 64, 128, 64, 64, 1), (8, 256, 512, 32, 32, 1), (16, 128, 200, 44, 36, 1),
                  # more than 256 input channels: always the 32-channel tile (Decoder 512 -> 256, the data gradient of 256 -> 512)
                  (4, 512, 256, 32, 32, 1), (32, 512, 256, 32, 32, 1), (3, 384, 100, 12, 12, 1)]:
        res += check_conv_fwd(shape)
        res += check_conv_dgrad(shape)
    # accumulate (the expand conv's data gradient is added onto the conv1 branch's), once per wave tile
    for B, Ci, Co, H, W in [(3, 128, 64, 12, 20), (16, 128, 64, 64, 64), (16, 64, 72, 128, 128)]:
        x, w, y0 = _rand(B, Ci, H, W, seed=1), _rand(Co, Ci, 1, 1, seed=2, scale=0.1), _rand(B, Co, H, W, seed=3)
        y = _d(y0)
        ops.conv2d_fwd(_d(x), ops.pack_weight(_d(w), 0), Co, 1, out=y, accumulate=True)
        res.append(("conv1x1_stream accumulate %s" % ((B, Ci, Co, H, W),), _err(y, y0 + _conv_ref(x, w)), 1e-5))
    return res


WINO4_TOL = 4e-5  # F(4x4,3x3) in fp32: transform coefficients up to 8 and 1/24 (measured 0.5-1.5e-5 per layer)


def check_wino4(shape, accumulate=False, stats=False, mode=0, b6=False):
    """F(4x4,3x3) kernel (conv_wino4.hip; b6: conv_wino4_b6.hip, six bf16 MFMAs per fp32 product — SAME tolerances) vs the
    fp64 convolution: forward (mode 0) / data gradient (mode 1)"""
    from sivae_hip import lib, ops
    B, Ci, Co, H, W = shape
    L = lib.load()
    # (2 / 3 / 4: 16 x 16 / 8 x 8 / 4 x 4 maps, a work item is a grid of 2 / 8 / 32 whole images)
    assert L.sivae_conv2d_wino4_supported(H, W) in ((1, 2) if b6 else (1, 2, 3, 4))
    assert B % L.sivae_conv2d_wino4_images_per_item(H, W) == 0
    x = _rand(B, Ci, H, W, seed=1)
    res = []
    if mode == 0:
        w = _rand(Co, Ci, 3, 3, seed=2, scale=1.0 / math.sqrt(Ci * 9))
        ref = _conv_ref(x, w)
    else:  # data gradient of a conv with weight [Ci_in_of_grad = Ci][Co]...: x plays dy, result has Co channels
        w = _rand(Ci, Co, 3, 3, seed=2, scale=1.0 / math.sqrt(Ci * 9))
        xin = _rand(B, Co, H, W, seed=5).requires_grad_()
        _conv_ref(xin, w).backward(x)
        ref = xin.grad
    wp = ops.PackedW(_d(w), mode)
    up = wp.wino4_b6() if b6 else wp.wino4()
    y0 = _rand(B, Co, H, W, seed=7) if accumulate else None
    y = _d(y0) if accumulate else torch.empty((B, Co, H, W), dtype=torch.float32, device=DEV)
    part = (torch.empty((L.sivae_conv2d_wino4_num_px_tiles(B, H, W), Co, 2), dtype=torch.float32, device=DEV)
            if stats else None)
    xd = _d(x)  # (named: a temporary would be freed — and its block reused — before the launch reads it)
    if b6:
        lib.call("sivae_conv2d_wino4_b6_fwd", ops._p(xd), ops._p(up), ops._p(y), None, None, None, None, 1.0, ops._p(part),
                 B, Ci, Co, H, W, int(accumulate), 0, ops._s())
    else:
        lib.call("sivae_conv2d_wino4_fwd", ops._p(xd), ops._p(up), ops._p(y), ops._p(part), B, Ci, Co, H, W,
                 int(accumulate), ops._s())
    torch.cuda.synchronize()
    if accumulate:
        ref = ref + y0
    tag = "wino4%s_%s%s%s" % ("b6" if b6 else "", "fwd" if mode == 0 else "dgrad", "_acc" if accumulate else "", shape)
    res.append((tag, _err(y, ref), WINO4_TOL))
    if stats:
        s = part.double().sum(0).cpu()
        res.append((tag + "_stats_sum", _err(s[:, 0], ref.sum((0, 2, 3))), 4e-5))
        res.append((tag + "_stats_sq", _err(s[:, 1], (ref * ref).sum((0, 2, 3))), 4e-5))
        # rows are in image order (segmented BatchNorm statistics rely on it)
        nrow = part.shape[0]
        if nrow >= B:
            per_img = part.double().view(B, -1, Co, 2).sum(1).cpu()
            res.append((tag + "_stats_rows", _err(per_img[..., 0], ref.sum((2, 3))), 4e-5))
        else:  # 16 x 16 maps: one row per image pair
            ref_rows = ref.view(nrow, B // nrow, Co, H, W).sum((1, 3, 4))
            res.append((tag + "_stats_rows", _err(part.double().cpu()[..., 0], ref_rows), 4e-5))
    return res


def check_wino4_splitk(shape, pro=False, nseg=1, accumulate=False, b6=False):
    """split-K form of the F(4x4,3x3) kernel (few work items): output, accumulate, per-image statistics, prologue/segments"""
    from sivae_hip import lib, ops
    B, Ci, Co, H, W = shape
    L = lib.load()
    S = L.sivae_conv2d_wino4_splitk(B, Ci, Co, H, W)
    res = [("wino4_splitk slices%s" % (shape,), float(S <= 1), 0.0)]  # (the shapes below are chosen to split)
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Co, Ci, 3, 3, seed=2, scale=1.0 / math.sqrt(Ci * 9))
    xin = x
    pm = pi = pg = pb = None
    if pro:
        mean = _rand(nseg, Ci, seed=3, scale=0.3)
        invstd = (_rand(nseg, Ci, seed=4).abs() + 0.5)
        gamma = _rand(Ci, seed=5) + 1.0
        beta = _rand(Ci, seed=6, scale=0.2)
        Bs = B // nseg
        xs = []
        for g in range(nseg):
            v = (x[g * Bs:(g + 1) * Bs] - mean[g].view(1, -1, 1, 1)) * (invstd[g] * gamma).view(1, -1, 1, 1) \
                + beta.view(1, -1, 1, 1)
            xs.append(torch.where(v > 0, v, 0.2 * v))
        xin = torch.cat(xs)
        pm, pi, pg, pb = _d(mean.reshape(-1)), _d(invstd.reshape(-1)), _d(gamma), _d(beta)
    ref = _conv_ref(xin, w)
    y0 = _rand(B, Co, H, W, seed=7) if accumulate else None
    y = _d(y0) if accumulate else torch.empty((B, Co, H, W), dtype=torch.float32, device=DEV)
    if accumulate:
        ref = ref + y0
    part = torch.empty((B, Co, 2), dtype=torch.float32, device=DEV)
    wp = ops.PackedW(_d(w), 0)
    xd, up = _d(x), (wp.wino4_b6() if b6 else wp.wino4())
    ws = ops.workspace(L.sivae_conv2d_wino4_splitk_workspace_bytes(B, Ci, Co, H, W), xd.device)
    lib.call("sivae_conv2d_wino4_b6_fwd_splitk" if b6 else "sivae_conv2d_wino4_fwd_splitk", ops._p(xd), ops._p(up), ops._p(y), ops._p(pm), ops._p(pi), ops._p(pg),
             ops._p(pb), 0.2, ops._p(part), B, Ci, Co, H, W, int(accumulate), (B // nseg) if nseg > 1 else 0, ops._p(ws),
             ws.numel(), ops._s())
    torch.cuda.synchronize()
    tag = "wino4%s_splitk%s%s%s%s" % ("b6" if b6 else "", "_pro" if pro else "", "_seg%d" % nseg if nseg > 1 else "", "_acc" if accumulate else "",
                                   shape)
    res.append((tag, _err(y, ref), WINO4_TOL))
    res.append((tag + "_stats_rows", _err(part.double().cpu()[..., 0], ref.sum((2, 3))), 4e-5))
    res.append((tag + "_stats_sq", _err(part.double().cpu()[..., 1], (ref * ref).sum((2, 3))), 4e-5))
    return res


def check_wino4_pro(shape, nseg=1, b6=False):
    """F(4x4,3x3) with the fused BatchNorm + LeakyReLU prologue (and per-segment statistics) vs fp64"""
    from sivae_hip import ops
    B, Ci, Co, H, W = shape
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Co, Ci, 3, 3, seed=2, scale=1.0 / math.sqrt(Ci * 9))
    mean = _rand(nseg, Ci, seed=3, scale=0.3)
    invstd = (_rand(nseg, Ci, seed=4).abs() + 0.5)
    gamma = _rand(Ci, seed=5) + 1.0
    beta = _rand(Ci, seed=6, scale=0.2)
    Bs = B // nseg
    xs = []
    for g in range(nseg):
        v = (x[g * Bs:(g + 1) * Bs] - mean[g].view(1, -1, 1, 1)) * (invstd[g] * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        xs.append(torch.where(v > 0, v, 0.2 * v))
    ref = _conv_ref(torch.cat(xs), w)
    from sivae_hip import lib
    L = lib.load()
    wp = ops.PackedW(_d(w), 0)
    y = torch.empty((B, Co, H, W), dtype=torch.float32, device=DEV)
    part = torch.empty((L.sivae_conv2d_wino4_num_px_tiles(B, H, W), Co, 2), dtype=torch.float32, device=DEV)
    pm, pi, pg, pb = _d(mean.reshape(-1)), _d(invstd.reshape(-1)), _d(gamma), _d(beta)
    xd, up = _d(x), (wp.wino4_b6() if b6 else wp.wino4())  # (named: a temporary would be freed before the launch reads it)
    # (straight through the C ABI: ops.conv2d_fwd only picks this kernel from one work item per CU up)
    lib.call("sivae_conv2d_wino4_b6_fwd" if b6 else "sivae_conv2d_wino4_fwd_pro", ops._p(xd), ops._p(up), ops._p(y), ops._p(pm), ops._p(pi), ops._p(pg),
             ops._p(pb), 0.2, ops._p(part), B, Ci, Co, H, W, 0, (B // nseg) if nseg > 1 else 0, ops._s())
    torch.cuda.synchronize()
    tag = "wino4%s_pro%s%s" % ("b6" if b6 else "", "_seg%d" % nseg if nseg > 1 else "", shape)
    s_ = part.double().sum(0).cpu()
    return [(tag, _err(y, ref), WINO4_TOL), (tag + "_stats_sum", _err(s_[:, 0], ref.sum((0, 2, 3))), 4e-5)]


def check_wino4_b6_pack():
    """the pre-split operand of conv_wino4_b6.hip: the three bf16 pieces of every entry add up to the fp32 U of
    sivae_pack_wino4_weight EXACTLY (the split is by truncation), in the MFMA-ready block order, both modes, ragged channels"""
    from sivae_hip import ops
    res = []
    for (Co, Ci, mode) in [(64, 64, 0), (72, 100, 0), (40, 130, 1), (128, 32, 1)]:
        w = _d(_rand(Co, Ci, 3, 3, seed=Co + Ci))
        wp = ops.PackedW(w, mode)
        kdim, ndim = (Ci, Co) if mode == 0 else (Co, Ci)
        kpad, npad = (kdim + 31) // 32 * 32, (ndim + 63) // 64 * 64
        u32 = wp.wino4().view(6, kpad, npad, 6).cpu()          # [j][k][n][i]
        raw = wp.wino4_b6().cpu().view(torch.int16)             # bf16 bit patterns
        blk = raw.view(6, kpad // 16, npad // 32, 6, 3, 64, 8)   # [j][step][sub][i][piece][lane][e]
        f = (blk.to(torch.int32) << 16).view(torch.float32).double().sum(4)  # pieces added (exactly representable sums)
        # lane = (n & 31) + 32 kg, k = 16 step + 8 kg + e
        f = f.view(6, kpad // 16, npad // 32, 6, 2, 32, 8)       # [j][step][sub][i][kg][n31][e]
        f = f.permute(0, 1, 4, 6, 2, 5, 3).reshape(6, kpad, npad, 6)  # [j][step,kg,e -> k][sub,n31 -> n][i]
        res.append(("wino4b6_pack_exact(Co%d,Ci%d,mode%d)" % (Co, Ci, mode), float((f - u32.double()).abs().max()), 0.0))
    return res


def check_wino4_wgrad(shape, pro=False, nseg=1):
    """F(4x4,3x3) weight gradient (conv_wino4_wgrad.hip) vs the fp64 one, straight through the C ABI: plain, with the fused
    BatchNorm + LeakyReLU prologue, and with per-segment statistics"""
    from sivae_hip import lib, ops
    B, Ci, Co, H, W = shape
    L = lib.load()
    assert L.sivae_conv2d_wino4_wgrad_supported(H, W) == 1
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Co, Ci, 3, 3, seed=2).requires_grad_()
    dy = _rand(B, Co, H, W, seed=4)
    xin = x
    pm = pi = pg = pb = None
    if pro:
        mean = _rand(nseg, Ci, seed=3, scale=0.3)
        invstd = (_rand(nseg, Ci, seed=5).abs() + 0.5)
        gamma = _rand(Ci, seed=6) + 1.0
        beta = _rand(Ci, seed=7, scale=0.2)
        Bs = B // nseg
        xs = []
        for g in range(nseg):
            v = (x[g * Bs:(g + 1) * Bs] - mean[g].view(1, -1, 1, 1)) * (invstd[g] * gamma).view(1, -1, 1, 1) \
                + beta.view(1, -1, 1, 1)
            xs.append(torch.where(v > 0, v, 0.2 * v))
        xin = torch.cat(xs)
        pm, pi, pg, pb = _d(mean.reshape(-1)), _d(invstd.reshape(-1)), _d(gamma), _d(beta)
    _conv_ref(xin, w).backward(dy)
    xd, dyd = _d(x), _d(dy)
    ws = ops.workspace(L.sivae_conv2d_wino4_wgrad_workspace_bytes(B, Ci, Co, H, W), xd.device)
    dw = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=DEV)
    lib.call("sivae_conv2d_wino4_wgrad", ops._p(xd), ops._p(dyd), ops._p(dw), ops._p(pm), ops._p(pi), ops._p(pg),
             ops._p(pb), 0.2, B, Ci, Co, H, W, (B // nseg) if nseg > 1 else 0, ops._p(ws), ws.numel(), ops._s())
    torch.cuda.synchronize()
    tag = "wino4_wgrad%s%s%s" % ("_pro" if pro else "", "_seg%d" % nseg if nseg > 1 else "", shape)
    return [(tag, _err(dw, w.grad), WINO4_TOL)]


def check_conv5_k75():
    """merged-contraction 5x5 kernel (<= 3 -> <= 64 channels): forward with BatchNorm partials / bias, ragged tiles,
    1-3 input channels, fewer than 64 outputs; and as the data gradient of a 64 -> 3 conv (flipped pack)"""
    from sivae_hip import lib
    L = lib.load()
    res = [("conv5_k75 supported", float(L.sivae_conv5_k75_supported(3, 64) != 1 or L.sivae_conv5_k75_supported(4, 64) != 0
                                         or L.sivae_conv5_k75_supported(3, 65) != 0), 0.0)]
    for shape in [(3, 3, 64, 32, 32, 5), (2, 1, 64, 28, 28, 5), (2, 3, 48, 20, 40, 5), (2, 2, 64, 9, 70, 5),
                  (5, 3, 64, 64, 64, 5)]:
        res += check_conv_fwd(shape, stats=True, wino=True)   # (wino=True: a PackedW, which is what selects the kernel)
    res += check_conv_fwd((2, 3, 64, 24, 40, 5), bias=True, wino=True)
    for shape in [(2, 64, 3, 32, 32, 5), (3, 40, 3, 20, 40, 5), (2, 64, 1, 28, 28, 5)]:
        res += check_conv_dgrad(shape, wino=True)
    return res


def check_randn():
    from sivae_hip import ops
    a = ops.randn((1 << 20,), 1234, 0, torch.device(DEV))
    b = ops.randn((1 << 20,), 1234, 0, torch.device(DEV))
    c = ops.randn((1 << 20,), 1234, 1 << 18, torch.device(DEV))
    res = [("randn_repro", float((a - b).abs().max()), 1e-12),
           ("randn_mean", abs(float(a.mean())), 5e-3),
           ("randn_std", abs(float(a.std()) - 1.0), 5e-3),
           ("randn_kurt", abs(float((a ** 4).mean()) - 3.0), 5e-2),
           ("randn_stream_indep", abs(float((a * c).mean())), 5e-3)]
    return res


def check_adam():
    from sivae_hip import ops
    n = 10007
    p0, g1, g2 = _rand(n, seed=1).float(), _rand(n, seed=2).float(), _rand(n, seed=3).float()
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=2e-4)
    pd = p0.to(DEV).clone()
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step, g in enumerate((g1, g2, g1), start=1):
        pr.grad = g.clone()
        opt.step()
        ops.adam_step(pd, g.to(DEV), m, v, step, 2e-4)
    # (the update is ~1e-3 of |p|, so compare p itself: fp32 rounding of p dominates the update error)
    return [("adam_3steps", _err(pd, pr.detach()), 1e-6),
            ("adam_3steps_update", _err(pd - p0.to(DEV), pr.detach() - p0), 2e-3)]


def check_bn_apply_resup():
    """BN-apply + residual read through nearest-2x upsample addressing + LeakyReLU"""
    from sivae_hip import ops
    res = []
    for (B, C, H, W) in [(2, 8, 8, 8), (3, 5, 16, 32), (1, 64, 6, 12)]:
        x, r = _rand(B, C, H, W, seed=1), _rand(B, C, H // 2, W // 2, seed=2)
        mean, invstd = _rand(C, seed=6), _rand(C, seed=7).abs() + 0.5
        gamma, beta = _rand(C, seed=8), _rand(C, seed=9)
        v = (x - mean.view(1, -1, 1, 1)) * (invstd * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        ref = F.leaky_relu(v + F.interpolate(r, scale_factor=2, mode="nearest"), 0.2)
        y = ops.bn_apply_act(_d(x), _d(r), _d(mean), _d(invstd), _d(gamma), _d(beta), 0.2, res_up=True)
        res.append(("bn_apply_resup(%d,%d,%d,%d)" % (B, C, H, W), _err(y, ref), 1e-6))
    return res


def check_conv_up(shape, pro=False, stats=False):
    """conv3x3(Upsample(2,'nearest')(x)) on the low-resolution tensor: phase-decomposed F(2x2,2x2) kernel"""
    from sivae_hip import ops
    B, Ci, Co, H, W, ks = shape  # H, W = output size
    xs = _rand(B, Ci, H // 2, W // 2, seed=5)
    w = _rand(Co, Ci, 3, 3, seed=2, scale=1.0 / math.sqrt(Ci * 9))
    xin = xs
    p = None
    if pro:
        mean, invstd = _rand(Ci, seed=6), _rand(Ci, seed=7).abs() + 0.5
        gamma, beta = _rand(Ci, seed=8), _rand(Ci, seed=9)
        p = (_d(mean), _d(invstd), _d(gamma), _d(beta), 0.2)
        xin = F.leaky_relu((xs - mean.view(1, -1, 1, 1)) * (invstd * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1), 0.2)
    ref = _conv_ref(F.interpolate(xin, scale_factor=2, mode="nearest"), w)
    assert ops.WINO_UP and ops._lib.load().sivae_conv2d_wino_up_supported(H, W) == 1
    out = ops.conv2d_fwd(_d(xs), ops.PackedW(_d(w), 0), Co, 3, pro=p, upsample=True, want_stats=stats)
    res = []
    tag = "wino_up%s%s" % ("_pro" if pro else "", shape)
    if stats:
        y, part = out
        sm = part.double().sum(0).cpu()
        res.append((tag + "_stats_sum", _err(sm[:, 0], ref.sum((0, 2, 3))), 2e-5))
        res.append((tag + "_stats_sq", _err(sm[:, 1], (ref * ref).sum((0, 2, 3))), 2e-5))
    else:
        y = out
    res.append((tag, _err(y, ref), WINO_TOL))
    return res


def check_conv_up_dgrad(shape, accumulate=False):
    """d/dx of conv3x3(Upsample(2,'nearest')(x)) with respect to the low-resolution x"""
    from sivae_hip import ops
    B, Ci, Co, H, W, ks = shape  # H, W = output (dy) size
    xs = _rand(B, Ci, H // 2, W // 2, seed=5).requires_grad_()
    w = _rand(Co, Ci, 3, 3, seed=2, scale=1.0 / math.sqrt(Ci * 9))
    dy = _rand(B, Co, H, W, seed=4)
    _conv_ref(F.interpolate(xs, scale_factor=2, mode="nearest"), w).backward(dy)
    ref = xs.grad
    base = _rand(B, Ci, H // 2, W // 2, seed=12)
    out = _d(base).clone() if accumulate else None
    dx = ops.conv2d_up_dgrad(_d(dy), ops.PackedW(_d(w), 0), Ci, out=out, accumulate=accumulate)
    if accumulate:
        ref = ref + base
    return [("wino_up_dgrad%s%s" % ("_acc" if accumulate else "", shape), _err(dx, ref), WINO_TOL)]


def check_wino4_dgrad_pool(shape, accumulate=False):
    """the same data gradient in one F(4x4,3x3) pass with the 2x2 block sum folded into the output transform
    (sivae_conv2d_wino4_dgrad_pool, straight through the C ABI) vs fp64"""
    from sivae_hip import lib, ops
    B, N, C, H, W = shape  # x: [B, N, H/2, W/2] -> conv N -> C at H x W; dy: [B, C, H, W]
    xs = _rand(B, N, H // 2, W // 2, seed=5).requires_grad_()
    w = _rand(C, N, 3, 3, seed=2, scale=1.0 / math.sqrt(N * 9))
    dy = _rand(B, C, H, W, seed=4)
    _conv_ref(F.interpolate(xs, scale_factor=2, mode="nearest"), w).backward(dy)
    ref = xs.grad
    base = _rand(B, N, H // 2, W // 2, seed=12)
    dx = _d(base).clone() if accumulate else torch.empty((B, N, H // 2, W // 2), dtype=torch.float32, device=DEV)
    wp1 = ops.PackedW(_d(w), 1)
    dyd, up = _d(dy), wp1.wino4()
    lib.call("sivae_conv2d_wino4_dgrad_pool", ops._p(dyd), ops._p(up), ops._p(dx), B, C, N, H, W, int(accumulate), ops._s())
    torch.cuda.synchronize()
    if accumulate:
        ref = ref + base
    return [("wino4_dgrad_pool%s%s" % ("_acc" if accumulate else "", shape), _err(dx, ref), WINO4_TOL)]


def check_space_to_depth():
    from sivae_hip import ops
    x = _rand(3, 5, 12, 20, seed=1)
    out = ops.space_to_depth2(_d(x)).cpu()
    ref = torch.stack([x[:, :, p::2, q::2] for p in (0, 1) for q in (0, 1)], dim=1).float()
    return [("space_to_depth2", float((out - ref).abs().max()), 0.0)]


def check_bn_bwd_dzsum():
    """act_mode-1 BN backward returning the residual-branch gradient as 2x2 block sums"""
    from sivae_hip import ops
    res = []
    for (B, C, H, W) in [(3, 8, 8, 16), (2, 5, 12, 8), (2, 16, 32, 32)]:
        x, r = _rand(B, C, H, W, seed=1), _rand(B, C, H, W, seed=2)
        gamma, beta = _rand(C, seed=8).abs() + 0.5, _rand(C, seed=9)
        dy = _rand(B, C, H, W, seed=4)
        xt, rt = x.clone().requires_grad_(), r.clone().requires_grad_()
        gt, bt = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        y = F.leaky_relu(F.batch_norm(xt, None, None, gt, bt, True, 0.1, 1e-5) + rt, 0.2)
        y.backward(dy)
        mean = x.mean((0, 2, 3))
        invstd = 1.0 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)
        dx, dzh, dg, db = ops.bn_bwd_dzsum(_d(dy), _d(y.detach()), _d(x), _d(mean), _d(invstd), _d(gamma), 0.2)
        ref_dzh = 4.0 * F.avg_pool2d(rt.grad, 2)
        tag = "(%d,%d,%d,%d)" % (B, C, H, W)
        res.append(("bn_bwd_dzsum_dx" + tag, _err(dx, xt.grad), 2e-5))
        res.append(("bn_bwd_dzsum_dzh" + tag, _err(dzh, ref_dzh), 1e-5))
        res.append(("bn_bwd_dzsum_dgamma" + tag, _err(dg, gt.grad), 2e-5))
        res.append(("bn_bwd_dzsum_dbeta" + tag, _err(db, bt.grad), 2e-5))
    return res


def check_bn_signmask():
    """the 1-bit LeakyReLU sign mask path of 'LeakyReLU(BN(c) + res)': outputs and mask of the three apply flavours
    (plain / half-resolution residual / + AvgPool2d, with and without the full-resolution output), and the three
    backward flavours (dz, 2x2 block-summed dz, pooled dy) against the saved-output kernels (must agree bit for bit:
    same arithmetic, only the source of the sign differs)"""
    from sivae_hip import ops
    res_ = []
    for (B, C, H, W) in [(3, 8, 8, 16), (2, 5, 6, 24), (2, 16, 32, 32), (1, 3, 2, 8)]:
        tag = "(%d,%d,%d,%d)" % (B, C, H, W)
        x = _d(_rand(B, C, H, W, seed=1))
        res = _d(_rand(B, C, H, W, seed=2))
        resh = _d(_rand(B, C, H // 2, W // 2, seed=3))
        gamma, beta = _d(_rand(C, seed=8).abs() + 0.5), _d(_rand(C, seed=9))
        dy = _d(_rand(B, C, H, W, seed=4))
        dyh = _d(_rand(B, C, H // 2, W // 2, seed=5))
        mean, invstd = ops.bn_stats(x)
        assert ops.bn_signmask_supported(x)

        def bits(mask):
            m = mask.cpu().numpy()
            import numpy as np
            return torch.from_numpy(np.unpackbits(m, bitorder="little")[: B * C * H * W].astype("float32")).view(B, C, H, W)

        # plain
        y_ref = ops.bn_apply_act(x, res, mean, invstd, gamma, beta, 0.2)
        y, yp, mask = ops.bn_apply_act_signmask(x, res, mean, invstd, gamma, beta, 0.2)
        res_.append(("signmask_apply" + tag, float((y - y_ref).abs().max()), 0.0))
        res_.append(("signmask_bits" + tag, float((bits(mask) - (y_ref > 0).float().cpu()).abs().max()), 0.0))
        dx_ref, dz_ref, dg_ref, db_ref = ops.bn_bwd(dy, y_ref, x, mean, invstd, gamma, 0.2, want_dz=True, act_mode=1)
        dx, dz, dg, db = ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2)
        for n_, a_, b_ in (("dx", dx, dx_ref), ("dz", dz, dz_ref), ("dgamma", dg, dg_ref), ("dbeta", db, db_ref)):
            res_.append(("signmask_bwd_%s%s" % (n_, tag), float((a_ - b_).abs().max()), 0.0))
        # pooled dy
        dx_ref, dz_ref, dg_ref, db_ref = ops.bn_bwd(dyh, y_ref, x, mean, invstd, gamma, 0.2, want_dz=True, act_mode=1,
                                                    dy_pooled=True)
        dx, dz, dg, db = ops.bn_bwd_signmask(dyh, mask, x, mean, invstd, gamma, 0.2, dy_pooled=True)
        for n_, a_, b_ in (("dx", dx, dx_ref), ("dz", dz, dz_ref), ("dgamma", dg, dg_ref), ("dbeta", db, db_ref)):
            res_.append(("signmask_bwd_pooled_%s%s" % (n_, tag), float((a_ - b_).abs().max()), 0.0))
        # + AvgPool2d, with and without the full-resolution output
        for want_full in (True, False):
            y2, yp2, mask2 = ops.bn_apply_act_signmask(x, res, mean, invstd, gamma, beta, 0.2, pool=True,
                                                       want_full=want_full)
            y_p, yp_p = ops.bn_apply_act_pool(x, res, mean, invstd, gamma, beta, 0.2)
            res_.append(("signmask_pool%d_yp%s" % (want_full, tag), float((yp2 - yp_p).abs().max()), 0.0))
            res_.append(("signmask_pool%d_bits%s" % (want_full, tag),
                         float((bits(mask2) - (y_ref > 0).float().cpu()).abs().max()), 0.0))
            if want_full:
                res_.append(("signmask_pool_y" + tag, float((y2 - y_p).abs().max()), 0.0))
        # half-resolution residual + block-summed dz
        y_ref = ops.bn_apply_act(x, resh, mean, invstd, gamma, beta, 0.2, res_up=True)
        y, _, mask = ops.bn_apply_act_signmask(x, resh, mean, invstd, gamma, beta, 0.2, res_up=True)
        res_.append(("signmask_resup" + tag, float((y - y_ref).abs().max()), 0.0))
        res_.append(("signmask_resup_bits" + tag, float((bits(mask) - (y_ref > 0).float().cpu()).abs().max()), 0.0))
        dx_ref, dzh_ref, dg_ref, db_ref = ops.bn_bwd_dzsum(dy, y_ref, x, mean, invstd, gamma, 0.2)
        dx, dzh, dg, db = ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, dz_sum=True)
        for n_, a_, b_ in (("dx", dx, dx_ref), ("dzh", dzh, dzh_ref), ("dgamma", dg, dg_ref), ("dbeta", db, db_ref)):
            res_.append(("signmask_bwd_dzsum_%s%s" % (n_, tag), float((a_ - b_).abs().max()), 0.0))
    return res_


def check_bn_apply_pool():
    """BN-apply (+ residual) + LeakyReLU + AvgPool2d(2) in one pass"""
    from sivae_hip import ops
    res = []
    for (B, C, H, W), has_res in [((2, 8, 8, 8), True), ((3, 5, 12, 16), False), ((1, 64, 6, 12), True)]:
        x, r = _rand(B, C, H, W, seed=1), _rand(B, C, H, W, seed=2)
        mean, invstd = _rand(C, seed=6), _rand(C, seed=7).abs() + 0.5
        gamma, beta = _rand(C, seed=8), _rand(C, seed=9)
        v = (x - mean.view(1, -1, 1, 1)) * (invstd * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        ref = F.leaky_relu(v + r if has_res else v, 0.2)
        y, yp = ops.bn_apply_act_pool(_d(x), _d(r) if has_res else None, _d(mean), _d(invstd), _d(gamma), _d(beta), 0.2)
        res.append(("bn_apply_pool_full(%d,%d,%d,%d)" % (B, C, H, W), _err(y, ref), 1e-6))
        res.append(("bn_apply_pool_pooled(%d,%d,%d,%d)" % (B, C, H, W), _err(yp, F.avg_pool2d(ref, 2)), 1e-6))
        _, yp2 = ops.bn_apply_act_pool(_d(x), None, _d(mean), _d(invstd), _d(gamma), _d(beta), 0.2, want_full=False)
        res.append(("bn_apply_pool_only(%d,%d,%d,%d)" % (B, C, H, W), _err(yp2, F.avg_pool2d(F.leaky_relu(v, 0.2), 2)), 1e-6))
    return res


def check_bn_bwd_pooled():
    """BN(+LeakyReLU) backward with dy given as the gradient of AvgPool2d(2)(y) at half resolution"""
    from sivae_hip import ops
    res = []
    for (B, C, H, W), mode in [((3, 8, 8, 16), 1), ((2, 5, 12, 8), 2), ((2, 16, 32, 32), 1)]:
        x = _rand(B, C, H, W, seed=1)
        r = _rand(B, C, H, W, seed=2)
        gamma, beta = _rand(C, seed=8).abs() + 0.5, _rand(C, seed=9)
        dyh = _rand(B, C, H // 2, W // 2, seed=4)
        xt = x.clone().requires_grad_()
        gt, bt = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        rt = r.clone().requires_grad_()
        z = F.batch_norm(xt, None, None, gt, bt, True, 0.1, 1e-5)
        y = F.leaky_relu(z + rt if mode == 1 else z, 0.2)
        F.avg_pool2d(y, 2).backward(dyh)
        mean = x.mean((0, 2, 3))
        invstd = 1.0 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)
        if mode == 1:
            dx, dz, dg, db = ops.bn_bwd(_d(dyh), _d(y.detach()), _d(x), _d(mean), _d(invstd), _d(gamma), 0.2,
                                        want_dz=True, act_mode=1, dy_pooled=True)
            res.append(("bn_bwd_pooled_dz(%d,%d,%d,%d)" % (B, C, H, W), _err(dz, rt.grad), 1e-5))
        else:
            dx, _, dg, db = ops.bn_bwd(_d(dyh), None, _d(x), _d(mean), _d(invstd), _d(gamma), 0.2, want_dz=False,
                                       beta=_d(beta), act_mode=2, dy_pooled=True)
        res.append(("bn_bwd_pooled_dx(%d,%d,%d,%d)" % (B, C, H, W), _err(dx, xt.grad), 2e-5))
        res.append(("bn_bwd_pooled_dgamma(%d,%d,%d,%d)" % (B, C, H, W), _err(dg, gt.grad), 2e-5))
        res.append(("bn_bwd_pooled_dbeta(%d,%d,%d,%d)" % (B, C, H, W), _err(db, bt.grad), 2e-5))
    return res


def check_conv_up_wgrad(shape):
    """d/dw of conv3x3(Upsample(2,'nearest')(x)) — phase-form F(2x2,2x2) weight gradient"""
    from sivae_hip import ops
    B, Ci, Co, H, W, ks = shape  # H, W = dy size
    xs = _rand(B, Ci, H // 2, W // 2, seed=5)
    w = _rand(Co, Ci, 3, 3, seed=2).requires_grad_()
    dy = _rand(B, Co, H, W, seed=4)
    _conv_ref(F.interpolate(xs, scale_factor=2, mode="nearest"), w).backward(dy)
    assert ops._lib.load().sivae_conv2d_wino_up_wgrad_supported(H // 2, W // 2) == 1
    dw = ops.conv2d_wgrad(_d(xs), _d(dy), 3, upsample=True)
    return [("wino_up_wgrad%s" % (shape,), _err(dw, w.grad), WINO_TOL)]


def check_dgrad_bnbwd(shape):
    """conv2 data gradient with the BatchNorm-1 backward sums reduced in its epilogue + the finishing dx pass"""
    from sivae_hip import ops
    B, Cm, Co, H, W, ks = shape
    a = _rand(B, Cm, H, W, seed=1).requires_grad_()
    gamma, beta = (_rand(Cm, seed=8).abs() + 0.5).requires_grad_(), _rand(Cm, seed=9).requires_grad_()
    w2 = _rand(Co, Cm, 3, 3, seed=2, scale=1.0 / math.sqrt(Cm * 9))
    dc = _rand(B, Co, H, W, seed=4)
    h = F.leaky_relu(F.batch_norm(a, None, None, gamma, beta, True, 0.1, 1e-5), 0.2)
    _conv_ref(h, w2).backward(dc)
    ad = a.detach()
    mean = ad.mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(ad.var((0, 2, 3), unbiased=False) + 1e-5)
    assert ops._lib.load().sivae_conv2d_wino_supported(H, W) == 1
    dh, part = ops.conv2d_dgrad_bnbwd(_d(dc), ops.PackedW(_d(w2), 1), Cm, _d(ad), _d(mean), _d(invstd),
                                      _d(gamma.detach()), _d(beta.detach()), 0.2)
    da, dg, db = ops.bn_bwd_from_partials(dh, _d(ad), _d(mean), _d(invstd), _d(gamma.detach()), _d(beta.detach()), part, 0.2)
    return [("dgrad_bnbwd_da%s" % (shape,), _err(da, a.grad), 5e-5),
            ("dgrad_bnbwd_dgamma%s" % (shape,), _err(dg, gamma.grad), 5e-5),
            ("dgrad_bnbwd_dbeta%s" % (shape,), _err(db, beta.grad), 5e-5)]


def check_input_u8():
    """uint8 -> fp32 (/255) with per-sample mirror, NCHW and NHWC sources (bit-exact: one multiply per element)"""
    from sivae_hip import ops
    g = torch.Generator().manual_seed(5)
    res = []
    for (B, C, H, W) in [(5, 3, 32, 32), (3, 1, 28, 28), (2, 3, 17, 23)]:
        src = torch.randint(0, 256, (B, C, H, W), generator=g, dtype=torch.uint8)
        flip = torch.tensor([i % 2 for i in range(B)], dtype=torch.int32)
        ref = src.float() * (1.0 / 255.0)
        ref_f = torch.where(flip.view(-1, 1, 1, 1) != 0, ref.flip(3), ref)
        out = ops.u8_to_f32(src.to(DEV), flip.to(DEV))
        res.append(("u8_to_f32_nchw_flip(%d,%d,%d,%d)" % (B, C, H, W), float((out.cpu() - ref_f).abs().max()), 0.0))
        out = ops.u8_to_f32(src.permute(0, 2, 3, 1).contiguous().to(DEV), None, nhwc=True)
        res.append(("u8_to_f32_nhwc(%d,%d,%d,%d)" % (B, C, H, W), float((out.cpu() - ref).abs().max()), 0.0))
    return res


def check_output_u8():
    """fp32 -> uint8 quantisation of generated images: np.clip(x * 255, 0, 255).astype(np.uint8) (bit-exact)"""
    import numpy as np
    from sivae_hip import ops
    g = torch.Generator().manual_seed(11)
    res = []
    for shape in [(4, 3, 32, 32), (2, 3, 5, 7), (1, 1, 28, 28), (3, 3, 64, 64)]:
        x = torch.rand(shape, generator=g) * 1.4 - 0.2
        x.view(-1)[:6] = torch.tensor([0.0, 1.0, 254.9999 / 255.0, -0.0, 1.0 + 1e-7, 0.5])
        ref = np.clip(x.numpy() * 255, 0, 255).astype(np.uint8)
        out = ops.f32_to_u8(x.to(DEV)).cpu().numpy()
        res.append(("f32_to_u8%s" % (shape,), float(np.abs(out.astype(np.int32) - ref.astype(np.int32)).max()), 0.0))
    x = torch.full((32,), float("nan"))
    res.append(("f32_to_u8_nan", float(ops.f32_to_u8(x.to(DEV)).max()), 0.0))
    return res


def check_bn_bwd_fused():
    """the one-pass persistent BatchNorm backward (bn_fused.hip) against the three-launch form (bn.hip) on the same inputs:
    every sign source (none / saved output / recomputed from x / 1-bit mask), pooled dy, dz at full resolution and as 2x2
    block sums, one and two segments, plane sets of one slab, several slabs and several groups per half-grid — and against
    torch fp64 for the plain variant.  Summation orders differ, values must agree to fp32 rounding."""
    from sivae_hip import ops
    res = []
    L = ops._lib.load()

    def both(fn):
        out = []
        for fused in (False, True):
            ops.BN_FUSED = fused
            try:
                out.append(fn())
            finally:
                ops.BN_FUSED = True
        return out

    shapes = [(4, 16, 4, 4, 1), (8, 8, 8, 16, 2), (6, 40, 16, 16, 1), (4, 24, 64, 64, 2), (2, 520, 8, 8, 1),
              (16, 3, 128, 128, 1), (32, 2, 256, 256, 2), (2, 700, 32, 32, 2),
              (128, 2, 256, 256, 1)]  # (the last: a plane set larger than a half-grid holds -> the one-grid form)
    for (B, C, H, W, nseg) in shapes:
        tag = "(%d,%d,%d,%d,seg%d)" % (B, C, H, W, nseg)
        assert L.sivae_bn_bwd_fused_supported(B, C, H, W, B // nseg) == 1, tag
        x = _d(_rand(B, C, H, W, seed=1) * 1.5 + 0.3)
        r = _d(_rand(B, C, H, W, seed=2))
        gamma, beta = _d(_rand(C, seed=8).abs() + 0.5), _d(_rand(C, seed=9))
        dy = _d(_rand(B, C, H, W, seed=4))
        dyh = _d(_rand(B, C, H // 2, W // 2, seed=5))
        Bs = B // nseg
        xs = x.view(nseg, Bs, C, H * W).double()
        mean = xs.mean((1, 3)).float().reshape(-1).contiguous()
        invstd = (1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).float().reshape(-1).contiguous()
        y = ops.bn_apply_act(x, r, mean, invstd, gamma, beta, 0.2, nseg=nseg)
        # sign recomputed from x (BatchNorm-1 of a block), no dz
        a, b = both(lambda: ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg))
        for n_, i_ in (("dx", 0), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused_act2_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
        # sign from the saved output, dz at full resolution; pooled dy
        a, b = both(lambda: ops.bn_bwd(dy, y, x, mean, invstd, gamma, 0.2, want_dz=True, act_mode=1, nseg=nseg))
        for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused_act1_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
        a, b = both(lambda: ops.bn_bwd(dyh, y, x, mean, invstd, gamma, 0.2, want_dz=True, act_mode=1, dy_pooled=True,
                                       nseg=nseg))
        for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused_act1_pooled_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
        a, b = both(lambda: ops.bn_bwd_dzsum(dy, y, x, mean, invstd, gamma, 0.2, nseg=nseg))
        for n_, i_ in (("dx", 0), ("dzh", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused_act1_dzsum_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
        # no activation
        a, b = both(lambda: ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, act_mode=0, nseg=nseg))
        res.append(("bn_fused_act0_dx%s" % tag, _err(b[0], a[0]), 3e-6))
        if W % 8 == 0:
            _, _, mask = ops.bn_apply_act_signmask(x, r, mean, invstd, gamma, beta, 0.2, nseg=nseg)
            a, b = both(lambda: ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, nseg=nseg))
            for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
                res.append(("bn_fused_mask_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
            a, b = both(lambda: ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, dz_sum=True, nseg=nseg))
            for n_, i_ in (("dx", 0), ("dzh", 1), ("dgamma", 2), ("dbeta", 3)):
                res.append(("bn_fused_mask_dzsum_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
            a, b = both(lambda: ops.bn_bwd_signmask(dyh, mask, x, mean, invstd, gamma, 0.2, dy_pooled=True, nseg=nseg))
            for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
                res.append(("bn_fused_mask_pooled_%s%s" % (n_, tag), _err(b[i_], a[i_]), 3e-6))
            # no parameter gradients wanted (frozen network): dx only
            a, b = both(lambda: ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, want_param_grads=False,
                                                    nseg=nseg))
            res.append(("bn_fused_mask_nopg_dx%s" % tag, _err(b[0], a[0]), 3e-6))
    # against torch fp64 (unsegmented, sign recomputed from x)
    B, C, H, W = 8, 12, 32, 32
    xt = (_rand(B, C, H, W, seed=1) * 2.0 + 0.7).requires_grad_()
    gt, bt = (_rand(C, seed=3) * 0.5 + 1.0).requires_grad_(), _rand(C, seed=4).requires_grad_()
    dyt = _rand(B, C, H, W, seed=7)
    F.leaky_relu(F.batch_norm(xt, None, None, gt, bt, True, 0.1, 1e-5), 0.2).backward(dyt)
    xd = _d(xt.detach())
    mean, invstd = ops.bn_stats(xd)
    dx, _, dg, db = ops.bn_bwd(_d(dyt), None, xd, mean, invstd, _d(gt.detach()), 0.2, beta=_d(bt.detach()), act_mode=2)
    res.append(("bn_fused_vs_fp64_dx", _err(dx, xt.grad), 2e-5))
    res.append(("bn_fused_vs_fp64_dgamma", _err(dg, gt.grad), 2e-5))
    res.append(("bn_fused_vs_fp64_dbeta", _err(db, bt.grad), 2e-5))
    # repeated calls leave the barrier / counter state consistent: same result, bit for bit
    dx2, _, dg2, db2 = ops.bn_bwd(_d(dyt), None, xd, mean, invstd, _d(gt.detach()), 0.2, beta=_d(bt.detach()), act_mode=2)
    res.append(("bn_fused_repeat_bitwise", float((dx2 - dx).abs().max() + (dg2 - dg).abs().max()), 0.0))
    return res


def _bn_fp64_ref(x, r, gamma, beta, dy, nseg, act, pooled):
    """torch CPU fp64 reference of BatchNorm2d (training statistics per batch SEGMENT, shared affine) [+ residual]
    + LeakyReLU(0.2) [+ AvgPool2d(2)] backward (reference ops: train_soft_intro_vae.py:57-63,71-74,90-93)
    -> dx, dz (gradient of the residual branch, or None), dgamma, dbeta"""
    xt = x.clone().requires_grad_()
    gt, bt = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rt = r.clone().requires_grad_() if act in (1, 3) else None
    Bs = x.shape[0] // nseg
    ys = []
    for g in range(nseg):
        sl = slice(g * Bs, (g + 1) * Bs)
        v = F.batch_norm(xt[sl], None, None, gt, bt, True, 0.1, 1e-5)
        if rt is not None:
            v = v + rt[sl]
        if act != 0:
            v = F.leaky_relu(v, 0.2)
        ys.append(v)
    y = torch.cat(ys)
    if pooled:
        y = F.avg_pool2d(y, 2)
    y.backward(dy)
    return xt.grad, (rt.grad if rt is not None else None), gt.grad, bt.grad


def check_bn_bwd_fused_fp64():
    """the PERSISTENT forms of the one-pass BatchNorm backward (bn_fused.hip) directly against torch CPU fp64 — every
    sign source, pooled dy, dz at full resolution and as 2x2 block sums — at shapes that force (i) two half-grids walking
    several groups each, (ii) the one-grid form (a plane set larger than a half-grid holds), (iii) two segments (the
    last-arriver fold of dgamma / dbeta over segments)."""
    from sivae_hip import ops
    res = []
    L = ops._lib.load()
    assert ops.BN_FUSED
    for (B, C, H, W, nseg, what) in [(32, 8, 256, 256, 1, "halfgrids"), (128, 2, 256, 256, 1, "onegrid"),
                                     (32, 6, 256, 256, 2, "2seg"), (64, 24, 128, 128, 2, "2seg128")]:
        tag = "_%s(%d,%d,%d,%d,seg%d)" % (what, B, C, H, W, nseg)
        Bs = B // nseg
        assert L.sivae_bn_bwd_fused_supported(B, C, H, W, Bs) == 1, tag
        assert Bs * H * W // 8 > 256 * 10, tag  # (not the barrier-free local form)
        x = _rand(B, C, H, W, seed=1) * 1.5 + 0.3
        r = _rand(B, C, H, W, seed=2)
        gamma, beta = _rand(C, seed=8).abs() + 0.5, _rand(C, seed=9)
        dy = _rand(B, C, H, W, seed=4)
        dyh = _rand(B, C, H // 2, W // 2, seed=5)
        xd, rd, gd, bd, dyd, dyhd = _d(x), _d(r), _d(gamma), _d(beta), _d(dy), _d(dyh)
        xf = xd.cpu().double()  # (the reference sees the fp32-rounded inputs)
        rf, gf, bf_, dyf, dyhf = rd.cpu().double(), gd.cpu().double(), bd.cpu().double(), dyd.cpu().double(), dyhd.cpu().double()
        xs = xf.view(nseg, Bs, C, H * W)
        mean = _d(xs.mean((1, 3)).reshape(-1))
        invstd = _d((1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).reshape(-1))
        tol = 2e-5
        # act 0 (no activation) and act 2 (sign recomputed from x)
        for act in (0, 2):
            ref = _bn_fp64_ref(xf, rf, gf, bf_, dyf, nseg, act, False)
            out = ops.bn_bwd(dyd, None, xd, mean, invstd, gd, 0.2, beta=bd, act_mode=act, nseg=nseg)
            for n_, i_ in (("dx", 0), ("dgamma", 2), ("dbeta", 3)):
                res.append(("bn_fused64_act%d_%s%s" % (act, n_, tag), _err(out[i_], ref[i_]), tol))
        # act 1 (sign from the saved output): dz full, pooled dy, dz block sums
        y = ops.bn_apply_act(xd, rd, mean, invstd, gd, bd, 0.2, nseg=nseg)
        ref = _bn_fp64_ref(xf, rf, gf, bf_, dyf, nseg, 1, False)
        out = ops.bn_bwd(dyd, y, xd, mean, invstd, gd, 0.2, want_dz=True, act_mode=1, nseg=nseg)
        for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused64_act1_%s%s" % (n_, tag), _err(out[i_], ref[i_]), tol))
        out = ops.bn_bwd_dzsum(dyd, y, xd, mean, invstd, gd, 0.2, nseg=nseg)
        refz = 4.0 * F.avg_pool2d(ref[1], 2)
        for n_, o_, r_ in (("dx", out[0], ref[0]), ("dzh", out[1], refz), ("dgamma", out[2], ref[2]), ("dbeta", out[3], ref[3])):
            res.append(("bn_fused64_act1_dzsum_%s%s" % (n_, tag), _err(o_, r_), tol))
        refp = _bn_fp64_ref(xf, rf, gf, bf_, dyhf, nseg, 1, True)
        out = ops.bn_bwd(dyhd, y, xd, mean, invstd, gd, 0.2, want_dz=True, act_mode=1, dy_pooled=True, nseg=nseg)
        for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused64_act1_pooled_%s%s" % (n_, tag), _err(out[i_], refp[i_]), tol))
        # act 3 (1-bit sign mask)
        _, _, mask = ops.bn_apply_act_signmask(xd, rd, mean, invstd, gd, bd, 0.2, nseg=nseg)
        out = ops.bn_bwd_signmask(dyd, mask, xd, mean, invstd, gd, 0.2, nseg=nseg)
        for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused64_mask_%s%s" % (n_, tag), _err(out[i_], ref[i_]), tol))
        out = ops.bn_bwd_signmask(dyhd, mask, xd, mean, invstd, gd, 0.2, dy_pooled=True, nseg=nseg)
        for n_, i_ in (("dx", 0), ("dz", 1), ("dgamma", 2), ("dbeta", 3)):
            res.append(("bn_fused64_mask_pooled_%s%s" % (n_, tag), _err(out[i_], refp[i_]), tol))
        del x, r, dy, xf, rf, dyf, ref, refp, out, y, mask
    return res


def _side_traffic(stop_after):
    """a streaming-store / copy load on a second stream (saturates the memory pipe next to the kernel under test);
    returns (stream, launcher)"""
    side = torch.cuda.Stream()
    buf = torch.empty(256 << 20, dtype=torch.float32, device=DEV)  # 1 GiB
    src = torch.empty(64 << 20, dtype=torch.float32, device=DEV)

    def pump(n):
        with torch.cuda.stream(side):
            for i in range(n):
                buf.fill_(float(i))
                buf[:src.numel()].copy_(src)
    return side, pump


def check_store_hazard_stress():
    """Hazard stress (round-4 finding: a 16-byte buffer store picked up rewritten data registers under back-pressure):
    200 back-to-back calls of the persistent BatchNorm backward at its headline shape 256 x 64 x 256^2 (two segments, sign
    mask, block-summed dz) with a streaming-store kernel running on a second stream — every result bit-equal to the first,
    and equal to the three-launch form to rounding; then the same over the 16-byte-store epilogues of conv_wino4
    (64 -> 64 @ 256^2) and conv_wino_up (64 -> 64 @ 128^2 -> 256^2), 60 calls each."""
    from sivae_hip import ops
    res = []
    side, pump = _side_traffic(0)
    B, C, H, W, nseg = 256, 64, 256, 256, 2
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, C, H, W, device=DEV, generator=g) * 1.5 + 0.3
    r = torch.randn(B, C, H, W, device=DEV, generator=g)
    dy = torch.randn(B, C, H, W, device=DEV, generator=g)
    gamma = torch.rand(C, device=DEV, generator=g) + 0.5
    beta = torch.randn(C, device=DEV, generator=g)
    Bs = B // nseg
    xs = x.view(nseg, Bs, C, H * W)
    mean = xs.mean((1, 3)).reshape(-1).contiguous()
    invstd = (1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).reshape(-1).contiguous()
    _, _, mask = ops.bn_apply_act_signmask(x, r, mean, invstd, gamma, beta, 0.2, nseg=nseg)
    del r
    first = ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, dz_sum=True, nseg=nseg)
    first = [t.clone() for t in first]
    torch.cuda.synchronize()
    nbad = 0
    for it in range(200):
        if it % 10 == 0:
            pump(6)
        out = ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, dz_sum=True, nseg=nseg)
        nbad += int(not all(torch.equal(a, b) for a, b in zip(out, first)))
        del out
    torch.cuda.synchronize()
    res.append(("hazard_bn_fused_200x_bitwise(256,64,256,256,seg2)", float(nbad), 0.0))
    ops.BN_FUSED = False
    try:
        ref = ops.bn_bwd_signmask(dy, mask, x, mean, invstd, gamma, 0.2, dz_sum=True, nseg=nseg)
    finally:
        ops.BN_FUSED = True
    for n_, i_ in (("dx", 0), ("dzh", 1), ("dgamma", 2), ("dbeta", 3)):
        res.append(("hazard_bn_fused_vs_3launch_%s" % n_, _err(first[i_], ref[i_]), 3e-6))
    ops.bn_fused_check()
    del first, ref, mask, dy, xs
    # conv_wino4 epilogue
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) / 24.0
    pw = ops.PackedW(w, 0)
    y0 = ops.conv2d_fwd(x, pw, 64, 3, want_stats=True)
    y0 = [t.clone() for t in y0] if isinstance(y0, (tuple, list)) else [y0.clone()]
    nbad = 0
    for it in range(60):
        if it % 10 == 0:
            pump(6)
        y1 = ops.conv2d_fwd(x, pw, 64, 3, want_stats=True)
        y1 = list(y1) if isinstance(y1, (tuple, list)) else [y1]
        nbad += int(not all(torch.equal(a, b) for a, b in zip(y1, y0) if a is not None))
        del y1
    torch.cuda.synchronize()
    res.append(("hazard_conv_wino4_60x_bitwise(256,64,64,256,256)", float(nbad), 0.0))
    del y0
    # conv_wino_up epilogue: conv3x3(upsample2(x_low))
    xl = x[:, :, :128, :128].contiguous()
    del x
    y0 = ops.conv2d_fwd(xl, pw, 64, 3, upsample=True, want_stats=True)
    y0 = [t.clone() for t in y0] if isinstance(y0, (tuple, list)) else [y0.clone()]
    nbad = 0
    for it in range(60):
        if it % 10 == 0:
            pump(6)
        y1 = ops.conv2d_fwd(xl, pw, 64, 3, upsample=True, want_stats=True)
        y1 = list(y1) if isinstance(y1, (tuple, list)) else [y1]
        nbad += int(not all(torch.equal(a, b) for a, b in zip(y1, y0) if a is not None))
        del y1
    torch.cuda.synchronize()
    res.append(("hazard_conv_wino_up_60x_bitwise(256,64,64,256,256)", float(nbad), 0.0))
    return res


def check_bn_fused_squatter():
    """Forward progress and bit-equality of the persistent BatchNorm backward while another kernel HOLDS part of the chip
    (the footprint of a collective on a side stream: 48 workgroups x 512 threads + 32 KB of LDS each, resident for
    ~0.4 s): headline shapes 256 x 64 x 256^2 and 256 x 128 x 128^2; then next to a real (world-size-1) `nccl` process
    group issuing all_reduce(op=AVG, async_op=True) of 110 MB in a loop (AVG: the one-rank path then runs a kernel).  The calls must complete within a bound (no deadlock,
    no trap, poison word clear) and reproduce the undisturbed bits."""
    import time
    from sivae_hip import ops
    L = ops._lib.load()
    res = []
    g = torch.Generator(device=DEV).manual_seed(5)
    stop = torch.zeros(1, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    for (B, C, H, W, nseg) in [(256, 64, 256, 256, 2), (256, 128, 128, 128, 2)]:
        tag = "(%d,%d,%d,%d,seg%d)" % (B, C, H, W, nseg)
        x = torch.randn(B, C, H, W, device=DEV, generator=g) * 1.5 + 0.3
        dy = torch.randn(B, C, H, W, device=DEV, generator=g)
        gamma = torch.rand(C, device=DEV, generator=g) + 0.5
        beta = torch.randn(C, device=DEV, generator=g)
        Bs = B // nseg
        xs = x.view(nseg, Bs, C, H * W)
        mean = xs.mean((1, 3)).reshape(-1).contiguous()
        invstd = (1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).reshape(-1).contiguous()
        alone = ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg)
        alone = [t.clone() for t in alone if t is not None]
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg)
        torch.cuda.synchronize()
        t_alone = (time.time() - t0) / 5
        # squatter: 48 x 512 threads x 32 KB for 0.4 s on the side stream, BN calls on the main stream meanwhile.
        # thin: the BatchNorm grid still fits beside it (slower); fat (~200 VGPRs per wave): 48 CUs take NO BatchNorm block,
        # the grid is not fully resident -> the first call waits at its barrier until the squatter leaves
        for fat in (0, 1):
            stop.zero_()
            torch.cuda.synchronize()
            rc = _support.load().testsupport_squatter(48, 512, 32768, fat, 40_000_000, ops._p(stop),
                                                      ctypes.c_void_p(side.cuda_stream))
            assert rc == 0, rc
            t0 = time.time()
            nbad = 0
            for _ in range(5):
                out = ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg)
                out = [t for t in out if t is not None]
                nbad += int(not all(torch.equal(a, b) for a, b in zip(out, alone)))
            torch.cuda.current_stream().synchronize()
            t_sq = (time.time() - t0) / 5
            stop.fill_(1)  # release the squatter
            torch.cuda.synchronize()
            ops.bn_fused_check()  # (no barrier gave up)
            kind = "fat" if fat else "thin"
            res.append(("squatter_%s_bn_fused_bitwise%s" % (kind, tag), float(nbad), 0.0))
            # bound: it may wait for the squatter to leave (0.4 s over 5 calls) but must not hang: < 1 s per call
            res.append(("squatter_%s_bn_fused_seconds_per_call%s alone=%.4f" % (kind, tag, t_alone), t_sq, 1.0))
        del x, dy, alone, xs
    # a real collective KERNEL on a side stream.  World size 1 (the box has one GPU): a plain in-place SUM is short-circuited
    # by RCCL (ncclLaunchOneRank: no kernel at all), so the loop uses ReduceOp.AVG — a PreMulSum reduction, for which the
    # one-rank path launches RCCL's reduce kernel over the whole 110-MB buffer.  tools/rccl_overlap.sh traces this check
    # with rocprofv3 and tools/rccl_overlap.py lists the RCCL kernel rows that overlap bn_bwd_fused_kernel in time
    # (profiles/r6_rccl_overlap_bn_fused.txt).
    import os
    import torch.distributed as dist
    made = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1)
        made = True
    try:
        B, C, H, W, nseg = 256, 64, 256, 256, 2
        x = torch.randn(B, C, H, W, device=DEV, generator=g) * 1.5 + 0.3
        dy = torch.randn(B, C, H, W, device=DEV, generator=g)
        gamma = torch.rand(C, device=DEV, generator=g) + 0.5
        beta = torch.randn(C, device=DEV, generator=g)
        xs = x.view(nseg, B // nseg, C, H * W)
        mean = xs.mean((1, 3)).reshape(-1).contiguous()
        invstd = (1.0 / torch.sqrt(xs.var((1, 3), unbiased=False) + 1e-5)).reshape(-1).contiguous()
        alone = [t.clone() for t in ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg)
                 if t is not None]
        flat = torch.randn(27_500_000, device=DEV, generator=g)  # 110 MB: the encoder's flat gradient
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)  # (communicator set-up happens in the first collective: not timed)
        torch.cuda.synchronize()
        t0 = time.time()
        nbad = 0
        works = []
        for it in range(20):
            with torch.cuda.stream(side):
                works.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True))
            out = [t for t in ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2, nseg=nseg)
                   if t is not None]
            nbad += int(not all(torch.equal(a, b) for a, b in zip(out, alone)))
        for w_ in works:
            w_.wait()
        torch.cuda.synchronize()
        ops.bn_fused_check()
        res.append(("nccl_allreduce_bn_fused_bitwise(256,64,256,256,seg2)", float(nbad), 0.0))
        res.append(("nccl_allreduce_bn_fused_seconds_per_call", (time.time() - t0) / 20, 1.0))
    finally:
        if made:
            dist.destroy_process_group()
    return res


def check_bn_fused_timeout():
    """the grid barrier's timeout path, without a hook in the product library: the barrier state is CALLER-owned, so the
    test corrupts it (every arrival counter of both half-grid areas starts far from zero: the arrivals of the launch never
    add up to a full XCD group, no generation flag ever advances) and shortens the spin limit for that call
    (SIVAE_BN_FUSED_SPIN_LIMIT is read per launch).  The launch must END — no trap, no hang —, set the poison word
    (ops.bn_fused_check raises and zeroes the state), and the next call must work and reproduce the result bit for bit."""
    import os
    from sivae_hip import ops
    L = ops._lib.load()
    res = []
    B, C, H, W = 32, 8, 256, 256
    x = _d(_rand(B, C, H, W, seed=1) * 1.5 + 0.3)
    dy = _d(_rand(B, C, H, W, seed=4))
    gamma, beta = _d(_rand(C, seed=8).abs() + 0.5), _d(_rand(C, seed=9))
    mean, invstd = ops.bn_stats(x)
    good = [t.clone() for t in ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2) if t is not None]
    torch.cuda.synchronize()
    ops.bn_fused_check()
    assert ops._bn_states, "the one-pass BatchNorm backward did not run (no barrier state was created)"
    area = 1024  # uints per half-grid barrier area (bn_fused_common.h: BF_BAR_UINTS), arrival counter of XCD i at 32 i
    assert L.sivae_bn_bwd_fused_poison_word() == area - 32
    for buf in ops._bn_states.values():
        for half in range(2):
            for xcd in range(8):
                buf[half * area + 32 * xcd] = 0x40000000
    old = os.environ.get("SIVAE_BN_FUSED_SPIN_LIMIT")
    os.environ["SIVAE_BN_FUSED_SPIN_LIMIT"] = str(1 << 14)  # (gives up after milliseconds instead of tens of seconds)
    try:
        ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2)  # abandoned after the spin limit
        torch.cuda.synchronize()  # (returns: the kernel ended by itself)
    finally:
        if old is None:
            os.environ.pop("SIVAE_BN_FUSED_SPIN_LIMIT", None)
        else:
            os.environ["SIVAE_BN_FUSED_SPIN_LIMIT"] = old
    raised = 0.0
    try:
        ops.bn_fused_check()
    except RuntimeError:
        raised = 1.0
    res.append(("bn_fused_timeout_raises", 1.0 - raised, 0.0))
    again = [t for t in ops.bn_bwd(dy, None, x, mean, invstd, gamma, 0.2, beta=beta, act_mode=2) if t is not None]
    torch.cuda.synchronize()
    ops.bn_fused_check()
    res.append(("bn_fused_after_reset_bitwise", float(not all(torch.equal(a, b) for a, b in zip(again, good))), 0.0))
    return res


def all_checks():
    """-> list of (label, thunk) ; every thunk returns a list of (name, err, tol)"""
    checks = []
    for s in CONV_SHAPES:
        checks.append(("conv_fwd%s" % (s,), lambda s=s: check_conv_fwd(s)))
        checks.append(("conv_dgrad%s" % (s,), lambda s=s: check_conv_dgrad(s)))
        checks.append(("conv_wgrad%s" % (s,), lambda s=s: check_conv_wgrad(s, wino=False)))
    checks.append(("conv_fwd_bias", lambda: check_conv_fwd((2, 64, 3, 32, 32, 5), bias=True)))
    checks.append(("conv_fwd_stats", lambda: check_conv_fwd((3, 64, 128, 32, 32, 3), stats=True)
                   + check_conv_fwd((3, 64, 64, 32, 32, 3), stats=True)
                   + check_conv_fwd((3, 24, 40, 12, 12, 3), stats=True)))
    for s in [(2, 64, 128, 32, 32, 3), (2, 128, 64, 16, 16, 1), (3, 64, 64, 8, 8, 3), (2, 64, 3, 16, 16, 5)]:
        checks.append(("conv_fused%s" % (s,), lambda s=s: check_conv_fused(s)))
    for s in WINO_SHAPES:
        checks.append(("wino_fwd%s" % (s,), lambda s=s: check_conv_fwd(s, wino=True)))
        checks.append(("wino_dgrad%s" % (s,), lambda s=s: check_conv_dgrad(s, wino=True)))
        if s[4] >= 16:
            checks.append(("wino_wgrad%s" % (s,), lambda s=s: check_conv_wgrad(s, wino=True)))
    checks.append(("wino_fwd_stats", lambda: check_conv_fwd((3, 64, 128, 32, 32, 3), stats=True, wino=True)
                   + check_conv_fwd((3, 24, 40, 12, 20, 3), stats=True, wino=True)
                   + check_conv_fwd((2, 64, 3, 16, 48, 3), bias=True, wino=True)))  # bias -> direct kernel
    for s in [(2, 64, 128, 32, 32, 3), (3, 40, 72, 16, 16, 3), (2, 16, 64, 24, 40, 3), (3, 64, 64, 8, 8, 3), (5, 32, 64, 4, 4, 3)]:
        checks.append(("wino_fused%s" % (s,), lambda s=s: check_conv_fused(s, wino=True)))
    checks.append(("wino_fwd_stats8", lambda: check_conv_fwd((5, 32, 96, 8, 8, 3), stats=True, wino=True)
                   + check_conv_fwd((7, 32, 96, 4, 4, 3), stats=True, wino=True)))
    for s in [(2, 64, 64, 32, 32, 3), (2, 128, 64, 16, 64, 3), (3, 24, 40, 24, 40, 3), (1, 512, 256, 32, 32, 3),
              (2, 16, 8, 20, 36, 3)]:
        checks.append(("wino_up%s" % (s,), lambda s=s: check_conv_up(s) + check_conv_up(s, pro=True)))
    for s in [(2, 64, 64, 32, 32, 3), (2, 64, 128, 16, 64, 3), (3, 40, 24, 24, 40, 3), (1, 256, 512, 32, 32, 3),
              (2, 8, 16, 20, 36, 3), (2, 130, 20, 16, 32, 3)]:
        checks.append(("wino_up_dgrad%s" % (s,), lambda s=s: check_conv_up_dgrad(s) + check_conv_up_dgrad(s, True)))
    for s in [(2, 64, 64, 32, 32, 3), (2, 64, 128, 16, 64, 3), (3, 40, 24, 24, 40, 3), (1, 256, 512, 32, 32, 3),
              (2, 8, 16, 20, 36, 3), (5, 130, 20, 8, 32, 3), (4, 64, 64, 64, 64, 3)]:
        checks.append(("wino_up_wgrad%s" % (s,), lambda s=s: check_conv_up_wgrad(s)))
    for s in [(2, 64, 64, 32, 32, 3), (3, 40, 72, 16, 16, 3), (4, 64, 32, 8, 8, 3), (6, 32, 64, 4, 4, 3), (2, 24, 16, 12, 20, 3)]:
        checks.append(("dgrad_bnbwd%s" % (s,), lambda s=s: check_dgrad_bnbwd(s)))
    checks.append(("wino_up_stats", lambda: check_conv_up((3, 32, 72, 32, 64, 3), stats=True)
                   + check_conv_up((2, 20, 33, 16, 32, 3), pro=True, stats=True)))
    for s in [(2, 64, 64, 32, 32), (1, 64, 64, 16, 64), (3, 40, 72, 48, 64), (2, 64, 128, 32, 32), (1, 24, 200, 16, 32)]:
        checks.append(("wino4_dgrad_pool%s" % (s,), lambda s=s: check_wino4_dgrad_pool(s) + check_wino4_dgrad_pool(s, True)))
    checks.append(("up_dgrad_splitk", check_up_dgrad_splitk))
    checks.append(("conv5_k75", check_conv5_k75))
    for s in [(2, 64, 64, 32, 32), (1, 64, 128, 16, 64), (3, 128, 64, 32, 32), (2, 32, 40, 16, 32), (1, 256, 256, 32, 32),
              (2, 100, 72, 48, 64)]:
        checks.append(("wino4%s" % (s,), lambda s=s: check_wino4(s, stats=True) + check_wino4(s, accumulate=True)
                       + check_wino4(s, mode=1)))
    for s in [(2, 64, 64, 32, 32), (4, 128, 64, 16, 32), (2, 100, 72, 48, 64), (2, 512, 64, 16, 32)]:
        checks.append(("wino4_pro%s" % (s,), lambda s=s: check_wino4_pro(s) + check_wino4_pro(s, nseg=2)))
    for s in [(2, 64, 64, 16, 16), (6, 40, 72, 16, 16), (4, 512, 128, 16, 16)]:  # 16 x 16 maps: image pairs
        checks.append(("wino4_pair%s" % (s,), lambda s=s: check_wino4(s, stats=True) + check_wino4(s, accumulate=True)
                       + check_wino4(s, mode=1)))
    for s in [(4, 64, 64, 16, 16), (8, 96, 40, 16, 16)]:
        checks.append(("wino4_pair_pro%s" % (s,), lambda s=s: check_wino4_pro(s) + check_wino4_pro(s, nseg=2)))
    # 8 x 8 and 4 x 4 maps (round 6): a work item is a grid of 4 x 2 / 8 x 4 whole images, every seam zero padding
    for s in [(8, 64, 64, 8, 8), (16, 40, 72, 8, 8), (24, 512, 128, 8, 8), (32, 64, 64, 4, 4), (64, 48, 100, 4, 4),
              (32, 512, 64, 4, 4)]:
        checks.append(("wino4_grid%s" % (s,), lambda s=s: check_wino4(s, stats=True) + check_wino4(s, accumulate=True)
                       + check_wino4(s, mode=1)))
    for s in [(16, 64, 64, 8, 8), (32, 96, 40, 8, 8), (64, 64, 64, 4, 4), (128, 96, 40, 4, 4)]:
        checks.append(("wino4_grid_pro%s" % (s,), lambda s=s: check_wino4_pro(s) + check_wino4_pro(s, nseg=2)))
    for s in [(16, 512, 128, 8, 8), (64, 512, 128, 4, 4), (256, 512, 512, 4, 4)]:
        checks.append(("wino4_grid_splitk%s" % (s,), lambda s=s: check_wino4_splitk(s) + check_wino4_splitk(s, accumulate=True)
                       + check_wino4_splitk(s, pro=True)))
    checks.append(("wino4_grid_splitk_seg", lambda: check_wino4_splitk((32, 512, 64, 8, 8), pro=True, nseg=2)
                   + check_wino4_splitk((64, 256, 64, 4, 4), pro=True, nseg=2)))
    for s in [(2, 256, 64, 32, 32), (4, 512, 128, 16, 16), (1, 128, 100, 16, 32), (1, 512, 64, 64, 64)]:
        checks.append(("wino4_splitk%s" % (s,), lambda s=s: check_wino4_splitk(s) + check_wino4_splitk(s, accumulate=True)
                       + check_wino4_splitk(s, pro=True)))
    checks.append(("wino4_splitk_seg", lambda: check_wino4_splitk((4, 256, 64, 32, 32), pro=True, nseg=2)
                   + check_wino4_splitk((8, 512, 64, 16, 16), pro=True, nseg=2)))
    # the same F(4x4,3x3) checks on the bf16-pipe kernel with fp32-exact products (conv_wino4_b6.hip): tolerances unchanged
    checks.append(("wino4b6_pack", check_wino4_b6_pack))
    for s in [(2, 64, 64, 32, 32), (1, 64, 128, 16, 64), (3, 128, 64, 32, 32), (2, 32, 40, 16, 32), (1, 256, 256, 32, 32),
              (2, 100, 72, 48, 64)]:
        checks.append(("wino4b6%s" % (s,), lambda s=s: check_wino4(s, stats=True, b6=True)
                       + check_wino4(s, accumulate=True, b6=True) + check_wino4(s, mode=1, b6=True)))
    for s in [(2, 64, 64, 32, 32), (4, 128, 64, 16, 32), (2, 100, 72, 48, 64), (2, 512, 64, 16, 32)]:
        checks.append(("wino4b6_pro%s" % (s,), lambda s=s: check_wino4_pro(s, b6=True) + check_wino4_pro(s, nseg=2, b6=True)))
    for s in [(2, 64, 64, 16, 16), (6, 40, 72, 16, 16), (4, 512, 128, 16, 16)]:
        checks.append(("wino4b6_pair%s" % (s,), lambda s=s: check_wino4(s, stats=True, b6=True)
                       + check_wino4(s, accumulate=True, b6=True) + check_wino4(s, mode=1, b6=True)))
    for s in [(4, 64, 64, 16, 16), (8, 96, 40, 16, 16)]:
        checks.append(("wino4b6_pair_pro%s" % (s,), lambda s=s: check_wino4_pro(s, b6=True) + check_wino4_pro(s, nseg=2, b6=True)))
    for s in [(2, 256, 64, 32, 32), (4, 512, 128, 16, 16), (1, 128, 100, 16, 32)]:
        checks.append(("wino4b6_splitk%s" % (s,), lambda s=s: check_wino4_splitk(s, b6=True)
                       + check_wino4_splitk(s, accumulate=True, b6=True) + check_wino4_splitk(s, pro=True, b6=True)))
    checks.append(("wino4b6_splitk_seg", lambda: check_wino4_splitk((4, 256, 64, 32, 32), pro=True, nseg=2, b6=True)
                   + check_wino4_splitk((8, 512, 64, 16, 16), pro=True, nseg=2, b6=True)))
    for s in [(2, 64, 64, 32, 32), (1, 32, 64, 16, 16), (4, 128, 64, 16, 32), (3, 100, 72, 48, 64), (2, 96, 160, 8, 48),
              (7, 64, 128, 4, 16)]:
        checks.append(("wino4_wgrad%s" % (s,), lambda s=s: check_wino4_wgrad(s) + check_wino4_wgrad(s, pro=True)))
    checks.append(("wino4_wgrad_seg", lambda: check_wino4_wgrad((4, 64, 64, 32, 32), pro=True, nseg=2)
                   + check_wino4_wgrad((6, 40, 72, 16, 48), pro=True, nseg=2)))
    # 8 x 8 and 4 x 4 maps (round 6): a stage's strip is 2 / 4 whole images side by side
    for s in [(2, 64, 64, 8, 8), (6, 40, 72, 8, 8), (16, 128, 64, 8, 8), (4, 64, 64, 4, 4), (12, 100, 72, 4, 4),
              (64, 64, 128, 4, 4)]:
        checks.append(("wino4_wgrad_grid%s" % (s,), lambda s=s: check_wino4_wgrad(s) + check_wino4_wgrad(s, pro=True)))
    checks.append(("wino4_wgrad_grid_seg", lambda: check_wino4_wgrad((8, 64, 64, 8, 8), pro=True, nseg=2)
                   + check_wino4_wgrad((24, 40, 72, 4, 4), pro=True, nseg=2)))
    checks.append(("conv1x1_stream", check_conv1x1_stream))
    checks.append(("conv5_edge", check_conv5_edge))
    checks.append(("linear", check_linear))
    checks.append(("linear_fast", check_linear_fast))
    checks.append(("wino_splitk", check_wino_splitk))
    for s in BN_SHAPES:
        checks.append(("bn%s" % (s,), lambda s=s: check_bn(s, False)))
        checks.append(("bn+res%s" % (s,), lambda s=s: check_bn(s, True)))
    checks.append(("bn_from_conv", check_bn_from_conv))
    checks.append(("eltwise", check_eltwise))
    checks.append(("input_u8", check_input_u8))
    checks.append(("bn_bwd_pooled", check_bn_bwd_pooled))
    checks.append(("bn_apply_pool", check_bn_apply_pool))
    checks.append(("bn_bwd_dzsum", check_bn_bwd_dzsum))
    checks.append(("bn_signmask", check_bn_signmask))
    checks.append(("bn_bwd_fused", check_bn_bwd_fused))
    checks.append(("bn_bwd_fused_fp64", check_bn_bwd_fused_fp64))
    checks.append(("store_hazard_stress", check_store_hazard_stress))
    checks.append(("bn_fused_squatter", check_bn_fused_squatter))
    checks.append(("bn_fused_timeout", check_bn_fused_timeout))
    checks.append(("output_u8", check_output_u8))
    checks.append(("space_to_depth", check_space_to_depth))
    checks.append(("bn_apply_resup", check_bn_apply_resup))
    checks.append(("losses", check_losses))
    checks.append(("kl_tensor_prior", check_kl_tensor_prior))
    checks.append(("adversarial", check_adversarial))
    checks.append(("randn", check_randn))
    checks.append(("adam", check_adam))
    return checks


def main():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "soft-intro-vae-pytorch_amd"))
    nfail = 0
    rows = []
    filt = sys.argv[1:]
    for label, thunk in all_checks():
        if filt and not any(f in label for f in filt):
            continue
        try:
            for name, err, tol in thunk():
                ok = err <= tol
                nfail += (not ok)
                rows.append("%-4s %-58s err=%.3e tol=%.1e" % ("ok" if ok else "FAIL", name, err, tol))
        except Exception:  # noqa: BLE001
            nfail += 1
            rows.append("EXC  %s\n%s" % (label, traceback.format_exc(limit=3)))
        torch.cuda.synchronize()
    print("\n".join(rows))
    print("kernel_checks: %d failures of %d" % (nfail, len(rows)))
    return nfail


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
