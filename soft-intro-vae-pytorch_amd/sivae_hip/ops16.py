"""Tensor-level wrappers over the bf16-mode entry points of libsivae_hip (sivae_bf16_*; no autograd here — see
`functional16.py`).

Activations are torch.bfloat16 tensors of shape [B, C16/8, H, W, 8] ("blocked NCHW": 8 consecutive channels per pixel
vector, channel count padded to a multiple of 16 with zeros — bf16_common.h).  Statistics, BatchNorm parameters and their
gradients, weights and weight gradients are float32.  Like `ops`, everything launches on torch's current stream of the
tensor's device and there is no CPU path.
"""
import os

import torch

from . import lib as _lib
from . import ops

_p, _s = ops._p, ops._s
LRELU_SLOPE = ops.LRELU_SLOPE
TIMER_KEYS = True


# SIVAE_BF16_POOL_DGRAD=0: the data gradient of a conv of an upsampled input is written at full resolution and summed by
# a second launch (A/B measurements)
POOL_DGRAD = os.environ.get("SIVAE_BF16_POOL_DGRAD", "1") != "0"
KS51 = 51  # ks code of the 5-row x 1-column conv (kw-packed RGB-side layers, include/sivae_hip.h)


def _taps(ks):
    return 5 if ks == KS51 else ks * ks


def cblocks(C):
    return ((C + 15) // 16) * 2


def _req16(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("sivae_hip: tensor is on %s — the bf16 kernels need a ROCm device tensor and have no "
                               "CPU fallback" % t.device)
        if t.dtype != torch.bfloat16 or t.dim() != 5 or t.shape[-1] != 8 or (t.shape[1] & 1):
            raise TypeError("sivae_hip: expected a blocked bfloat16 activation [B, C16/8, H, W, 8], got %s %s"
                            % (t.dtype, tuple(t.shape)))
        if not t.is_contiguous():
            raise ValueError("sivae_hip: tensor must be contiguous")


def empty_blocked(B, C, H, W, device):
    return torch.empty((B, cblocks(C), H, W, 8), dtype=torch.bfloat16, device=device)


def from_f32(x, scale=1.0):
    """fp32 NCHW -> blocked bf16 (padded channels zero)"""
    ops._require(x)
    B, C, H, W = x.shape
    y = empty_blocked(B, C, H, W, x.device)
    _lib.call("sivae_bf16_from_f32_nchw", _p(x), _p(y), B, C, H, W, float(scale), _s(x))
    return y


def to_f32(xb, C):
    """blocked bf16 -> fp32 NCHW with C channels"""
    _req16(xb)
    B, Cb, H, W, _ = xb.shape
    assert cblocks(C) == Cb
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=xb.device)
    _lib.call("sivae_bf16_to_f32_nchw", _p(xb), _p(y), B, C, H, W, _s(xb))
    return y


def im2col_kw5(x, sgn):
    """fp32 NCHW [B, C<=3, H, W] -> blocked bf16 16 channels: [kw*C + c][h][w] = x[c][h][w + sgn*(kw-2)]"""
    ops._require(x)
    B, C, H, W = x.shape
    y = empty_blocked(B, 16, H, W, x.device)
    _lib.call("sivae_bf16_im2col_kw5", _p(x), _p(y), B, C, H, W, int(sgn), _s(x))
    return y


def fold_kw5(g, bias, C, sgn):
    """fp32 NCHW [B, 5C, H, W] -> fp32 NCHW [B, C, H, W]: bias[c] + sum_kw g[kw*C + c][h][w + sgn*(kw-2)]"""
    ops._require(g, bias)
    B, C5, H, W = g.shape
    assert C5 == 5 * C
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
    _lib.call("sivae_bf16_fold_kw5", _p(g), _p(bias), _p(y), B, C, H, W, int(sgn), _s(g))
    return y


class PackedW16:
    """bf16 MFMA-operand slabs of one fp32 master weight (mode 0 forward, 1 data gradient)"""

    def __init__(self, w, mode):
        ops._require(w)
        if w.dim() == 2:
            Co, Ci, ks = w.shape[0], w.shape[1], 1
        else:
            Co, Ci, ks = w.shape[0], w.shape[1], w.shape[2]
            if w.shape[3] != w.shape[2]:
                if tuple(w.shape[2:]) != (5, 1):
                    raise _lib.SivaeError("sivae_bf16_pack_conv_weight", -4)
                ks = KS51
        self.Co, self.Ci, self.ks, self.mode = Co, Ci, ks, mode
        self.w = w  # (the batched repack rebuilds `data` in place from here: functional16.repack16)
        nbytes = _lib.load().sivae_bf16_pack_conv_weight_bytes(Co, Ci, ks, mode)
        if nbytes == 0:
            raise _lib.SivaeError("sivae_bf16_pack_conv_weight_bytes", -3)
        self.data = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
        _lib.call("sivae_bf16_pack_conv_weight", _p(w), _p(self.data), Co, Ci, ks, mode, _s(w))


def conv2d(x, wp, Ci, Co, ks, bias=None, pro=None, upsample=False, want_stats=False, out=None, accumulate=False,
           out_f32=False):
    """x blocked [B, Cib, Hs, Ws, 8] -> y blocked [B, Cob, H, W, 8] (or fp32 NCHW [B, Co, H, W] with out_f32).
    wp: PackedW16 of the layer (mode 0: forward, Ci/Co the conv's; mode 1: data gradient, Ci/Co swapped by the
    caller).  pro = (mean, invstd, gamma, beta, slope): producer BatchNorm + LeakyReLU fused into the load (3x3)."""
    _req16(x, None if out_f32 else out)
    B, Cib, Hs, Ws, _ = x.shape
    assert Cib == cblocks(Ci), (Cib, Ci)
    H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
    L = _lib.load()
    if out is not None:
        y = out
    elif out_f32:
        y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
    else:
        y = empty_blocked(B, Co, H, W, x.device)
    splitk = (ks == 3 and not out_f32 and bias is None
              and L.sivae_bf16_conv2d_splitk(B, Ci, Co, H, W, ks) > 1)  # small grids: split the input-channel range
    stats = None
    if want_stats:
        rows = (L.sivae_bf16_conv2d_splitk_stats_rows(B, Ci, Co, H, W, ks) if splitk
                else L.sivae_bf16_conv2d_num_px_tiles(B, Co, H, W, ks))
        stats = torch.empty((rows, Co, 2), dtype=torch.float32, device=x.device)
    pm = pi = pg = pb = None
    slope = 1.0
    if pro is not None:
        pm, pi, pg, pb, slope = pro
        ops._require(pm, pi, pg, pb)
    ops._require(bias)
    t0 = ops.TIMER.begin() if ops.TIMER is not None else None
    if splitk:
        ws = ops.workspace(L.sivae_bf16_conv2d_splitk_workspace_bytes(B, Ci, Co, H, W, ks), x.device)
        _lib.call("sivae_bf16_conv2d_fwd_splitk", _p(x), _p(wp.data), _p(y), _p(pm), _p(pi), _p(pg), _p(pb),
                  float(slope), _p(stats), B, Ci, Co, H, W, ks, int(bool(upsample)), int(bool(accumulate)), _p(ws),
                  ws.numel(), _s(x))
    else:
        _lib.call("sivae_bf16_conv2d_fwd", _p(x), _p(wp.data), _p(y), _p(bias), _p(pm), _p(pi), _p(pg), _p(pb),
                  float(slope), _p(stats), B, Ci, Co, H, W, ks, int(bool(upsample)), int(bool(accumulate)),
                  int(bool(out_f32)), _s(x))
    if t0 is not None:
        ops.TIMER.end("bf16_conv_kernel<%d,%s>" % (ks, "co32" if Co <= 32 else ("co64" if Co <= 64 else "co128")),
                      2.0 * B * H * W * Co * Ci * _taps(ks), t0)
    return (y, stats) if want_stats else y


def conv2d_pool_supported(B, Ci, Co, H, W, ks):
    """can conv2d_pool take this layer?  (3x3, even maps, and the grid fills the chip without a K split: the small
    512-channel maps keep conv2d + upsample2_bwd, whose split-K form has no pooled epilogue)"""
    return (POOL_DGRAD and ks == 3 and H % 2 == 0 and W % 2 == 0
            and _lib.load().sivae_bf16_conv2d_splitk(B, Ci, Co, H, W, ks) <= 1)


def conv2d_pool(x, wp, Ci, Co, out=None, accumulate=False):
    """x blocked [B, Cib, H, W, 8] -> the 2x2 BLOCK SUMS of conv3x3(x), blocked [B, Cob, H/2, W/2, 8] (written into /
    accumulated onto `out`): the data gradient of a conv that read its input through upsample addressing, the adjoint
    of the nn.Upsample folded into the conv's epilogue (bf16_conv.hip `poolsum`).  wp: the mode-1 pack."""
    _req16(x, out)
    B, Cib, H, W, _ = x.shape
    assert Cib == cblocks(Ci), (Cib, Ci)
    y = out if out is not None else empty_blocked(B, Co, H // 2, W // 2, x.device)
    assert tuple(y.shape) == (B, cblocks(Co), H // 2, W // 2, 8)
    t0 = ops.TIMER.begin() if ops.TIMER is not None else None
    _lib.call("sivae_bf16_conv2d_fwd_pool", _p(x), _p(wp.data), _p(y), B, Ci, Co, H, W, int(bool(accumulate)), _s(x))
    if t0 is not None:
        ops.TIMER.end("bf16_conv_kernel<3,%s>" % ("co32" if Co <= 32 else ("co64" if Co <= 64 else "co128")),
                      2.0 * B * H * W * Co * Ci * 9, t0)
    return y


def conv2d_wgrad(x, dy, Ci, Co, ks, pro=None, upsample=False, out=None):
    """-> dW fp32 [Co, Ci, ks, ks]  ([Co, Ci, 5, 1] for ks = KS51); written into `out` when given"""
    _req16(x, dy)
    B, Cob, H, W, _ = dy.shape
    assert Cob == cblocks(Co) and x.shape[1] == cblocks(Ci)
    L = _lib.load()
    ws = ops.workspace(L.sivae_bf16_conv2d_wgrad_workspace_bytes(B, Ci, Co, H, W, ks), x.device)
    dw = ops._out(out, (Co, Ci, 5, 1) if ks == KS51 else (Co, Ci, ks, ks), x.device)
    pm = pi = pg = pb = None
    slope = 1.0
    if pro is not None:
        pm, pi, pg, pb, slope = pro
        ops._require(pm, pi, pg, pb)
    t0 = ops.TIMER.begin() if ops.TIMER is not None else None
    _lib.call("sivae_bf16_conv2d_wgrad", _p(x), _p(dy), _p(dw), _p(pm), _p(pi), _p(pg), _p(pb), float(slope), B, Ci,
              Co, H, W, ks, int(bool(upsample)), _p(ws), ws.numel(), _s(x))
    if t0 is not None:
        ops.TIMER.end("bf16_wgrad_kernel<%d>" % ks, 2.0 * B * H * W * Co * Ci * _taps(ks), t0)
    return dw


def _seg_images(B, C, mean, nseg):
    """images per segment of a SEGMENTED batch (nseg passes laid end to end, mean / invstd [nseg][C])"""
    if nseg < 1 or B % nseg or mean.numel() != nseg * C:
        raise ValueError("sivae_hip: %d images / %d statistics entries do not make %d segments of %d channels"
                         % (B, mean.numel(), nseg, C))
    return B // nseg


def bn_apply_act(x, res, mean, invstd, gamma, beta, C, slope=LRELU_SLOPE, res_up=False, want_full=True, pool=False,
                 want_mask=False, nseg=1):
    """LeakyReLU(BN(x) + res) -> (y or None, AvgPool2d(2)(y) or None[, sign mask])
    want_mask: also return the activation's sign bits (uint8, one byte per 8-channel pixel vector) for bn_bwd
    nseg > 1: SEGMENTED batch, mean / invstd [nseg][C] (functional.py)"""
    _req16(x, res)
    ops._require(mean, invstd, gamma, beta)
    B, Cb, H, W, _ = x.shape
    bs = _seg_images(B, C, mean, nseg)
    assert want_full or pool
    y = torch.empty_like(x) if want_full else None
    yp = torch.empty((B, Cb, H // 2, W // 2, 8), dtype=torch.bfloat16, device=x.device) if pool else None
    mask = torch.empty((B, Cb, H, W), dtype=torch.uint8, device=x.device) if want_mask else None
    _lib.call("sivae_bf16_bn_apply_act_seg", _p(x), _p(res), int(bool(res_up)), _p(mean), _p(invstd), _p(gamma),
              _p(beta), float(slope), _p(y), _p(yp), _p(mask), B, C, H, W, bs, _s(x))
    return (y, yp, mask) if want_mask else (y, yp)


def bn_bwd(dy, y, x, mean, invstd, gamma, beta, C, slope=LRELU_SLOPE, dy_pooled=False, want_dz=False, dz_sum=False,
           want_param_grads=True, pg_out=None, nseg=1):
    """-> dx, dz (full resolution, its 2x2 block sums with dz_sum, or None), dgamma, dbeta.
    y: the saved block output (bf16) or its sign mask (uint8 from bn_apply_act(want_mask=True)) — the sign of the
    activation —, or None to recompute the sign from x (needs beta).
    nseg > 1: SEGMENTED batch, mean / invstd [nseg][C]; dgamma / dbeta summed over the segments in segment order."""
    mask = None
    if y is not None and y.dtype == torch.uint8:
        mask, y = y, None
        if not mask.is_cuda or mask.numel() != x.numel() // 8:
            raise TypeError("sivae_hip: bn_bwd needs the uint8 device mask bn_apply_act(want_mask=True) returned")
    _req16(dy, y, x)
    ops._require(mean, invstd, gamma, beta)
    B, Cb, H, W, _ = x.shape
    bs = _seg_images(B, C, mean, nseg)
    dx = torch.empty_like(x)
    dz = None
    if dz_sum:
        dz = torch.empty((B, Cb, H // 2, W // 2, 8), dtype=torch.bfloat16, device=x.device)
    elif want_dz:
        dz = torch.empty_like(x)
    dgamma, dbeta = ops._pg(pg_out, C, x.device, want_param_grads)
    L = _lib.load()
    if (ops.BN_FUSED and ops.SYNC_BN is None and os.environ.get("SIVAE_DP_SAME_DEVICE", "0") != "1"
            and L.sivae_bf16_bn_bwd_fused_seg_supported(B, C, H, W, bs) == 1):
        # one launch, dy and x read once (bf16_bn_fused.hip)
        ws = ops.workspace(L.sivae_bf16_bn_bwd_fused_seg_workspace_bytes(B, C, H, W, bs), x.device)
        _lib.call("sivae_bf16_bn_bwd_fused_seg", _p(dy), int(bool(dy_pooled)), _p(y), _p(mask), _p(x), _p(mean),
                  _p(invstd), _p(gamma), _p(beta), float(slope), _p(dx), _p(dz), int(bool(dz_sum)), _p(dgamma),
                  _p(dbeta), B, C, H, W, bs, _p(ops.bn_fused_state(x.device)), _p(ws), ws.numel(), _s(x))
        return dx, dz, dgamma, dbeta
    # three-launch form (maps the one-launch form does not take, synchronised BatchNorm, two ranks on one device): a
    # segment is a contiguous run of images, so a segmented batch is nseg calls on its slices; dgamma / dbeta are the
    # segments' sums added in segment order
    ws = ops.workspace(L.sivae_bf16_bn_bwd_workspace_bytes(bs, C, H, W), x.device)
    for g in range(nseg):
        sl = slice(g * bs, (g + 1) * bs)
        pg = nseg > 1 and g > 0 and dgamma is not None
        dgg = torch.empty_like(dgamma) if pg else dgamma
        dbg = torch.empty_like(dbeta) if pg else dbeta
        _lib.call("sivae_bf16_bn_bwd", _p(dy[sl]), int(bool(dy_pooled)), _p(None if y is None else y[sl]),
                  _p(None if mask is None else mask[sl]), _p(x[sl]), _p(mean[g * C:(g + 1) * C]),
                  _p(invstd[g * C:(g + 1) * C]), _p(gamma), _p(beta), float(slope), _p(dx[sl]),
                  _p(None if dz is None else dz[sl]), int(bool(dz_sum)), _p(dgg), _p(dbg), bs, C, H, W, _p(ws),
                  ws.numel(), _s(x))
        if pg:
            dgamma += dgg
            dbeta += dbg
    return dx, dz, dgamma, dbeta


def upsample2_fwd(x, C):
    _req16(x)
    B, Cb, H, W, _ = x.shape
    y = torch.empty((B, Cb, 2 * H, 2 * W, 8), dtype=torch.bfloat16, device=x.device)
    _lib.call("sivae_bf16_upsample2_fwd", _p(x), _p(y), B, C, H, W, _s(x))
    return y


def upsample2_bwd(dy, C):
    _req16(dy)
    B, Cb, H2, W2, _ = dy.shape
    dx = torch.empty((B, Cb, H2 // 2, W2 // 2, 8), dtype=torch.bfloat16, device=dy.device)
    _lib.call("sivae_bf16_upsample2_bwd", _p(dy), _p(dx), B, C, H2 // 2, W2 // 2, _s(dy))
    return dx


def add_(y, x):
    _req16(y, x)
    assert y.shape == x.shape
    _lib.call("sivae_bf16_add_inplace", _p(y), _p(x), y.numel() // 8, _s(y))
    return y
