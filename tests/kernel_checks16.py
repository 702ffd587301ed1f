"""Kernel-level parity checks of the bf16 mode (sivae_bf16_* entry points) against stock torch on the CPU in fp64.

The reference op is evaluated in fp64 on the SAME bf16-rounded operands the kernel consumes, so what is measured is
the kernel's own arithmetic (fp32 accumulation order, the final rounding to bf16), not the cost of the format:
  * fp32 outputs (weight gradients, Decoder.predict's fp32 output, BatchNorm partial sums / parameter gradients):
    err = max|hip - ref| / max|ref| <= 2e-5
  * bf16 outputs: one rounding to 8 mantissa bits -> <= 2^-8 of the element; checked as max|hip - ref| / max|ref| <= 6e-3
    plus element-wise |hip - ref| <= 2^-7 |ref| + 1e-3 max|ref|
Used by tests/test_kernels16_gpu.py (pytest -m gpu) and `python tests/kernel_checks16.py [filter...]`.
"""
import sys
import traceback

import torch
import torch.nn.functional as F

DEV = "cuda"
TOL_F32 = 2e-5
TOL_BF16 = 6e-3
SLOPE = 0.2


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def _r16(t):
    """round to bf16, return fp64"""
    return t.float().bfloat16().double()


def cblocks(C):
    return ((C + 15) // 16) * 2


def to_blocked(x):
    """[B, C, H, W] (any float dtype) -> blocked bf16 [B, Cb, H, W, 8] on the CPU"""
    B, C, H, W = x.shape
    Cb = cblocks(C)
    xp = torch.zeros(B, Cb * 8, H, W, dtype=torch.float32)
    xp[:, :C] = x.float()
    return xp.view(B, Cb, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().bfloat16()


def from_blocked(xb, C):
    """blocked bf16 [B, Cb, H, W, 8] (any device) -> fp64 [B, C, H, W] on the CPU"""
    xb = xb.detach().cpu().float()
    B, Cb, H, W, _ = xb.shape
    return xb.permute(0, 1, 4, 2, 3).reshape(B, Cb * 8, H, W)[:, :C].double()


def _err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if a.shape != ref.shape or not torch.isfinite(a).all():
        return float("inf")
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))


def _err16(a, ref):
    """bf16 output: max-norm error, made inf when an element misses the element-wise rounding bound"""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if a.shape != ref.shape or not torch.isfinite(a).all():
        return float("inf")
    m = float(ref.abs().max()) + 1e-30
    if bool(((a - ref).abs() > ref.abs() * 2.0 ** -7 + 1e-3 * m).any()):
        return float("inf")
    return float((a - ref).abs().max() / m)


def _padded_zero(xb, C):
    """the padded channels of a blocked tensor must be exactly zero"""
    xb = xb.detach().cpu().float()
    B, Cb, H, W, _ = xb.shape
    full = xb.permute(0, 1, 4, 2, 3).reshape(B, Cb * 8, H, W)
    return float(full[:, C:].abs().max()) if Cb * 8 > C else 0.0


CONV16_SHAPES = [
    # (B, Ci, Co, H, W, ks)
    (2, 64, 128, 32, 32, 3),
    (2, 128, 64, 16, 16, 3),
    (3, 64, 64, 64, 64, 3),
    (4, 160, 136, 8, 8, 3),   # channel counts that are not multiples of the tiles
    (8, 512, 512, 4, 4, 3),   # long K loop, four images per pixel tile
    (5, 16, 32, 8, 8, 3),     # Co <= 32 configuration
    (2, 24, 16, 12, 20, 3),   # ragged tiles, padded channels (24 -> 32, 16)
    (2, 64, 64, 28, 28, 3),
    (2, 64, 128, 32, 32, 1),
    (2, 128, 64, 16, 16, 1),
    (3, 48, 40, 8, 8, 1),     # 1x1 with 16-channel chunks
    (3, 3, 64, 32, 32, 5),    # encoder stem
    (2, 64, 3, 32, 32, 5),    # decoder predict
    (2, 1, 64, 28, 28, 5),
    (3, 15, 64, 32, 32, 51),  # kw-packed stem (5 taps over 5*3 channels)
    (2, 64, 15, 32, 32, 51),  # kw-packed predict
    (2, 15, 32, 24, 40, 51),
    (2, 32, 5, 28, 28, 51),
]


def _khw(ks):
    """ks code 51 = 5 rows x 1 column (kw-packed RGB-side layers)"""
    return (5, 1) if ks == 51 else (ks, ks)


def check_convert():
    from sivae_hip import ops16
    res = []
    for shape in [(2, 3, 8, 8), (3, 64, 16, 16), (2, 40, 4, 4), (1, 512, 4, 4)]:
        x = _rand(*shape, seed=1)
        xb = ops16.from_f32(x.float().to(DEV))
        res.append(("from_f32%s" % (shape,), _err(xb.cpu().float(), to_blocked(x).float()), 0.0))
        back = ops16.to_f32(xb, shape[1])
        res.append(("to_f32%s" % (shape,), _err(back, _r16(x)), 0.0))
    return res


def check_conv(shape, bias=False, stats=False, out_f32=False):
    from sivae_hip import ops16
    B, Ci, Co, H, W, ks = shape
    kh, kw = _khw(ks)
    x = _r16(_rand(B, Ci, H, W, seed=1))
    w = _rand(Co, Ci, kh, kw, seed=2, scale=1.0 / (Ci * kh * kw) ** 0.5)
    b = _rand(Co, seed=3) if bias else None
    ref = F.conv2d(x, _r16(w), b.float().double() if bias else None, padding=(kh // 2, kw // 2))
    wp = ops16.PackedW16(w.float().to(DEV), 0)
    out = ops16.conv2d(to_blocked(x).to(DEV), wp, Ci, Co, ks, bias=None if b is None else b.float().to(DEV),
                       want_stats=stats, out_f32=out_f32)
    res = []
    tag = "conv16%s%s%s" % (shape, "+bias" if bias else "", "+f32out" if out_f32 else "")
    if stats:
        y, part = out
        yr = from_blocked(y, Co)  # statistics are those of the ROUNDED output
        s = part.double().cpu().sum(0)
        res.append((tag + " sum", _err(s[:, 0], yr.sum((0, 2, 3))), TOL_F32 * 5))
        res.append((tag + " sumsq", _err(s[:, 1], (yr * yr).sum((0, 2, 3))), TOL_F32 * 5))
    else:
        y = out
    if out_f32:
        res.append((tag, _err(y, ref), TOL_F32))
    else:
        res.append((tag, _err16(from_blocked(y, Co), ref), TOL_BF16))
        res.append((tag + " pad", _padded_zero(y, Co), 0.0))
    return res


def check_conv_dgrad(shape):
    from sivae_hip import ops16
    B, Ci, Co, H, W, ks = shape
    kh, kw = _khw(ks)
    dy = _r16(_rand(B, Co, H, W, seed=4))
    w = _rand(Co, Ci, kh, kw, seed=2, scale=1.0 / (Ci * kh * kw) ** 0.5)
    ref = F.conv_transpose2d(dy, _r16(w), padding=(kh // 2, kw // 2))
    wp = ops16.PackedW16(w.float().to(DEV), 1)
    dx = ops16.conv2d(to_blocked(dy).to(DEV), wp, Co, Ci, ks)
    return [("dgrad16%s" % (shape,), _err16(from_blocked(dx, Ci), ref), TOL_BF16)]


def check_conv_dgrad_pool(shape):
    """sivae_bf16_conv2d_fwd_pool: the data gradient of a 3x3 conv of an UPSAMPLED input = the 2x2 block sums of the
    transposed conv of dy (the adjoint of nn.Upsample folded into the epilogue), fresh and accumulated onto a
    half-resolution tensor; against fp64 on the same bf16 operands, ONE rounding of the block sum"""
    from sivae_hip import ops16
    B, Ci, Co, H, W, _ = shape
    dy = _r16(_rand(B, Co, H, W, seed=4))
    w = _rand(Co, Ci, 3, 3, seed=2, scale=1.0 / (Ci * 9) ** 0.5)
    ref = F.avg_pool2d(F.conv_transpose2d(dy, _r16(w), padding=1), 2) * 4
    wp = ops16.PackedW16(w.float().to(DEV), 1)
    dyb = to_blocked(dy).to(DEV)
    dx = ops16.conv2d_pool(dyb, wp, Co, Ci)
    res = [("dgrad16_pool%s" % (shape,), _err16(from_blocked(dx, Ci), ref), TOL_BF16),
           ("dgrad16_pool%s pad" % (shape,), _padded_zero(dx, Ci), 0.0)]
    old = _r16(_rand(B, Ci, H // 2, W // 2, seed=6))
    acc = to_blocked(old).to(DEV)
    ops16.conv2d_pool(dyb, wp, Co, Ci, out=acc, accumulate=True)
    res.append(("dgrad16_pool%s accumulate" % (shape,), _err16(from_blocked(acc, Ci), ref + old), TOL_BF16))
    # and against the two-launch form it replaces (full-resolution bf16 gradient, then the block sum): two roundings there
    two = ops16.upsample2_bwd(ops16.conv2d(dyb, wp, Co, Ci, 3), Ci)
    res.append(("dgrad16_pool%s vs conv + upsample2_bwd" % (shape,), _err(from_blocked(dx, Ci), from_blocked(two, Ci)),
                2 * TOL_BF16))
    return res


def check_conv_wgrad(shape, pro=False, upsample=False):
    from sivae_hip import ops16
    B, Ci, Co, H, W, ks = shape
    Hs, Ws = (H // 2, W // 2) if upsample else (H, W)
    x = _r16(_rand(B, Ci, Hs, Ws, seed=1))
    dy = _r16(_rand(B, Co, H, W, seed=4))
    xin = x
    prot = None
    if pro:
        mean, var = _rand(Ci, seed=5, scale=0.3), _rand(Ci, seed=6).abs() + 0.5
        gamma, beta = _rand(Ci, seed=7, scale=0.5) + 1.0, _rand(Ci, seed=8, scale=0.2)
        invstd = (var + 1e-5).rsqrt()
        sc = (invstd.float() * gamma.float()).double()
        sh = (beta.float() - mean.float() * sc.float()).double()
        xin = _r16(F.leaky_relu((x.float() * sc.float().view(1, -1, 1, 1) + sh.float().view(1, -1, 1, 1)).double(),
                                SLOPE))
        prot = tuple(t.float().to(DEV) for t in (mean, invstd, gamma, beta)) + (SLOPE,)
    if upsample:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    kh, kw = _khw(ks)
    ref = torch.nn.grad.conv2d_weight(xin, (Co, Ci, kh, kw), dy, padding=(kh // 2, kw // 2))
    dw = ops16.conv2d_wgrad(to_blocked(x).to(DEV), to_blocked(dy).to(DEV), Ci, Co, ks, pro=prot, upsample=upsample)
    return [("wgrad16%s%s%s" % (shape, "+pro" if pro else "", "+up" if upsample else ""), _err(dw, ref),
             1e-3 if pro else TOL_F32)]


def check_conv_fused(shape):
    """prologue + upsample + stats + accumulate in one call"""
    from sivae_hip import ops16
    B, Ci, Co, H, W, ks = shape
    res = []
    x = _r16(_rand(B, Ci, H // 2, W // 2, seed=1))
    w = _rand(Co, Ci, ks, ks, seed=2, scale=1.0 / (Ci * ks * ks) ** 0.5)
    wp = ops16.PackedW16(w.float().to(DEV), 0)
    # upsample addressing
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), _r16(w), padding=ks // 2)
    y = ops16.conv2d(to_blocked(x).to(DEV), wp, Ci, Co, ks, upsample=True)
    res.append(("conv16_up%s" % (shape,), _err16(from_blocked(y, Co), ref), TOL_BF16))
    # accumulate
    y0 = _r16(_rand(B, Co, H, W, seed=9))
    yb = to_blocked(y0).to(DEV)
    ops16.conv2d(to_blocked(x).to(DEV), wp, Ci, Co, ks, upsample=True, out=yb, accumulate=True)
    res.append(("conv16_acc%s" % (shape,), _err16(from_blocked(yb, Co), ref + y0), TOL_BF16))
    if ks == 3:
        xf = _r16(_rand(B, Ci, H, W, seed=1))
        mean, var = _rand(Ci, seed=5, scale=0.3), _rand(Ci, seed=6).abs() + 0.5
        gamma, beta = _rand(Ci, seed=7, scale=0.5) + 1.0, _rand(Ci, seed=8, scale=0.2)
        invstd = (var + 1e-5).rsqrt()
        sc = (invstd.float() * gamma.float())
        sh = (beta.float() - mean.float() * sc)
        h = _r16(F.leaky_relu((xf.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double(), SLOPE))
        ref = F.conv2d(h, _r16(w), padding=1)
        prot = tuple(t.float().to(DEV) for t in (mean, invstd, gamma, beta)) + (SLOPE,)
        y, part = ops16.conv2d(to_blocked(xf).to(DEV), wp, Ci, Co, ks, pro=prot, want_stats=True)
        # the prologue's own rounding of h to bf16 can differ by one ulp from the reference's (fma vs mul+add): allow
        # a looser max-norm bound here
        res.append(("conv16_pro%s" % (shape,), _err(from_blocked(y, Co), ref), 1.5e-2))
        yr = from_blocked(y, Co)
        s = part.double().cpu().sum(0)
        res.append(("conv16_pro_stats%s" % (shape,), _err(s[:, 0], yr.sum((0, 2, 3))), 1e-4))
    return res


def check_conv_big_tiles():
    """the 64 co x 512 px and 128 co x 256 px tiles of the 3x3 kernels (128 accumulators per wave) are taken only where the
    grid gives every CU two blocks — the shapes of the headline's large maps, which the small check shapes never reach:
    grids of exactly 512 blocks with an ODD number of 16-channel chunks (3 / 5: both halo buffers of the AD form end a
    K loop), statistics rows, upsampled input, accumulation, prologue, and the pooled data gradient on 32-pixel-wide tiles"""
    from sivae_hip import lib
    L = lib.load()
    res = []
    for (B, Ci, Co, H, W, ks), tpx in (((16, 48, 64, 128, 128, 3), 512), ((16, 80, 256, 64, 64, 3), 256)):
        assert L.sivae_bf16_conv2d_num_px_tiles(B, Co, H, W, ks) == B * H * W // tpx, "not the big pixel tile"
    res += check_conv((16, 48, 64, 128, 128, 3), stats=True)
    res += check_conv_fused((16, 80, 256, 64, 64, 3))
    res += check_conv_dgrad_pool((16, 64, 48, 128, 128, 3))
    return res


BN16_SHAPES = [(4, 64, 16, 16), (2, 24, 8, 12), (8, 512, 4, 4), (3, 128, 32, 32)]


def _bn_params(C, x):
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    gamma, beta = _rand(C, seed=7, scale=0.5) + 1.0, _rand(C, seed=8, scale=0.2)
    return [t.float().double() for t in (mean, invstd, gamma, beta)]


def check_bn(shape, res_mode, pool):
    """res_mode: 0 none, 1 same-resolution residual, 2 half-resolution residual (upsample addressing)"""
    from sivae_hip import ops16
    B, C, H, W = shape
    x = _r16(_rand(B, C, H, W, seed=1))
    mean, invstd, gamma, beta = _bn_params(C, x)
    r = None
    if res_mode == 1:
        r = _r16(_rand(B, C, H, W, seed=2))
    elif res_mode == 2:
        r = _r16(_rand(B, C, H // 2, W // 2, seed=2))
    rfull = None if r is None else (r if res_mode == 1 else F.interpolate(r, scale_factor=2, mode="nearest"))
    sc = (invstd.float() * gamma.float())
    sh = beta.float() - mean.float() * sc
    z = (x.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double()
    if rfull is not None:
        z = z + rfull
    yref = F.leaky_relu(z, SLOPE)
    dev = [t.float().to(DEV) for t in (mean, invstd, gamma, beta)]
    y, yp, mask = ops16.bn_apply_act(to_blocked(x).to(DEV), None if r is None else to_blocked(r).to(DEV), *dev, C,
                                     res_up=res_mode == 2, want_full=True, pool=pool, want_mask=True)
    tag = "bn16%s res%d%s" % (shape, res_mode, "+pool" if pool else "")
    res = [(tag, _err(from_blocked(y, C), yref), TOL_BF16)]
    # the sign mask: bit e of byte (b, cb, h, w) <=> output channel 8*cb + e > 0; padded channels 0
    yb = y.detach().cpu().float()
    bits = ((yb > 0).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(-1)
    res.append((tag + " signmask", float((bits - mask.cpu().to(torch.int32)).abs().max()), 0.0))
    if pool:
        res.append((tag + " pooled", _err(from_blocked(yp, C), F.avg_pool2d(from_blocked(y, C), 2)), TOL_BF16))
    # ---- backward
    yr = from_blocked(y, C)  # the kernel takes the activation sign from ITS rounded output
    for dy_pooled, dz_sum in ((False, False), (True, False), (False, True)):
        dyf = _r16(_rand(B, C, H // 2, W // 2, seed=3) if dy_pooled else _rand(B, C, H, W, seed=3))
        dy_full = F.interpolate(dyf, scale_factor=2, mode="nearest") * 0.25 if dy_pooled else dyf
        g = dy_full * torch.where(yr > 0, 1.0, SLOPE)
        xh = (x - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        sg, sgx = g.sum((0, 2, 3)), (g * xh).sum((0, 2, 3))
        n = B * H * W
        dxr = (gamma * invstd).view(1, -1, 1, 1) * (g - (sg / n).view(1, -1, 1, 1) - xh * (sgx / n).view(1, -1, 1, 1))
        dx, dz, dg, db = ops16.bn_bwd(to_blocked(dyf).to(DEV), y, to_blocked(x).to(DEV), *dev, C, dy_pooled=dy_pooled,
                                      want_dz=True, dz_sum=dz_sum)
        # the mask-based backward must reproduce the output-based one bit for bit
        dxm, dzm, dgm, dbm = ops16.bn_bwd(to_blocked(dyf).to(DEV), mask, to_blocked(x).to(DEV), *dev, C,
                                          dy_pooled=dy_pooled, want_dz=True, dz_sum=dz_sum)
        same = torch.equal(dx, dxm) and torch.equal(dz, dzm) and torch.equal(dg, dgm) and torch.equal(db, dbm)
        res.append((tag + (" pooled-dy" if dy_pooled else (" dzsum" if dz_sum else "")) + " bwd mask == bwd output",
                    0.0 if same else float("inf"), 0.0))
        t2 = tag + (" bwd pooled-dy" if dy_pooled else (" bwd dzsum" if dz_sum else " bwd"))
        res.append((t2 + " dx", _err(from_blocked(dx, C), dxr), 1.2e-2))
        dzr = F.avg_pool2d(g, 2) * 4 if dz_sum else g
        res.append((t2 + " dz", _err(from_blocked(dz, C), dzr), TOL_BF16))
        res.append((t2 + " dgamma", _err(dg, sgx), 1e-4))
        res.append((t2 + " dbeta", _err(db, sg), 1e-4))
    if res_mode == 0:
        # sign recomputed from x (no saved output): stem / BatchNorm-1 form
        zz = (x - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
        # (drop elements within rounding of the kink: the recomputed sign may legitimately differ there)
        dyf = _r16(_rand(B, C, H, W, seed=3))
        g = dyf * torch.where(zz > 0, 1.0, SLOPE)
        xh = (x - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        sg, sgx = g.sum((0, 2, 3)), (g * xh).sum((0, 2, 3))
        n = B * H * W
        dxr = (gamma * invstd).view(1, -1, 1, 1) * (g - (sg / n).view(1, -1, 1, 1) - xh * (sgx / n).view(1, -1, 1, 1))
        dx, _, dg, db = ops16.bn_bwd(to_blocked(dyf).to(DEV), None, to_blocked(x).to(DEV), *dev, C)
        near = zz.abs() < 1e-4
        d = (from_blocked(dx, C) - dxr).abs()
        d[near] = 0
        res.append((tag + " bwd recomputed-sign dx", float(d.max() / dxr.abs().max()), 1.2e-2))
        res.append((tag + " bwd recomputed-sign dbeta", _err(db, sg), 2e-3))
    return res


def check_bn_fused16():
    """the one-pass BatchNorm backward of the bf16 mode (bf16_bn_fused.hip) against the three-launch form on the same
    inputs: barrier-free form (plane sets of one block), persistent half-grids, several groups, the one-grid form; every
    sign source, pooled dy, dz at full resolution and as block sums.  The partial sums are fp32 in both forms but folded
    in different orders: dgamma / dbeta to fp32 rounding of long sums, dx / dz to one bf16 rounding."""
    from sivae_hip import ops, ops16
    res = []
    L = ops._lib.load()
    for (B, C, H, W) in [(4, 64, 16, 16), (8, 520, 4, 4), (16, 64, 64, 64), (6, 300, 32, 32), (64, 16, 128, 128)]:
        tag = "bn16_fused(%d,%d,%d,%d)" % (B, C, H, W)
        assert L.sivae_bf16_bn_bwd_fused_supported(B, C, H, W) == 1, tag
        x = _r16(_rand(B, C, H, W, seed=1))
        r = _r16(_rand(B, C, H, W, seed=2))
        mean, invstd, gamma, beta = _bn_params(C, x)
        dev = [t.float().to(DEV) for t in (mean, invstd, gamma, beta)]
        xb, rb = to_blocked(x).to(DEV), to_blocked(r).to(DEV)
        y, _, mask = ops16.bn_apply_act(xb, rb, *dev, C, want_full=True, want_mask=True)
        dy = to_blocked(_r16(_rand(B, C, H, W, seed=3))).to(DEV)
        dyh = to_blocked(_r16(_rand(B, C, H // 2, W // 2, seed=4))).to(DEV)
        cases = [("mask", lambda: ops16.bn_bwd(dy, mask, xb, *dev, C, want_dz=True)),
                 ("mask dzsum", lambda: ops16.bn_bwd(dy, mask, xb, *dev, C, want_dz=True, dz_sum=True)),
                 ("mask pooled", lambda: ops16.bn_bwd(dyh, mask, xb, *dev, C, dy_pooled=True, want_dz=True)),
                 ("output", lambda: ops16.bn_bwd(dy, y, xb, *dev, C, want_dz=True)),
                 ("recomputed", lambda: ops16.bn_bwd(dy, None, xb, *dev, C)),
                 ("recomputed pooled", lambda: ops16.bn_bwd(dyh, None, xb, *dev, C, dy_pooled=True))]
        for name, fn in cases:
            outs = []
            for fused in (False, True):
                ops.BN_FUSED = fused
                try:
                    outs.append(fn())
                finally:
                    ops.BN_FUSED = True
            a, b = outs
            res.append(("%s %s dx" % (tag, name), _err(b[0].float(), a[0].float()), 1e-2))
            if a[1] is not None:
                res.append(("%s %s dz" % (tag, name), _err(b[1].float(), a[1].float()), 1e-2))
            res.append(("%s %s dgamma" % (tag, name), _err(b[2], a[2]), 2e-4))
            res.append(("%s %s dbeta" % (tag, name), _err(b[3], a[3]), 2e-4))
            # most elements agree bit for bit (the coefficients differ in their last bits only)
            same = float((b[0].float() == a[0].float()).float().mean())
            res.append(("%s %s dx mostly identical" % (tag, name), 1.0 - same, 0.02))
    return res


def check_bn_seg16():
    """SEGMENTED batches of the bf16 BatchNorm kernels (sivae_bf16_bn_apply_act_seg / sivae_bf16_bn_bwd_fused_seg and the
    per-segment calls of the three-launch form): two passes with their own batch statistics laid end to end must give
    what the two separate calls give — outputs, sign mask, pooled output, dx and dz BIT FOR BIT (the same arithmetic per
    element, the same partial-sum order per (segment, channel block)); dgamma / dbeta are the two passes' sums added in
    fp64 before the one rounding (separate calls round twice): 1e-6."""
    from sivae_hip import ops, ops16
    res = []
    for (Bs, C, H, W) in [(4, 64, 16, 16), (8, 520, 4, 4), (16, 64, 64, 64), (3, 40, 8, 8), (32, 16, 128, 128)]:
        tag = "bn16_seg(2x%d,%d,%d,%d)" % (Bs, C, H, W)
        xs = [_r16(_rand(Bs, C, H, W, seed=1)), _r16(_rand(Bs, C, H, W, seed=11) * 1.7 + 0.3)]
        rs = [_r16(_rand(Bs, C, H, W, seed=2)), _r16(_rand(Bs, C, H, W, seed=12))]
        rh = [_r16(_rand(Bs, C, H // 2, W // 2, seed=5)), _r16(_rand(Bs, C, H // 2, W // 2, seed=15))]
        prm = [_bn_params(C, x) for x in xs]
        gamma, beta = prm[0][2].float().to(DEV), prm[0][3].float().to(DEV)
        mean = [p[0].float().to(DEV) for p in prm]
        invstd = [p[1].float().to(DEV) for p in prm]
        mean2, invstd2 = torch.cat(mean), torch.cat(invstd)
        xb = [to_blocked(x).to(DEV) for x in xs]
        rb = [to_blocked(r).to(DEV) for r in rs]
        rhb = [to_blocked(r).to(DEV) for r in rh]
        x2, r2, rh2 = torch.cat(xb), torch.cat(rb), torch.cat(rhb)
        dy = [to_blocked(_r16(_rand(Bs, C, H, W, seed=3 + 10 * g))).to(DEV) for g in range(2)]
        dyh = [to_blocked(_r16(_rand(Bs, C, H // 2, W // 2, seed=4 + 10 * g))).to(DEV) for g in range(2)]
        dy2, dyh2 = torch.cat(dy), torch.cat(dyh)

        def same(name, got, parts):
            ok = all(torch.equal(got[g * Bs:(g + 1) * Bs], parts[g]) for g in range(2))
            res.append(("%s %s" % (tag, name), 0.0 if ok else float("inf"), 0.0))

        # ---- apply: plain / residual + mask / half-resolution residual + pool
        for name, kw, rr in (("apply", dict(), (None, None, None)),
                             ("apply res+mask+pool", dict(want_mask=True, pool=True), (rb[0], rb[1], r2)),
                             ("apply res_up", dict(res_up=True, want_mask=True), (rhb[0], rhb[1], rh2))):
            sep = [ops16.bn_apply_act(xb[g], rr[g], mean[g], invstd[g], gamma, beta, C, **kw) for g in range(2)]
            seg = ops16.bn_apply_act(x2, rr[2], mean2, invstd2, gamma, beta, C, nseg=2, **kw)
            for i, part in enumerate(seg):
                if part is not None:
                    same("%s out%d" % (name, i), part, [sep[0][i], sep[1][i]])
        masks = [ops16.bn_apply_act(xb[g], rb[g], mean[g], invstd[g], gamma, beta, C, want_mask=True) for g in range(2)]
        y = [m[0] for m in masks]
        mask = [m[2] for m in masks]
        y2, mask2 = torch.cat(y), torch.cat(mask)
        # ---- backward: every sign source, pooled dy, dz full / block sums; one-launch and three-launch forms
        cases = [("mask", lambda d, sg, yy, mk, xx, mu, iv, n: ops16.bn_bwd(d[0], mk, xx, mu, iv, gamma, beta, C, want_dz=True, nseg=n)),
                 ("mask dzsum", lambda d, sg, yy, mk, xx, mu, iv, n: ops16.bn_bwd(d[0], mk, xx, mu, iv, gamma, beta, C, want_dz=True, dz_sum=True, nseg=n)),
                 ("mask pooled", lambda d, sg, yy, mk, xx, mu, iv, n: ops16.bn_bwd(d[1], mk, xx, mu, iv, gamma, beta, C, dy_pooled=True, want_dz=True, nseg=n)),
                 ("output", lambda d, sg, yy, mk, xx, mu, iv, n: ops16.bn_bwd(d[0], yy, xx, mu, iv, gamma, beta, C, want_dz=True, nseg=n)),
                 ("recomputed", lambda d, sg, yy, mk, xx, mu, iv, n: ops16.bn_bwd(d[0], None, xx, mu, iv, gamma, beta, C, nseg=n)),
                 ("recomputed pooled", lambda d, sg, yy, mk, xx, mu, iv, n: ops16.bn_bwd(d[1], None, xx, mu, iv, gamma, beta, C, dy_pooled=True, nseg=n))]
        for fused in (True, False):
            ops.BN_FUSED = fused
            try:
                for name, fn in cases:
                    sep = [fn((dy[g], dyh[g]), g, y[g], mask[g], xb[g], mean[g], invstd[g], 1) for g in range(2)]
                    seg = fn((dy2, dyh2), None, y2, mask2, x2, mean2, invstd2, 2)
                    nm = "%s %s" % ("fused" if fused else "3-launch", name)
                    same(nm + " dx", seg[0], [sep[0][0], sep[1][0]])
                    if seg[1] is not None:
                        same(nm + " dz", seg[1], [sep[0][1], sep[1][1]])
                    res.append(("%s %s dgamma" % (tag, nm), _err(seg[2], sep[0][2].double() + sep[1][2].double()), 1e-6))
                    res.append(("%s %s dbeta" % (tag, nm), _err(seg[3], sep[0][3].double() + sep[1][3].double()), 1e-6))
            finally:
                ops.BN_FUSED = True
        # the per-channel arrival counters are back at zero: a second segmented call gives the same parameter gradients
        a1 = ops16.bn_bwd(dy2, mask2, x2, mean2, invstd2, gamma, beta, C, nseg=2)
        a2 = ops16.bn_bwd(dy2, mask2, x2, mean2, invstd2, gamma, beta, C, nseg=2)
        res.append((tag + " repeat dgamma/dbeta", 0.0 if (torch.equal(a1[2], a2[2]) and torch.equal(a1[3], a2[3])) else float("inf"), 0.0))
    return res


def check_eltwise16():
    from sivae_hip import ops16
    res = []
    x = _r16(_rand(3, 40, 6, 10, seed=1))
    xb = to_blocked(x).to(DEV)
    res.append(("upsample16_fwd", _err(from_blocked(ops16.upsample2_fwd(xb, 40), 40),
                                       F.interpolate(x, scale_factor=2, mode="nearest")), 0.0))
    dy = _r16(_rand(3, 40, 12, 20, seed=2))
    res.append(("upsample16_bwd", _err(from_blocked(ops16.upsample2_bwd(to_blocked(dy).to(DEV), 40), 40),
                                       F.avg_pool2d(dy, 2) * 4), TOL_BF16))
    a, b = _r16(_rand(2, 64, 8, 8, seed=3)), _r16(_rand(2, 64, 8, 8, seed=4))
    ab = to_blocked(a).to(DEV)
    ops16.add_(ab, to_blocked(b).to(DEV))
    res.append(("add16", _err(from_blocked(ab, 64), a + b), TOL_BF16))
    return res


def check_splitk16():
    """split-K bf16 conv (small grids): same numbers as the single-pass kernel incl. per-image stats, accumulate"""
    from sivae_hip import lib, ops16
    L = lib.load()
    res = []
    for shape in [(16, 512, 512, 4, 4, 3), (32, 512, 512, 8, 8, 3), (8, 256, 384, 4, 4, 3), (3, 208, 72, 8, 8, 3)]:
        B, Ci, Co, H, W, ks = shape
        S = L.sivae_bf16_conv2d_splitk(B, Ci, Co, H, W, ks)
        res.append(("splitk16%s slices=%d" % (shape, S), 0.0 if S > 1 else float("inf"), 0.5))
        res += check_conv(shape, stats=True)
        res += check_conv_dgrad(shape)
    res += check_conv_fused((4, 256, 128, 8, 8, 3))
    res.append(("splitk16 off on large grids", float(L.sivae_bf16_conv2d_splitk(128, 128, 128, 64, 64, 3) != 1), 0.0))
    return res


def check_kwpack(C, Cw, H, W, B=2):
    """kw-packed form of the RGB-side 5x5 layers (C <= 3 image channels, Cw feature channels), each piece against the
    plain 5x5 op of torch on the same bf16-rounded operands: im2col / fold kernels, stem forward / weight gradient /
    input gradient, predict forward / weight gradient / input gradient — through the SAME virtual-weight and
    gradient-unpacking helpers the autograd blocks use (functional16._virtual / _unpack_dw)."""
    from sivae_hip import functional16 as SF16
    from sivae_hip import ops16
    res = []
    tag = "kwpack(C=%d,Cw=%d,%dx%d) " % (C, Cw, H, W)
    img = _rand(B, C, H, W, seed=1)
    # ---- layout kernels against their definition
    for sgn in (+1, -1):
        ref = torch.zeros(B, 16, H, W, dtype=torch.float64)
        for kw in range(5):
            for w_ in range(W):
                ws = w_ + sgn * (kw - 2)
                if 0 <= ws < W:
                    ref[:, kw * C:(kw + 1) * C, :, w_] = img[:, :, :, ws]
        got = ops16.im2col_kw5(img.float().to(DEV), sgn)
        res.append((tag + "im2col sgn%+d" % sgn, _err(from_blocked(got, 16), _r16(ref)), 0.0))
        g = _rand(B, 5 * C, H, W, seed=2).float()
        bias = _rand(C, seed=3).float()
        reff = bias.double().view(1, C, 1, 1).expand(B, C, H, W).clone()
        for kw in range(5):
            for w_ in range(W):
                ws = w_ + sgn * (kw - 2)
                if 0 <= ws < W:
                    reff[:, :, :, w_] += g.double()[:, kw * C:(kw + 1) * C, :, ws]
        got = ops16.fold_kw5(g.to(DEV), bias.to(DEV), C, sgn)
        res.append((tag + "fold sgn%+d" % sgn, _err(got, reff), 1e-6))
    # ---- stem (C -> Cw): forward, weight gradient, input gradient
    w = _rand(Cw, C, 5, 5, seed=4, scale=1.0 / (C * 25) ** 0.5)
    wd = w.float().to(DEV)
    x16 = _r16(img)
    xk = ops16.im2col_kw5(img.float().to(DEV), +1)
    y = ops16.conv2d(xk, ops16.PackedW16(SF16._virtual(wd, "in"), 0), 5 * C, Cw, ops16.KS51)
    res.append((tag + "stem fwd", _err16(from_blocked(y, Cw), F.conv2d(x16, _r16(w), padding=2)), TOL_BF16))
    da = _r16(_rand(B, Cw, H, W, seed=5))
    dab = to_blocked(da).to(DEV)
    dw = SF16._unpack_dw(ops16.conv2d_wgrad(xk, dab, 5 * C, Cw, ops16.KS51), Cw, C, "in")
    res.append((tag + "stem wgrad", _err(dw, torch.nn.grad.conv2d_weight(x16, (Cw, C, 5, 5), da, padding=2)), TOL_F32))
    gk = ops16.conv2d(dab, ops16.PackedW16(SF16._virtual(wd, "in_d"), 0), Cw, 5 * C, ops16.KS51, out_f32=True)
    dx = ops16.fold_kw5(gk, None, C, -1)
    res.append((tag + "stem dgrad", _err(dx, F.conv_transpose2d(da, _r16(w), padding=2)), TOL_F32))
    # ---- predict (Cw -> C): forward (+ bias, fp32 out), weight gradient, input gradient
    wp = _rand(C, Cw, 5, 5, seed=6, scale=1.0 / (Cw * 25) ** 0.5)
    wpd = wp.float().to(DEV)
    bias = _rand(C, seed=7).float()
    xf = _r16(_rand(B, Cw, H, W, seed=8))
    xfb = to_blocked(xf).to(DEV)
    yk = ops16.conv2d(xfb, ops16.PackedW16(SF16._virtual(wpd, "out"), 0), Cw, 5 * C, ops16.KS51, out_f32=True)
    yp = ops16.fold_kw5(yk, bias.to(DEV), C, +1)
    res.append((tag + "predict fwd", _err(yp, F.conv2d(xf, _r16(wp), bias.double(), padding=2)), TOL_F32))
    dy = _rand(B, C, H, W, seed=9)
    dyk = ops16.im2col_kw5(dy.float().to(DEV), -1)
    dwp = SF16._unpack_dw(ops16.conv2d_wgrad(xfb, dyk, Cw, 5 * C, ops16.KS51), C, Cw, "out")
    res.append((tag + "predict wgrad",
                _err(dwp, torch.nn.grad.conv2d_weight(xf, (C, Cw, 5, 5), _r16(dy), padding=2)), TOL_F32))
    dxp = ops16.conv2d(dyk, ops16.PackedW16(SF16._virtual(wpd, "out_d"), 0), 5 * C, Cw, ops16.KS51)
    res.append((tag + "predict dgrad",
                _err16(from_blocked(dxp, Cw), F.conv_transpose2d(_r16(dy), _r16(wp), padding=2)), TOL_BF16))
    return res


def all_checks():
    checks = [("convert16", check_convert)]
    for cfg in [(3, 64, 32, 32), (3, 32, 24, 40), (1, 64, 28, 28), (2, 48, 9, 7)]:
        checks.append(("kwpack%s" % (cfg,), lambda cfg=cfg: check_kwpack(*cfg)))
    for s in CONV16_SHAPES:
        checks.append(("conv16%s" % (s,), lambda s=s: check_conv(s)))
        checks.append(("dgrad16%s" % (s,), lambda s=s: check_conv_dgrad(s)))
        checks.append(("wgrad16%s" % (s,), lambda s=s: check_conv_wgrad(s)))
    # pooled data gradient: 32-wide tiles (row pairs = accumulator tiles), 16- and 8-wide maps (row pairs = lanes), the
    # 32 / 64 / 128-channel tile configurations, big pixel tiles (B = 16 at 64x64), ragged tiles, padded channels
    for sh in [(2, 64, 64, 64, 64, 3), (16, 64, 64, 64, 64, 3), (2, 128, 64, 32, 32, 3), (3, 256, 128, 16, 16, 3),
               (4, 160, 136, 8, 8, 3), (2, 24, 16, 12, 20, 3), (2, 64, 128, 48, 40, 3), (8, 128, 128, 32, 32, 3)]:
        checks.append(("dgrad16_pool%s" % (sh,), lambda sh=sh: check_conv_dgrad_pool(sh)))
    checks.append(("conv16_bias_f32out", lambda: check_conv((2, 64, 3, 32, 32, 5), bias=True, out_f32=True)
                   + check_conv((2, 64, 15, 32, 32, 51), out_f32=True)
                   + check_conv((2, 64, 3, 16, 16, 5), bias=True)))
    checks.append(("conv16_stats", lambda: check_conv((3, 64, 128, 32, 32, 3), stats=True)
                   + check_conv((3, 64, 64, 32, 32, 3), stats=True) + check_conv((3, 24, 40, 12, 12, 3), stats=True)
                   + check_conv((5, 32, 96, 4, 4, 3), stats=True)))
    for s in [(2, 64, 128, 32, 32, 3), (2, 128, 64, 16, 16, 1), (3, 64, 64, 8, 8, 3), (2, 24, 40, 12, 20, 3)]:
        checks.append(("conv16_fused%s" % (s,), lambda s=s: check_conv_fused(s)))
    checks.append(("conv16_big_tiles", check_conv_big_tiles))
    for s in [(2, 64, 128, 32, 32, 3), (3, 24, 40, 12, 20, 3), (4, 64, 64, 8, 8, 3)]:
        checks.append(("wgrad16_fused%s" % (s,), lambda s=s: check_conv_wgrad(s, pro=True)
                       + check_conv_wgrad(s, upsample=True)))
    for s in BN16_SHAPES:
        for rm in (0, 1, 2):
            checks.append(("bn16%s res%d" % (s, rm), lambda s=s, rm=rm: check_bn(s, rm, pool=rm != 2)))
    checks.append(("bn16_fused", check_bn_fused16))
    checks.append(("bn16_seg", check_bn_seg16))
    checks.append(("eltwise16", check_eltwise16))
    checks.append(("splitk16", check_splitk16))
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
                rows.append("%-4s %-66s err=%.3e tol=%.1e" % ("ok" if ok else "FAIL", name, err, tol))
        except Exception:  # noqa: BLE001
            nfail += 1
            rows.append("EXC  %s\n%s" % (label, traceback.format_exc(limit=4)))
        torch.cuda.synchronize()
    print("\n".join(rows))
    print("kernel_checks16: %d failures of %d" % (nfail, len(rows)))
    return nfail


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
