"""bf16-mode twins of the differentiable blocks in `functional.py` (config 3 of BASELINE.json, build-defined: the
reference has no mixed precision).

Same module granularity and the same saved-tensor policy as the fp32 blocks (one autograd node per ResidualBlock /
stem / predict; BatchNorm-1's output is never written, its LeakyReLU sign is recomputed in the backward), but every
activation and activation gradient between the blocks is a blocked bf16 tensor [B, C16/8, H, W, 8] (`ops16`), the convs
run on the bf16 matrix pipe with fp32 accumulation, and BatchNorm statistics / all parameter gradients stay fp32 (they
land in the same flat fp32 gradient buffers the fused Adam and the RCCL all-reduce use).  The two Linear layers, the
sampler and the loss kernels keep their fp32 kernels: `to_blocked` / `from_blocked` are the differentiable layout
changes at those seams (reference call sites: train_soft_intro_vae.py:116-122 encoder fc, :161-168 decoder fc/view).
"""
import os

import torch

from . import functional as SF
from . import ops
from . import ops16

SLOPE = SF.SLOPE
# h = LeakyReLU(BN1(conv1)) is WRITTEN (one bf16 pass) and kept for the backward in this mode: the producer-BatchNorm
# prologue costs the bf16 conv / weight-gradient kernels more (unpack + affine + LeakyReLU + repack of every staged vector
# between two short MFMA phases: conv2 137 vs 77 us, its weight gradient 190 vs 96 us per launch at 128x128 bs128) than
# the extra half-width pass (~30 us).  In fp32 the same trade was a wash (functional.MATERIALIZE_H).
# SIVAE_BF16_MATERIALIZE_H=0 keeps the fused prologue (A/B measurements).
MATERIALIZE_H = os.environ.get("SIVAE_BF16_MATERIALIZE_H", "1") != "0"
# 1-bit LeakyReLU sign mask for the block-output BatchNorm (SIVAE_BF16_SIGNMASK=0: the backward re-reads the output)
SIGNMASK = os.environ.get("SIVAE_BF16_SIGNMASK", "1") != "0"


# kw-packed form of the RGB-side 5x5 layers (ops16.im2col_kw5 / fold_kw5, ks code 51): SIVAE_BF16_KWPACK=0 runs them
# as padded 25-tap convs (A/B measurements)
KWPACK = os.environ.get("SIVAE_BF16_KWPACK", "1") != "0"


def _virtual(w, virt):
    """the [*, *, 5, 1] weight of the 5-tap conv that stands for a 5x5 layer with <= 3 channels on one side
    (W = w[co][ci][kh][kw];  j = kw*C + c with c the index of the narrow side)"""
    if virt == "in":        # narrow input, forward:        V[co][j][kh]  = W[co][ci][kh][kw]
        v = w.permute(0, 3, 1, 2).reshape(w.shape[0], 5 * w.shape[1], 5, 1)
    elif virt == "in_d":    # narrow input, data gradient:  V[j][co][kh'] = W[co][ci][4-kh'][kw]
        v = w.flip(2).permute(3, 1, 0, 2).reshape(5 * w.shape[1], w.shape[0], 5, 1)
    elif virt == "out":     # narrow output, forward:       V[j][ci][kh]  = W[co][ci][kh][kw]
        v = w.permute(3, 0, 1, 2).reshape(5 * w.shape[0], w.shape[1], 5, 1)
    elif virt == "out_d":   # narrow output, data gradient: V[ci][j][kh'] = W[co][ci][4-kh'][kw]
        v = w.flip(2).permute(1, 3, 0, 2).reshape(w.shape[1], 5 * w.shape[0], 5, 1)
    else:
        raise ValueError(virt)
    return v.contiguous()


def _unpack_dw(dwv, Co, Ci, virt, out=None):
    """gradient of the virtual weight ([.., .., 5, 1] from the ks-51 weight-gradient kernel) -> [Co][Ci][kh][kw]
    (a fresh tensor, or copied into `out`: the parameter's gradient slab)"""
    if virt == "in":    # [Co][kw*Ci + ci][kh]
        v = dwv.view(Co, 5, Ci, 5).permute(0, 2, 3, 1)
    else:               # "out": [kw*Co + co][Ci][kh]
        v = dwv.view(5, Co, Ci, 5).permute(1, 2, 3, 0)
    if out is None:
        return v.contiguous()
    out.copy_(v)
    return out


def packed16(w, mode, virt=None):
    """bf16 operand slabs of a master weight, cached ON the parameter (dies with it) and rebuilt when it changes"""
    tag = (w._version, getattr(w, "_sivae_gen", 0), w.data_ptr(), SF.cache_epoch())
    store = w.__dict__.setdefault("_sivae_pack16", {})
    hit = store.get((mode, virt))
    if hit is not None and hit[0] == tag:
        return hit[1]
    wp = ops16.PackedW16(w.detach() if virt is None else _virtual(w.detach(), virt), mode)
    store[(mode, virt)] = (tag, wp)
    return wp


PACK_FORM_BF16 = 6  # (include/sivae_hip.h: operand form 6 of sivae_pack_job_fill / sivae_pack_batch)


def repack16(params, owner):
    """The bf16 mode's part of `functional.repack`: after `params` were updated in place (FlatAdam.step, generation already
    bumped) every bf16 operand slab cached on them is rebuilt IN PLACE by ONE launch (sivae_pack_batch, form 6) and its
    cache entry re-validated.  Slabs of the virtual [*, *, 5, 1] weights (permuted copies of the RGB-side 5x5 layers) and
    slabs whose source alias no longer is the parameter's storage are dropped and rebuilt on demand.  The job table lives
    on `owner` and is rebuilt only when the set of cached slabs changes (the first iterations)."""
    import ctypes
    if not SF.PACK_BATCH:
        return
    L = ops._lib.load()
    entries = []
    for p in params:
        store = p.__dict__.get("_sivae_pack16")
        if not store:
            continue
        for slot, (tag, obj) in list(store.items()):
            if slot[1] is not None or obj.w.data_ptr() != p.data_ptr() or obj.w.shape != p.shape or tag[2] != p.data_ptr():
                del store[slot]
                continue
            entries.append((p, store, slot, obj))
    if not entries:
        return
    key = tuple((obj.w.data_ptr(), obj.data.data_ptr(), obj.mode) for _, _, _, obj in entries)
    plan = owner.__dict__.get("_sivae_pack16_plan")
    if plan is None or plan["key"] != key:
        if torch.cuda.is_current_stream_capturing() or len(entries) > 32767:
            return  # (uploading a job table is not capturable: the slabs stay invalid and are rebuilt one by one on use)
        jb = L.sivae_pack_job_bytes()
        host = ctypes.create_string_buffer(jb * len(entries))
        block_job, nblocks = [], 0
        for i, (_, _, _, obj) in enumerate(entries):
            nb = L.sivae_pack_job_fill(host, i, PACK_FORM_BF16, ctypes.c_void_p(obj.w.data_ptr()),
                                       ctypes.c_void_p(obj.data.data_ptr()), obj.Co, obj.Ci, obj.ks, obj.mode, nblocks)
            if nb <= 0:
                raise ops._lib.SivaeError("sivae_pack_job_fill", nb)
            block_job.extend([i] * nb)
            nblocks += nb
        dev = entries[0][0].device
        jt = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
        bj = torch.tensor(block_job, dtype=torch.int16).to(dev)
        prev = owner.__dict__.get("_sivae_pack16_plan")
        if prev is not None:  # (a HIP graph captured while an earlier table was current still launches with it)
            old = owner.__dict__.setdefault("_sivae_pack16_plan_old", [])
            old.append(prev)
            del old[:-4]
        plan = {"key": key, "jt": jt, "bj": bj, "nblocks": nblocks, "keep": [obj.data for _, _, _, obj in entries]}
        owner.__dict__["_sivae_pack16_plan"] = plan
    ops._lib.call("sivae_pack_batch", PACK_FORM_BF16, ops._p(plan["jt"]), ops._p(plan["bj"]), plan["nblocks"], ops._s(plan["jt"]))
    for p, store, slot, obj in entries:
        store[slot] = ((p._version, getattr(p, "_sivae_gen", 0), p.data_ptr(), SF.cache_epoch()), obj)


def _kwpack_ok(w, narrow):
    return KWPACK and w.dim() == 4 and tuple(w.shape[2:]) == (5, 5) and 5 * narrow <= 16


class ToBlockedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.C = x.shape[1]
        return ops16.from_f32(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops16.to_f32(g.contiguous(), ctx.C)


class FromBlockedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xb, C):
        return ops16.to_f32(xb, C)

    @staticmethod
    def backward(ctx, g):
        return ops16.from_f32(g.contiguous()), None


def to_blocked(x):
    return ToBlockedFn.apply(x)


def from_blocked(xb, C):
    return FromBlockedFn.apply(xb, C)


class ResBlockFn16(torch.autograd.Function):
    """ResidualBlock.forward (train_soft_intro_vae.py:65-75) on blocked bf16 activations; arguments as
    functional.ResBlockFn (x_up: x is stored at half resolution and stands for Upsample(2)(x); nseg > 1: a SEGMENTED
    batch — nseg passes of the network laid end to end, one set of BatchNorm batch statistics per pass, the running
    buffers updated once per pass in pass order (seg_rev: last first); replay_update=False: a replay of a `cache_segment`
    view leaves the running buffers alone).  The bf16 convolutions take a segmented batch as it is: they have no
    BatchNorm prologue (MATERIALIZE_H), their statistics rows are in image order and never straddle two passes
    (ops.bn_stats_from_conv checks), and the weight gradients sum over the whole batch like the reference's two
    backward passes through one parameter do."""

    @staticmethod
    def forward(ctx, x, w_exp, w1, g1, b1, w2, g2, b2, st1, st2, post, cache=None, x_up=False, nseg=1, seg_rev=False,
                replay_update=True):
        SF._claim(ctx, ((1, w_exp), (2, w1), (3, g1), (4, b1), (5, w2), (6, g2), (7, b2)))
        if nseg > 1 and not MATERIALIZE_H:
            raise NotImplementedError("sivae_hip: segmented bf16 batches need SIVAE_BF16_MATERIALIZE_H=1 (the fused "
                                      "BatchNorm prologue of the bf16 convs has no per-segment form)")
        ctx.nseg = nseg
        B, Cib, Hs, Ws, _ = x.shape
        H, W = (2 * Hs, 2 * Ws) if x_up else (Hs, Ws)
        Cm, Ci, Co = w1.shape[0], w1.shape[1], w2.shape[0]
        ctx.x_up, ctx.post, ctx.has_exp = x_up, post, w_exp is not None
        ctx.training = st1.training and st2.training
        ctx.dims = (Ci, Cm, Co)
        tag = SF.cache_tag((w_exp, w1, g1, b1, w2, g2, b2))
        if cache is not None and cache.get("y") is not None and cache.get("tag") == tag:
            a, h, c, out, mean1, invstd1, mean2, invstd2, y = (cache[k] for k in (
                "a", "h", "c", "out", "mean1", "invstd1", "mean2", "invstd2", "y"))
            if replay_update:
                SF._replay_bn(st1, mean1, invstd1, (B // nseg) * H * W, nseg, seg_rev)
                SF._replay_bn(st2, mean2, invstd2, (B // nseg) * H * W, nseg, seg_rev)
            ctx.save_for_backward(x, a, h, c, out, mean1, invstd1, mean2, invstd2, w_exp, w1, g1, b1, w2, g2, b2)
            return y.view_as(y)
        if not replay_update:
            # (functional.ResBlockFn: a `cache_segment` view that cannot be replayed must not be recomputed with a
            # running-statistics update the reference never makes)
            raise RuntimeError("sivae_hip: replay_update=False needs a filled, current replay cache for this block")
        idt = x
        if w_exp is not None:
            idt = ops16.conv2d(x, packed16(w_exp, 0), Ci, Co, 1)  # (at half resolution with x_up: commutes with Upsample)
        if st1.training:
            a, p1 = ops16.conv2d(x, packed16(w1, 0), Ci, Cm, 3, want_stats=True, upsample=x_up)
        else:
            a, p1 = ops16.conv2d(x, packed16(w1, 0), Ci, Cm, 3, upsample=x_up), None
        mean1, invstd1 = SF._stats(p1, B, Cm, H * W, st1, nseg, seg_rev)
        if MATERIALIZE_H:
            h, _ = ops16.bn_apply_act(a, None, mean1, invstd1, g1.detach(), b1.detach(), Cm, SLOPE, nseg=nseg)
            pro1 = None
        else:
            h, pro1 = a, (mean1, invstd1, g1.detach(), b1.detach(), SLOPE)
        if st2.training:
            c, p2 = ops16.conv2d(h, packed16(w2, 0), Cm, Co, 3, pro=pro1, want_stats=True)
        else:
            c, p2 = ops16.conv2d(h, packed16(w2, 0), Cm, Co, 3, pro=pro1), None
        mean2, invstd2 = SF._stats(p2, B, Co, H * W, st2, nseg, seg_rev)
        pool = post == "pool"
        if ctx.training and SIGNMASK:
            # the backward takes the LeakyReLU sign from a 1-bit-per-element mask written here (`out` below IS that mask),
            # and a pooled block never writes its full-resolution output
            full, yp, out = ops16.bn_apply_act(c, idt, mean2, invstd2, g2.detach(), b2.detach(), Co, SLOPE,
                                               res_up=x_up, want_full=not pool, pool=pool, want_mask=True, nseg=nseg)
        else:
            full, yp = ops16.bn_apply_act(c, idt, mean2, invstd2, g2.detach(), b2.detach(), Co, SLOPE, res_up=x_up,
                                          want_full=True, pool=pool, nseg=nseg)
            out = full
        if pool:
            y = yp
        elif post == "up":
            y = ops16.upsample2_fwd(full, Co)
        else:
            y = full
        if cache is not None:
            cache.update(a=a, h=h, c=c, out=out, mean1=mean1, invstd1=invstd1, mean2=mean2, invstd2=invstd2, y=y,
                         tag=tag)
        ctx.save_for_backward(x, a, h, c, out, mean1, invstd1, mean2, invstd2, w_exp, w1, g1, b1, w2, g2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("sivae_hip: backward through eval-mode BatchNorm is not supported")
        x, a, h, c, out, mean1, invstd1, mean2, invstd2, w_exp, w1, g1, b1, w2, g2, b2 = ctx.saved_tensors
        k_we, k_w1, k_g1, k_b1, k_w2, k_g2, k_b2 = ctx.use
        h_saved = h.data_ptr() != a.data_ptr()
        need = ctx.needs_input_grad
        need_x, need_we, need_w1, need_bn1, need_w2, need_bn2 = need[0], need[1], need[2], need[3] or need[4], \
            need[5], need[6] or need[7]
        Ci, Cm, Co = ctx.dims
        x_up, nseg = ctx.x_up, ctx.nseg
        dy = dy.contiguous()
        pool = ctx.post == "pool"
        d_out = ops16.upsample2_bwd(dy, Co) if ctx.post == "up" else dy
        need_dz = need_x or (need_we and ctx.has_exp)
        pg2 = SF._pg_dst(g2, k_g2, b2, k_b2) if need_bn2 else None
        pg1 = SF._pg_dst(g1, k_g1, b1, k_b1) if need_bn1 else None
        dc, dz, dg2, db2 = ops16.bn_bwd(d_out, out, c, mean2, invstd2, g2, b2, Co, SLOPE, dy_pooled=pool,
                                        want_dz=need_dz and not x_up, dz_sum=need_dz and x_up,
                                        want_param_grads=need_bn2, pg_out=pg2, nseg=nseg)
        if pg2 is not None:
            SF._done(g2, b2)
        del d_out
        pro1 = None if h_saved else (mean1, invstd1, g1, b1, SLOPE)
        dw2 = ops16.conv2d_wgrad(h, dc, Cm, Co, 3, pro=pro1, out=SF._dst(w2, k_w2)) if need_w2 else None
        if need_w2 and k_w2 >= 0:
            SF._done(w2)
        dh = ops16.conv2d(dc, packed16(w2, 1), Co, Cm, 3)
        del dc
        # BatchNorm-1 + LeakyReLU: the sign is recomputed from a (x-hat * gamma + beta) even when h was stored — both
        # backward passes then read two tensors (dh, a) instead of three (2 of 7 tensor passes of this BatchNorm)
        da, _, dg1, db1 = ops16.bn_bwd(dh, None, a, mean1, invstd1, g1, b1, Cm, SLOPE, want_param_grads=need_bn1,
                                       pg_out=pg1, nseg=nseg)
        if pg1 is not None:
            SF._done(g1, b1)
        del dh
        dw1 = ops16.conv2d_wgrad(x, da, Ci, Cm, 3, upsample=x_up, out=SF._dst(w1, k_w1)) if need_w1 else None
        if need_w1 and k_w1 >= 0:
            SF._done(w1)
        dwe = None
        dx = None
        if need_x:
            if not x_up and not ctx.has_exp:
                # identity skip at the block's own resolution: conv1's data gradient is accumulated straight into the
                # skip gradient (one read-modify-write in the conv epilogue instead of a separate 3-pass add)
                dx = dz
                ops16.conv2d(da, packed16(w1, 1), Cm, Ci, 3, out=dx, accumulate=True)
            elif x_up and ops16.conv2d_pool_supported(da.shape[0], Cm, Ci, da.shape[2], da.shape[3], 3):
                # conv1 read x through upsample addressing: its data gradient is needed as 2x2 block sums only (the adjoint
                # of the deferred nn.Upsample) — summed in the conv's epilogue, and for an identity skip straight onto
                # the skip gradient (dz already holds ITS block sums: bn_bwd(dz_sum=True))
                if ctx.has_exp:
                    dx = ops16.conv2d_pool(da, packed16(w1, 1), Cm, Ci)
                    ops16.conv2d(dz, packed16(w_exp, 1), Co, Ci, 1, out=dx, accumulate=True)
                else:
                    dx = ops16.conv2d_pool(da, packed16(w1, 1), Cm, Ci, out=dz, accumulate=True)
            else:
                dx = ops16.conv2d(da, packed16(w1, 1), Cm, Ci, 3)
                if x_up:
                    dx = ops16.upsample2_bwd(dx, Ci)  # adjoint of the deferred nn.Upsample
                if ctx.has_exp:
                    ops16.conv2d(dz, packed16(w_exp, 1), Co, Ci, 1, out=dx, accumulate=True)
                else:
                    ops16.add_(dx, dz)
        if need_we and ctx.has_exp:
            dwe = ops16.conv2d_wgrad(x, dz, Ci, Co, 1, out=SF._dst(w_exp, k_we))
            if k_we >= 0:
                SF._done(w_exp)
        return (dx, dwe if k_we < 0 else None, dw1 if k_w1 < 0 else None,
                dg1 if (need[3] and pg1 is None) else None, db1 if (need[4] and pg1 is None) else None,
                dw2 if k_w2 < 0 else None,
                dg2 if (need[6] and pg2 is None) else None, db2 if (need[7] and pg2 is None) else None,
                None, None, None, None, None, None, None, None)


class StemFn16(torch.autograd.Function):
    """conv5x5 -> BatchNorm -> LeakyReLU -> AvgPool2d(2) (train_soft_intro_vae.py:88-93).  x: the fp32 NCHW image batch;
    out: blocked bf16.  With <= 3 image channels the conv runs kw-packed (5 taps over 5C channels)."""

    @staticmethod
    def forward(ctx, x, w, g, b, st, nseg=1, seg_rev=False):
        SF._claim(ctx, ((1, w), (2, g), (3, b)))
        ctx.nseg = nseg
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        ctx.kw = _kwpack_ok(w, Ci)
        if ctx.kw:
            xb, wp, ci_eff, ks = ops16.im2col_kw5(x.contiguous(), +1), packed16(w, 0, "in"), 5 * Ci, ops16.KS51
        else:
            xb, wp, ci_eff, ks = ops16.from_f32(x.contiguous()), packed16(w, 0), Ci, w.shape[2]
        if st.training:
            a, p = ops16.conv2d(xb, wp, ci_eff, Co, ks, want_stats=True)
        else:
            a, p = ops16.conv2d(xb, wp, ci_eff, Co, ks), None
        mean, invstd = SF._stats(p, B, Co, H * W, st, nseg, seg_rev)
        _, out = ops16.bn_apply_act(a, None, mean, invstd, g.detach(), b.detach(), Co, SLOPE, want_full=False,
                                    pool=True, nseg=nseg)
        ctx.training = st.training
        ctx.Ci = Ci
        ctx.save_for_backward(xb, a, mean, invstd, w, g, b)
        return out

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("sivae_hip: backward through eval-mode BatchNorm is not supported")
        xb, a, mean, invstd, w, g, b = ctx.saved_tensors
        need = ctx.needs_input_grad
        k_w, k_g, k_b = ctx.use
        Co, Ci, ks = w.shape[0], ctx.Ci, w.shape[2]
        pg = SF._pg_dst(g, k_g, b, k_b) if (need[2] or need[3]) else None
        da, _, dg, db = ops16.bn_bwd(dy.contiguous(), None, a, mean, invstd, g, b, Co, SLOPE, dy_pooled=True,
                                     want_param_grads=need[2] or need[3], pg_out=pg, nseg=ctx.nseg)
        if pg is not None:
            SF._done(g, b)
        dw = dx = None
        dst = SF._dst(w, k_w) if need[1] else None
        if ctx.kw:
            if need[1]:
                dw = _unpack_dw(ops16.conv2d_wgrad(xb, da, 5 * Ci, Co, ops16.KS51), Co, Ci, "in", out=dst)
            if need[0]:
                gk = ops16.conv2d(da, packed16(w, 0, "in_d"), Co, 5 * Ci, ops16.KS51, out_f32=True)
                dx = ops16.fold_kw5(gk, None, Ci, -1)
        else:
            if need[1]:
                dw = ops16.conv2d_wgrad(xb, da, Ci, Co, ks, out=dst)
            if need[0]:
                dx = ops16.to_f32(ops16.conv2d(da, packed16(w, 1), Co, Ci, ks), Ci)
        if dst is not None:
            SF._done(w)
        return (dx, dw if k_w < 0 else None, dg if (need[2] and pg is None) else None,
                db if (need[3] and pg is None) else None, None, None, None)


class PredictFn16(torch.autograd.Function):
    """Decoder.predict (conv5x5 + bias, train_soft_intro_vae.py:159): blocked bf16 in, fp32 NCHW out (the
    reconstruction feeds the fp32 loss kernels directly).  With <= 3 image channels the conv runs kw-packed (5 taps into
    5C fp32 channels, then the column fold adds the bias)."""

    @staticmethod
    def forward(ctx, x, w, bias, cache=None):
        SF._claim(ctx, ((1, w), (2, bias)))
        ctx.save_for_backward(x, w, bias)
        ctx.has_bias = bias is not None
        tag = SF.cache_tag((w, bias))
        if cache is not None and cache.get("y") is not None and cache.get("tag") == tag:
            return cache["y"].view_as(cache["y"])
        Co, Ci, ks = w.shape[0], w.shape[1], w.shape[2]
        bv = None if bias is None else bias.detach()
        if _kwpack_ok(w, Co):
            yk = ops16.conv2d(x, packed16(w, 0, "out"), Ci, 5 * Co, ops16.KS51, out_f32=True)
            y = ops16.fold_kw5(yk, bv, Co, +1)
        else:
            y = ops16.conv2d(x, packed16(w, 0), Ci, Co, ks, bias=bv, out_f32=True)
        if cache is not None:
            cache["y"], cache["tag"] = y, tag
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        k_w, k_b = ctx.use
        Co, Ci, ks = w.shape[0], w.shape[1], w.shape[2]
        dy = dy.contiguous()
        db = ops.channel_sum(dy, out=SF._dst(bias, k_b)) if (ctx.has_bias and need[2]) else None
        dw = dx = None
        dst = SF._dst(w, k_w) if need[1] else None
        if _kwpack_ok(w, Co):
            dyk = ops16.im2col_kw5(dy, -1)                                       # [kw*Co + co][h][w] = dy[co][h][w-kw+2]
            if need[1]:
                dw = _unpack_dw(ops16.conv2d_wgrad(x, dyk, Ci, 5 * Co, ops16.KS51), Co, Ci, "out", out=dst)
            if need[0]:
                dx = ops16.conv2d(dyk, packed16(w, 0, "out_d"), 5 * Co, Ci, ops16.KS51)
        else:
            dyb = ops16.from_f32(dy)
            dw = ops16.conv2d_wgrad(x, dyb, Ci, Co, ks, out=dst) if need[1] else None
            dx = ops16.conv2d(dyb, packed16(w, 1), Co, Ci, ks) if need[0] else None
        SF._done(w if dst is not None else None, bias if (db is not None and k_b >= 0) else None)
        return dx, dw if k_w < 0 else None, db if k_b < 0 else None, None


def residual_block(x, w_exp, w1, g1, b1, w2, g2, b2, st1, st2, post=None, cache=None, x_up=False, nseg=1,
                   seg_rev=False, replay_update=True):
    return SF._apply(ResBlockFn16, x, w_exp, w1, g1, b1, w2, g2, b2, st1, st2, post, cache, x_up, nseg, seg_rev,
                     replay_update)


def stem(x, w, g, b, st, nseg=1, seg_rev=False):
    return SF._apply(StemFn16, x, w, g, b, st, nseg, seg_rev)


def conv_bias(x, w, bias, cache=None):
    return SF._apply(PredictFn16, x, w, bias, cache)
