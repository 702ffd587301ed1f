"""Differentiable building blocks of the Soft-IntroVAE networks, composed from the HIP kernels in `ops`.

Granularity is the reference's module granularity (one autograd node per ResidualBlock / stem / predict /
fc) so that the hand-written backward controls exactly which tensors are saved and which fusions run:

  forward of a residual block (train_soft_intro_vae.py:65-75)
      idt = conv1x1(x)                                   (only if inc != outc)
      a   = conv3x3(x)        + BatchNorm partial sums in the conv epilogue
      c   = conv3x3(h)        with h = LeakyReLU(BN1(a)) applied while the conv stages its input tile —
                              h is never written to HBM; + partial sums for BN2 in the epilogue
      out = LeakyReLU(BN2(c) + idt)
  saved for backward: x, a, c, out and 4 per-channel stat vectors (the reference graph saves x, a, h, c, out).
  backward: the LeakyReLU sign of BN1 is recomputed from `a`; conv2's weight gradient re-applies the
  BN1+LeakyReLU prologue on load; the two branches of dx are summed by the dgrad kernel's accumulate epilogue.

Frozen networks (requires_grad=False) still run BatchNorm in training mode and update the running
statistics, as in the reference; weight gradients are skipped per `needs_input_grad` — this is what makes
the iteration cost 13 F_E + 19 F_D.
"""
import os

import torch

from . import ops

SLOPE = ops.LRELU_SLOPE
# SIVAE_MATERIALIZE_H=1: store h = LeakyReLU(BN1(conv1)) once and keep it for backward instead of recomputing it in
# the operand load of conv2 and of conv2's weight gradient.  Measured a wash at 256x256 bs128 (conv kernels -25 ms,
# the extra BatchNorm-apply passes +25 ms per iteration), so the default keeps the smaller activation footprint.
MATERIALIZE_H = os.environ.get("SIVAE_MATERIALIZE_H", "0") == "1"
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# ---------------------------------------------------------------------------------------------------
# packed-weight cache: one forward pack and one dgrad pack per parameter, rebuilt when the parameter
# changes (torch's version counter for in-place updates, plus a generation counter bumped by the fused
# optimizer which writes through raw pointers).
# ---------------------------------------------------------------------------------------------------
_epoch = [0]  # bumped by clear_pack_cache(): every cached pack / replay cache built before it is stale


def cache_epoch():
    return _epoch[0]


def bump_generation(params):
    for p in params:
        p._sivae_gen = getattr(p, "_sivae_gen", 0) + 1


def _wtag(w):
    return (w._version, getattr(w, "_sivae_gen", 0), w.data_ptr(), _epoch[0])


def _cached_pack(w, slot, build):
    """packs live ON the parameter object (they die with it; no id() aliasing, nothing accumulates across models)"""
    store = w.__dict__.setdefault("_sivae_pack", {})
    tag = _wtag(w)
    hit = store.get(slot)
    if hit is not None and hit[0] == tag:
        return hit[1]
    wp = build()
    store[slot] = (tag, wp)
    return wp


def packed(w, mode):
    return _cached_pack(w, mode, lambda: ops.PackedW(w.detach(), mode))


# SIVAE_PACK_BATCH=0: every cached operand form is rebuilt by its own launch on first use after an optimizer step
# (35 launches of 5-11 us per network) instead of one launch per operand form (5 per network) right after the step.
PACK_BATCH = os.environ.get("SIVAE_PACK_BATCH", "1") != "0"


def repack(params, owner):
    """After `params` were updated in place through raw pointers (FlatAdam.step, generation already bumped): rebuild every
    operand form cached on them IN PLACE with one launch per form (sivae_pack_batch) and re-validate the caches.  The job
    tables live on `owner` and are rebuilt only when the set of cached forms changes (the first iterations)."""
    import ctypes
    if not PACK_BATCH:
        return
    L = ops._lib.load()
    entries, jobs = [], [[] for _ in range(6)]
    for p in params:
        store = p.__dict__.get("_sivae_pack")
        if not store:
            continue
        for slot, (tag, obj) in list(store.items()):
            if isinstance(obj, ops.PackedW):
                # a pack captured BEFORE the parameter's storage moved (FlatAdam re-homes p.data into its slab; .to();
                # a second optimizer on a used model) holds a detached alias of the OLD storage: rebuilding it from there
                # and re-tagging it valid would freeze the weights of that slot — drop it, it is rebuilt on demand
                if obj.w.data_ptr() != p.data_ptr() or obj.w.shape != p.shape or tag[2] != p.data_ptr():
                    del store[slot]
                    continue
                forms = obj.batch_forms()
                if forms:
                    entries.append((p, store, slot, obj))
                    for f, buf in forms:
                        jobs[f].append((obj, buf))
                else:
                    del store[slot]
            else:
                del store[slot]  # (the small-channel 5x5 packs: rebuilt on demand)
    if not entries:
        return
    key = tuple((f, obj.w.data_ptr(), buf.data_ptr()) for f in range(6) for obj, buf in jobs[f])
    plan = owner.__dict__.get("_sivae_pack_plan")
    if plan is None or plan["key"] != key:
        if torch.cuda.is_current_stream_capturing():
            # (building the job tables uploads them — not capturable; the forms stay invalid and are rebuilt one by one on
            # their next use, inside the capture, as before round 4)
            return
        jb = L.sivae_pack_job_bytes()
        dev = entries[0][0].device
        launches = []
        for f in range(6):
            if not jobs[f]:
                continue
            if len(jobs[f]) > 32767:
                return  # (the block -> job map is 16 bits wide)
            host = ctypes.create_string_buffer(jb * len(jobs[f]))
            block_job, nblocks = [], 0
            for i, (obj, buf) in enumerate(jobs[f]):
                w = obj.w
                Co, Ci = w.shape[0], w.shape[1]
                ks = w.shape[2] if w.dim() == 4 else 1
                nb = L.sivae_pack_job_fill(host, i, f, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(buf.data_ptr()),
                                           Co, Ci, ks, obj.mode, nblocks)
                if nb <= 0:
                    raise ops._lib.SivaeError("sivae_pack_job_fill", nb)
                block_job.extend([i] * nb)
                nblocks += nb
            jt = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
            bj = torch.tensor(block_job, dtype=torch.int16).to(dev)  # (indices < 32768: uint16 and int16 agree)
            launches.append((f, jt, bj, nblocks))
        plan = {"key": key, "launches": launches, "keep": [b for fl in jobs for _, b in fl]}
        # (job tables of earlier plans stay alive: a HIP graph captured while they were current still launches with them)
        # — the last few only: a workload whose set of cached forms oscillates would otherwise accumulate device memory
        prev = owner.__dict__.get("_sivae_pack_plan")
        if prev is not None:
            old = owner.__dict__.setdefault("_sivae_pack_plan_old", [])
            old.append(prev)
            del old[:-4]
        owner.__dict__["_sivae_pack_plan"] = plan
    for f, jt, bj, nblocks in plan["launches"]:
        ops._lib.call("sivae_pack_batch", f, ops._p(jt), ops._p(bj), nblocks, ops._s(jt))
    for p, store, slot, obj in entries:
        obj.refreshed()
        store[slot] = (_wtag(p), obj)


def packed5(w, mode):
    """pack for the small-channel 5x5 kernels (same invalidation rules as `packed`)"""
    return _cached_pack(w, 10 + mode, lambda: ops.pack5_smallco(w.detach(), mode))


def _is_edge5(w):
    return w.dim() == 4 and w.shape[2] == 5 and w.shape[3] == 5 and min(w.shape[0], w.shape[1]) <= 3


def clear_pack_cache():
    _epoch[0] += 1


def cache_tag(weights):
    """validity tag of a replayable forward pass: (version, optimizer generation, storage) of every weight it used"""
    return tuple(None if w is None else _wtag(w) for w in weights)


class BNState:
    """The non-differentiable part of a BatchNorm2d: buffers + mode. Passed by reference into the Functions."""

    __slots__ = ("running_mean", "running_var", "num_batches_tracked", "training", "eps", "momentum")

    def __init__(self, bn_module):
        self.running_mean = bn_module.running_mean
        self.running_var = bn_module.running_var
        self.num_batches_tracked = bn_module.num_batches_tracked
        self.training = bn_module.training
        self.eps = bn_module.eps
        self.momentum = bn_module.momentum


def _stats(partials, B, C, HW, st, nseg=1, seg_rev=False):
    """batch statistics from conv-epilogue partials (training) or the running buffers (eval).
    nseg > 1 (segmented batch: nseg passes of B / nseg images laid end to end): one set of statistics per pass,
    returned as nseg * C entries; the running buffers get one update per pass (seg_rev: last pass first)."""
    if st.training:
        return ops.bn_stats_from_conv(partials, B, C, HW, st.running_mean, st.running_var, st.num_batches_tracked,
                                      st.eps, st.momentum, nseg=nseg, seg_rev=seg_rev)
    mean, invstd = st.running_mean, torch.rsqrt(st.running_var + st.eps)
    if nseg > 1:
        mean, invstd = mean.repeat(nseg), invstd.repeat(nseg)
    return mean, invstd


def _post_fwd(out, post):
    if post == "pool":
        return ops.avgpool2_fwd(out)
    if post == "up":
        return ops.upsample2_fwd(out)
    return out


def _post_bwd(dy, post, shape):
    if post == "pool":
        return ops.avgpool2_bwd(dy, shape[2], shape[3])
    if post == "up":
        return ops.upsample2_bwd(dy)
    return dy


# ---------------------------------------------------------------------------------------------------
def _replay_bn(st, mean, invstd, count, nseg=1, seg_rev=False):
    if ops.SYNC_BN is not None:
        count = count * ops.SYNC_BN(None)  # (None -> just the world size)
    if st.training:
        ops.bn_update_running(mean, invstd, count, st.running_mean, st.running_var, st.num_batches_tracked, st.eps,
                              st.momentum, nseg=nseg, seg_rev=seg_rev)


# ---- parameter gradients written straight into the optimizer's slabs ------------------------------------------------
# A parameter used by several passes of one backward (the encoder runs three times inside lossE, the decoder four times
# inside lossD: train_soft_intro_vae.py:566-571, :601-619) gets one gradient per use, and autograd folds them with one
# elementwise add per tensor and use: 343 tiny launches per iteration.  With optim.FlatAdam.enable_slabs() every USE
# owns a slab (a second / third flat buffer laid out like flat_grad): the forward reserves the slab index, the backward
# hands the slab view to the kernel that produces the gradient as its destination and returns None to autograd, and one
# launch per network (`FlatAdam.fold_slabs`) sums the slabs into flat_grad before the all-reduce and the Adam step.
# SIVAE_DIRECT_GRADS=0 (or a parameter without slabs: plain autograd use of the modules) keeps autograd's accumulation.
DIRECT_GRADS = os.environ.get("SIVAE_DIRECT_GRADS", "1") != "0"


_GRAD_MODE = [True]  # torch.is_grad_enabled() at the call site (inside Function.forward autograd has switched it off)


def _apply(fn, *args):
    """fn.apply(*args), remembering whether the CALLER records a graph: under torch.no_grad() (test_iter dumps,
    model.sample, the FID feed) needs_input_grad is still True for parameters, and a slab reserved there would never be
    written — the next backward would run out of slabs and lose the overlapped gradient sync"""
    prev = _GRAD_MODE[0]
    _GRAD_MODE[0] = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _GRAD_MODE[0] = prev


def _claim(ctx, indexed_params):
    """forward: reserve a slab index for each (input position, parameter) that will receive a gradient from this node"""
    use = []
    for i, p in indexed_params:
        k = -1
        if DIRECT_GRADS and _GRAD_MODE[0] and p is not None and ctx.needs_input_grad[i]:
            slabs = p.__dict__.get("_sivae_slabs")
            if slabs is not None:
                u = p.__dict__.get("_sivae_use", 0)
                if u < len(slabs):
                    k = u
                    p.__dict__["_sivae_use"] = u + 1
        use.append(k)
    ctx.use = tuple(use)


def _dst(p, k):
    """the slab view reserved for this use of parameter p (None: return the gradient to autograd)"""
    return p.__dict__["_sivae_slabs"][k] if k >= 0 else None


def _pg_dst(g, kg, b, kb):
    """(dgamma, dbeta) destinations of a BatchNorm backward, or None (both or neither go direct)"""
    if kg < 0 or kb < 0:
        return None
    return (_dst(g, kg), _dst(b, kb))


def _done(*params):
    """after the kernel that wrote a use's gradient into its slab has been enqueued: tell the gradient synchroniser
    (dp.GradSync watches the tail of the flat buffer to start its all-reduce during the backward)"""
    for p in params:
        cb = None if p is None else p.__dict__.get("_sivae_on_grad")
        if cb is not None:
            cb(p)


class ResBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_exp, w1, g1, b1, w2, g2, b2, st1, st2, post, cache=None, x_up=False, nseg=1, seg_rev=False,
                replay_update=True):
        """cache: None, or a dict owned by the caller.  An empty dict is FILLED with this pass's activations;
        a filled one is REPLAYED: no kernels run except the BatchNorm running-stat updates, the outputs and the
        tensors saved for backward are the cached ones.  A replay is only valid while x and all weights are unchanged:
        the weights are checked here (`cache_tag`: torch version counter, optimizer generation, storage) and a stale
        entry is recomputed; the input is checked once per pass by Decoder.forward.

        x_up: x is stored at HALF resolution and stands for Upsample(2,'nearest')(x) (train_soft_intro_vae.py:155):
        every consumer (conv1, conv_expand or the identity add, both weight gradients) reads it through upsample
        addressing, so the 4x tensor is never written.  post == "up_deferred": the Upsample after this block is
        left to the next block's x_up (the output is returned at this block's resolution).

        nseg > 1: SEGMENTED batch — x holds nseg independent passes of the network (B / nseg images each) laid end to
        end; the convolutions run once over the whole batch, every BatchNorm keeps one set of batch statistics per pass
        (what the reference's separate calls compute, train_soft_intro_vae.py:567-568, :601-608), the running buffers
        are updated once per pass in pass order (seg_rev: last segment first).
        replay_update=False: a replay leaves the running buffers alone (the pass that filled the cache already counted:
        `cache_segment` views)."""
        x = x.contiguous()
        _claim(ctx, ((1, w_exp), (2, w1), (3, g1), (4, b1), (5, w2), (6, g2), (7, b2)))
        B, Ci, H, W = x.shape
        if x_up:
            H, W = 2 * H, 2 * W
        ctx.x_up = x_up
        Cm, Co = w1.shape[0], w2.shape[0]
        tag = cache_tag((w_exp, w1, g1, b1, w2, g2, b2))
        if cache is not None and cache.get("y") is not None and cache.get("tag") == tag:
            a, h, c, out, mean1, invstd1, mean2, invstd2, y = (cache[k] for k in (
                "a", "h", "c", "out", "mean1", "invstd1", "mean2", "invstd2", "y"))
            if replay_update:
                _replay_bn(st1, mean1, invstd1, (B // nseg) * H * W, nseg, seg_rev)
                _replay_bn(st2, mean2, invstd2, (B // nseg) * H * W, nseg, seg_rev)
            ctx.nseg = nseg
            ctx.post = post
            ctx.has_exp = w_exp is not None
            ctx.training = st1.training and st2.training
            ctx.save_for_backward(x, a, h, c, out, mean1, invstd1, mean2, invstd2, w_exp, w1, g1, b1, w2, g2, b2)
            return y.view_as(y)
        if not replay_update:
            # a `cache_segment` view that cannot be replayed (sub-cache not filled, weights changed since the fill) would be
            # recomputed HERE with a third running-statistics update the reference never makes — fail instead
            raise RuntimeError("sivae_hip: replay_update=False needs a filled, current replay cache for this block")
        idt = x
        if w_exp is not None:
            # a 1x1 conv commutes with nearest upsampling pixel for pixel (bit-exactly): with x_up it runs on the
            # half-resolution tensor (1/4 of the FLOPs and output bytes) and the residual add reads it through
            # upsample addressing like an identity skip
            idt = ops.conv2d_fwd(x, packed(w_exp, 0), Co, 1)
        if st1.training:
            a, p1 = ops.conv2d_fwd(x, packed(w1, 0), Cm, 3, want_stats=True, upsample=x_up, nseg=nseg)
        else:
            a, p1 = ops.conv2d_fwd(x, packed(w1, 0), Cm, 3, upsample=x_up), None
        mean1, invstd1 = _stats(p1, B, Cm, H * W, st1, nseg, seg_rev)
        if MATERIALIZE_H or (nseg > 1 and not ops.seg_prologue_supported(H, W)):
            # h = LeakyReLU(BN1(a)) written once (2 HBM passes over a Cm-channel tensor) and kept for backward
            # (segmented batches on the 4x4 / 8x8 maps: the kernels that take those maps have no per-segment prologue)
            h = ops.bn_apply_act(a, None, mean1, invstd1, g1.detach(), b1.detach(), SLOPE, nseg=nseg)
            pro1 = None
        else:
            h = a
            pro1 = (mean1, invstd1, g1.detach(), b1.detach(), SLOPE)
        if st2.training:
            c, p2 = ops.conv2d_fwd(h, packed(w2, 0), Co, 3, pro=pro1, want_stats=True, nseg=nseg)
        else:
            c, p2 = ops.conv2d_fwd(h, packed(w2, 0), Co, 3, pro=pro1, nseg=nseg), None
        mean2, invstd2 = _stats(p2, B, Co, H * W, st2, nseg, seg_rev)
        fused = None
        pool_fusable = post == "pool" and not (x_up and w_exp is None)
        if st1.training and st2.training and ops.bn_signmask_supported(c):
            # the backward takes the LeakyReLU sign from a 1-bit mask written here, not from the output: `out` below
            # is that mask (uint8), and a pooled block never writes its full-resolution output
            full, y, out = ops.bn_apply_act_signmask(c, idt, mean2, invstd2, g2.detach(), b2.detach(), SLOPE,
                                                     res_up=x_up, pool=pool_fusable, want_full=not pool_fusable,
                                                     nseg=nseg)
            if not pool_fusable:
                y = _post_fwd(full, post)
            del full
        else:
            if pool_fusable:
                fused = ops.bn_apply_act_pool(c, idt, mean2, invstd2, g2.detach(), b2.detach(), SLOPE, nseg=nseg)
            if fused is not None:
                out, y = fused  # BatchNorm + residual + LeakyReLU and the AvgPool2d that follows, one pass
            else:
                out = ops.bn_apply_act(c, idt, mean2, invstd2, g2.detach(), b2.detach(), SLOPE, res_up=x_up, nseg=nseg)
                y = _post_fwd(out, post)
        if cache is not None:
            cache.update(a=a, h=h, c=c, out=out, mean1=mean1, invstd1=invstd1, mean2=mean2, invstd2=invstd2, y=y,
                         tag=tag)
        ctx.nseg = nseg
        ctx.post = post
        ctx.has_exp = w_exp is not None
        ctx.training = st1.training and st2.training
        ctx.save_for_backward(x, a, h, c, out, mean1, invstd1, mean2, invstd2, w_exp, w1, g1, b1, w2, g2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("sivae_hip: backward through eval-mode BatchNorm is not supported")
        x, a, h, c, out, mean1, invstd1, mean2, invstd2, w_exp, w1, g1, b1, w2, g2, b2 = ctx.saved_tensors
        k_we, k_w1, k_g1, k_b1, k_w2, k_g2, k_b2 = ctx.use
        nseg = ctx.nseg
        h_saved = h.data_ptr() != a.data_ptr()
        need = ctx.needs_input_grad
        need_x, need_we, need_w1, need_bn1, need_w2, need_bn2 = need[0], need[1], need[2], need[3] or need[4], \
            need[5], need[6] or need[7]
        Cm = w1.shape[0]
        # BN2 + residual + LeakyReLU; the AvgPool2d that follows an encoder block is undone while reading dy
        x_up = ctx.x_up
        pg2 = _pg_dst(g2, k_g2, b2, k_b2) if need_bn2 else None
        pg1 = _pg_dst(g1, k_g1, b1, k_b1) if need_bn1 else None
        dzh = None  # 2x2 block sums of dz: all a block behind an Upsample ever needs of it
        if out.dtype == torch.uint8:  # `out` is the LeakyReLU sign mask (1 bit per element)
            if ctx.post == "pool" and not (x_up and not ctx.has_exp):
                dc, dz, dg2, db2 = ops.bn_bwd_signmask(dy.contiguous(), out, c, mean2, invstd2, g2, SLOPE,
                                                       dy_pooled=True, want_param_grads=need_bn2, pg_out=pg2,
                                                       nseg=nseg)
            else:
                d_out = _post_bwd(dy.contiguous(), ctx.post, c.shape)
                want_sum = x_up and ctx.post != "pool"
                dc, dz, dg2, db2 = ops.bn_bwd_signmask(d_out, out, c, mean2, invstd2, g2, SLOPE, dz_sum=want_sum,
                                                       want_param_grads=need_bn2, pg_out=pg2, nseg=nseg)
                if want_sum:
                    dzh, dz = dz, None
                del d_out
        elif x_up and ctx.post != "pool" and ops.bn_bwd_dzsum_supported(c):
            d_out = _post_bwd(dy.contiguous(), ctx.post, out.shape)
            dc, dzh, dg2, db2 = ops.bn_bwd_dzsum(d_out, out, c, mean2, invstd2, g2, SLOPE, want_param_grads=need_bn2,
                                                 pg_out=pg2, nseg=nseg)
            dz = None
            del d_out
        elif ctx.post == "pool":
            dc, dz, dg2, db2 = ops.bn_bwd(dy.contiguous(), out, c, mean2, invstd2, g2, SLOPE, want_dz=True,
                                          want_param_grads=need_bn2, act_mode=1, dy_pooled=True, pg_out=pg2, nseg=nseg)
        else:
            d_out = _post_bwd(dy.contiguous(), ctx.post, out.shape)
            dc, dz, dg2, db2 = ops.bn_bwd(d_out, out, c, mean2, invstd2, g2, SLOPE, want_dz=True,
                                          want_param_grads=need_bn2, act_mode=1, pg_out=pg2, nseg=nseg)
            del d_out
        pro1 = None if h_saved else (mean1, invstd1, g1, b1, SLOPE)
        if pg2 is not None:
            _done(g2, b2)
        dw2 = ops.conv2d_wgrad(h, dc, 3, pro=pro1, out=_dst(w2, k_w2), nseg=nseg) if need_w2 else None
        if need_w2 and k_w2 >= 0:
            _done(w2)
        fuse_bn1 = (not h_saved) and nseg == 1 and ops.conv2d_dgrad_bnbwd_supported(dc.shape[2], dc.shape[3])
        if fuse_bn1:
            # conv2's data gradient also reduces BatchNorm-1's backward sums in its epilogue (one pass fewer over dh, a)
            dh, part1 = ops.conv2d_dgrad_bnbwd(dc, packed(w2, 1), Cm, a, mean1, invstd1, g1, b1, SLOPE)
        else:
            dh = ops.conv2d_fwd(dc, packed(w2, 1), Cm, 3)
        del dc
        # BN1 + LeakyReLU (sign from the saved h, or recomputed from a when h was never stored)
        if fuse_bn1:
            da, dg1, db1 = ops.bn_bwd_from_partials(dh, a, mean1, invstd1, g1, b1, part1, SLOPE,
                                                    want_param_grads=need_bn1, pg_out=pg1)
        elif h_saved:
            da, _, dg1, db1 = ops.bn_bwd(dh, h, a, mean1, invstd1, g1, SLOPE, want_dz=False,
                                         want_param_grads=need_bn1, act_mode=1, pg_out=pg1, nseg=nseg)
        else:
            da, _, dg1, db1 = ops.bn_bwd(dh, None, a, mean1, invstd1, g1, SLOPE, want_dz=False,
                                         want_param_grads=need_bn1, beta=b1, act_mode=2, pg_out=pg1, nseg=nseg)
        del dh
        if pg1 is not None:
            _done(g1, b1)
        dw1 = ops.conv2d_wgrad(x, da, 3, upsample=x_up, out=_dst(w1, k_w1)) if need_w1 else None
        if need_w1 and k_w1 >= 0:
            _done(w1)
        dwe = None
        dx = None
        # conv1's data gradient straight to the low-resolution x (phase-folded F(2x2,2x2) kernel)
        up_dg = x_up and ops.conv2d_up_dgrad_supported(x.shape[2], x.shape[3])
        if ctx.has_exp and x_up:
            # the expand conv ran at half resolution: its gradients do too (dz summed over each 2x2 block first)
            if dzh is None:
                dzh = ops.upsample2_bwd(dz) if (need_we or need_x) else None
            if need_we:
                dwe = ops.conv2d_wgrad(x, dzh, 1, out=_dst(w_exp, k_we))
            if need_x:
                if up_dg:
                    dx = ops.conv2d_up_dgrad(da, packed(w1, 0), x.shape[1], wp1=packed(w1, 1))
                else:
                    dx = ops.upsample2_bwd(ops.conv2d_fwd(da, packed(w1, 1), x.shape[1], 3))
                ops.conv2d_fwd(dzh, packed(w_exp, 1), x.shape[1], 1, out=dx, accumulate=True)
        elif up_dg and need_x:
            dx = dzh if dzh is not None else ops.upsample2_bwd(dz)  # identity branch, already at low resolution
            ops.conv2d_up_dgrad(da, packed(w1, 0), x.shape[1], out=dx, accumulate=True, wp1=packed(w1, 1))
        elif x_up and dzh is not None and dz is None:
            # identity skip behind an Upsample on a map the phase kernel does not take: reduce conv1's full-resolution
            # data gradient, then add the (already reduced) skip gradient
            if need_x:
                dx = ops.upsample2_bwd(ops.conv2d_fwd(da, packed(w1, 1), x.shape[1], 3))
                ops.add_(dx, dzh)
        else:
            if ctx.has_exp:
                if need_we:
                    dwe = ops.conv2d_wgrad(x, dz, 1, out=_dst(w_exp, k_we))
                if need_x:
                    dx = ops.conv2d_fwd(da, packed(w1, 1), x.shape[1], 3)
                    ops.conv2d_fwd(dz, packed(w_exp, 1), x.shape[1], 1, out=dx, accumulate=True)
            elif need_x:
                dx = dz  # identity branch gradient; add the conv1 branch on top
                ops.conv2d_fwd(da, packed(w1, 1), x.shape[1], 3, out=dx, accumulate=True)
            if x_up and dx is not None:
                dx = ops.upsample2_bwd(dx)  # adjoint of the deferred Upsample: sum each 2x2 block
        if need_we and ctx.has_exp and k_we >= 0:
            _done(w_exp)
        # (gradients that went into a slab are not handed to autograd)
        return (dx, dwe if k_we < 0 else None, dw1 if k_w1 < 0 else None,
                dg1 if (need[3] and pg1 is None) else None, db1 if (need[4] and pg1 is None) else None,
                dw2 if k_w2 < 0 else None,
                dg2 if (need[6] and pg2 is None) else None, db2 if (need[7] and pg2 is None) else None,
                None, None, None, None, None, None, None, None)


class StemFn(torch.autograd.Function):
    """conv5x5 -> BatchNorm -> LeakyReLU -> AvgPool2d(2)   (train_soft_intro_vae.py:88-93)"""

    @staticmethod
    def forward(ctx, x, w, g, b, st, nseg=1, seg_rev=False):
        x = x.contiguous()
        _claim(ctx, ((1, w), (2, g), (3, b)))
        B, Ci, H, W = x.shape
        Co = w.shape[0]
        if st.training:
            a, p = ops.conv2d_fwd(x, packed(w, 0), Co, 5, want_stats=True)
            if nseg > 1 and p.shape[0] % nseg:
                raise ValueError("sivae_hip: the stem's statistics rows do not split into %d segments" % nseg)
        else:
            a, p = ops.conv2d_fwd(x, packed(w, 0), Co, 5), None
        mean, invstd = _stats(p, B, Co, H * W, st, nseg, seg_rev)
        fused = ops.bn_apply_act_pool(a, None, mean, invstd, g.detach(), b.detach(), SLOPE, want_full=False, nseg=nseg)
        if fused is not None:
            out = fused[1]  # (the full-resolution activation is never written: backward recomputes it from `a`)
        else:
            out = ops.avgpool2_fwd(ops.bn_apply_act(a, None, mean, invstd, g.detach(), b.detach(), SLOPE, nseg=nseg))
        ctx.nseg = nseg
        ctx.training = st.training
        ctx.save_for_backward(x, a, mean, invstd, w, g, b)
        return out

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("sivae_hip: backward through eval-mode BatchNorm is not supported")
        x, a, mean, invstd, w, g, b = ctx.saved_tensors
        need = ctx.needs_input_grad
        k_w, k_g, k_b = ctx.use
        pg = _pg_dst(g, k_g, b, k_b) if (need[2] or need[3]) else None
        da, _, dg, db = ops.bn_bwd(dy.contiguous(), None, a, mean, invstd, g, SLOPE, want_dz=False,
                                   want_param_grads=need[2] or need[3], beta=b, act_mode=2, dy_pooled=True,
                                   pg_out=pg, nseg=ctx.nseg)
        edge = _is_edge5(w) and w.shape[1] <= 3
        dw = None
        if need[1]:
            dst = _dst(w, k_w)
            dw = ops.conv5_edge_wgrad(x, da, out=dst) if edge else ops.conv2d_wgrad(x, da, 5, out=dst)
            if dst is not None:
                _done(w)
        if pg is not None:
            _done(g, b)
        dx = None
        if need[0]:
            dx = (ops.conv5_smallco_fwd(da, packed5(w, 1), x.shape[1]) if edge
                  else ops.conv2d_fwd(da, packed(w, 1), x.shape[1], 5))
        return (dx, dw if k_w < 0 else None, dg if (need[2] and pg is None) else None,
                db if (need[3] and pg is None) else None, None, None, None)


class ConvBiasFn(torch.autograd.Function):
    """plain conv + bias (Decoder.predict, train_soft_intro_vae.py:159)"""

    @staticmethod
    def forward(ctx, x, w, bias, cache=None):
        x = x.contiguous()
        _claim(ctx, ((1, w), (2, bias)))
        ks = w.shape[2]
        ctx.save_for_backward(x, w, bias)
        ctx.has_bias = bias is not None
        tag = cache_tag((w, bias))
        if cache is not None and cache.get("y") is not None and cache.get("tag") == tag:
            return cache["y"].view_as(cache["y"])
        b_ = None if bias is None else bias.detach()
        if _is_edge5(w) and w.shape[0] <= 3:
            y = ops.conv5_smallco_fwd(x, packed5(w, 0), w.shape[0], bias=b_)
        else:
            y = ops.conv2d_fwd(x, packed(w, 0), w.shape[0], ks, bias=b_)
        if cache is not None:
            cache["y"], cache["tag"] = y, tag
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        k_w, k_b = ctx.use
        dy = dy.contiguous()
        ks = w.shape[2]
        dw = None
        if need[1]:
            dst = _dst(w, k_w)
            dw = ops.conv5_edge_wgrad(x, dy, out=dst) if _is_edge5(w) else ops.conv2d_wgrad(x, dy, ks, out=dst)
        db = ops.channel_sum(dy, out=_dst(bias, k_b)) if (ctx.has_bias and need[2]) else None
        _done(w if (need[1] and k_w >= 0) else None, bias if (db is not None and k_b >= 0) else None)
        dx = ops.conv2d_fwd(dy, packed(w, 1), x.shape[1], ks) if need[0] else None
        return dx, dw if k_w < 0 else None, db if k_b < 0 else None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b (optionally followed by ReLU): the small-batch GEMM kernels of linear.hip (split-contraction, ~1000
    waves per call); shapes they do not take run on the ks=1 path of the conv kernels (H = W = 1)."""

    @staticmethod
    def forward(ctx, x, w, bias, relu):
        x = x.contiguous()
        if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
            # nn.Linear's error (e.g. a conditional model called without its condition, reference :118-120): the raw
            # kernels take pointers and would read past the end of x
            raise RuntimeError("mat1 and mat2 shapes cannot be multiplied (%s and %dx%d)" % (
                "x".join(str(d) for d in x.shape), w.shape[1] if w.dim() == 2 else -1, w.shape[0]))
        _claim(ctx, ((1, w), (2, bias)))
        B, K = x.shape
        N = w.shape[0]
        ctx.fast = ops.linear_supported(B, K, N)
        b_ = None if bias is None else bias.detach()
        if ctx.fast:
            y = ops.linear_fwd(x, w.detach(), b_, relu)
        else:
            y = ops.conv2d_fwd(x.view(B, K, 1, 1), packed(w, 0), N, 1, bias=b_).view(B, N)
            if relu:
                ops.relu_fwd(y, inplace=True)
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, y if relu else None, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, bias = ctx.saved_tensors
        need = ctx.needs_input_grad
        k_w, k_b = ctx.use
        dy = dy.contiguous()
        if ctx.relu:
            dy = ops.relu_bwd(dy, y)
        B, K = x.shape
        N = w.shape[0]
        dy4 = dy.view(B, N, 1, 1)
        db = ops.channel_sum(dy4, out=_dst(bias, k_b)) if (ctx.has_bias and need[2]) else None
        dst = _dst(w, k_w) if need[1] else None
        if ctx.fast:
            dw = ops.linear_wgrad(dy, x, out=dst) if need[1] else None
            dx = ops.linear_dgrad(dy, w.detach()) if need[0] else None
        else:
            dw = None
            if need[1]:
                dw = ops.conv2d_wgrad(x.view(B, K, 1, 1), dy4, 1, out=None if dst is None else dst.view(N, K, 1, 1))
                dw = dw.view(N, K)
            dx = ops.conv2d_fwd(dy4, packed(w, 1), K, 1).view(B, K) if need[0] else None
        _done(w if (need[1] and k_w >= 0) else None, bias if (db is not None and k_b >= 0) else None)
        return dx, dw if k_w < 0 else None, db if k_b < 0 else None, None


# ---------------------------------------------------------------------------------------------------
# sampler / losses
# ---------------------------------------------------------------------------------------------------
class ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        z = ops.reparam_fwd(mu, logvar, eps)
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dmu, dlv = ops.reparam_bwd(dz.contiguous(), logvar, eps)
        return dmu, dlv, None


class KLFn(torch.autograd.Function):
    """calc_kl: per-sample KL, optionally reduced ('sum' / 'mean') — train_soft_intro_vae.py:231-251.
    mu_o / logvar_o: python floats, or tensors broadcastable to [B, Z] (read on the device, no host sync)."""

    @staticmethod
    def forward(ctx, logvar, mu, mu_o, logvar_o, reduce):
        tensor_prior = isinstance(mu_o, torch.Tensor) or isinstance(logvar_o, torch.Tensor)
        if tensor_prior:
            mu_o = mu_o if isinstance(mu_o, torch.Tensor) else torch.tensor(float(mu_o))
            logvar_o = logvar_o if isinstance(logvar_o, torch.Tensor) else torch.tensor(float(logvar_o))
            per = ops.kl_fwd_t(logvar, mu, mu_o, logvar_o)
        else:
            per = ops.kl_fwd(logvar, mu, mu_o, logvar_o)
        ctx.save_for_backward(logvar, mu)
        ctx.cfg = (mu_o, logvar_o, reduce, mu.shape[0], tensor_prior)
        if reduce == "sum":
            return ops.vec_sum(per, 1.0)
        if reduce == "mean":
            return ops.vec_sum(per, 1.0 / mu.shape[0])
        return per

    @staticmethod
    def backward(ctx, g):
        logvar, mu = ctx.saved_tensors
        mu_o, logvar_o, reduce, B, tensor_prior = ctx.cfg
        g = g.contiguous()
        bwd = ops.kl_bwd_t if tensor_prior else ops.kl_bwd
        if reduce == "sum":
            dlv, dmu = bwd(g, False, 1.0, logvar, mu, mu_o, logvar_o)
        elif reduce == "mean":
            dlv, dmu = bwd(g, False, 1.0 / B, logvar, mu, mu_o, logvar_o)
        else:
            dlv, dmu = bwd(g, True, 1.0, logvar, mu, mu_o, logvar_o)
        return dlv, dmu, None, None, None


class ReconFn(torch.autograd.Function):
    """calc_reconstruction_loss — train_soft_intro_vae.py:268-294.
    mode: 'rows' -> [B] per-sample sums, 'total' -> scalar (scale applied), 'elem' -> [B, D]"""

    @staticmethod
    def forward(ctx, x, recon, loss_type, mode, scale):
        B = x.shape[0]
        x2 = x.contiguous().view(B, -1)
        r2 = recon.contiguous().view(B, -1)
        ctx.save_for_backward(x2, r2)
        ctx.cfg = (loss_type, mode, scale, x.shape, recon.shape)
        if mode == "elem":
            return ops.recon_elem_fwd(x2, r2, loss_type)
        rows = ops.recon_rowsum_fwd(x2, r2, loss_type)
        if mode == "rows":
            return rows
        return ops.vec_sum(rows, scale)

    @staticmethod
    def backward(ctx, g):
        x2, r2 = ctx.saved_tensors
        loss_type, mode, scale, xshape, rshape = ctx.cfg
        need = ctx.needs_input_grad
        g = g.contiguous()
        g_mode = {"rows": 0, "total": 1, "elem": 2}[mode]
        d_r, d_x = ops.recon_bwd(x2, r2, loss_type, g, g_mode, scale if mode == "total" else 1.0,
                                 want_drecon=need[1], want_dx=need[0])
        return (d_x.view(xshape) if d_x is not None else None, d_r.view(rshape) if d_r is not None else None,
                None, None, None)


class ExpElboFn(torch.autograd.Function):
    """mean_i exp(-2*scale*(beta_rec*L_i + beta_neg*KL_i)) — train_soft_intro_vae.py:580-581"""

    @staticmethod
    def forward(ctx, L, KL, scale, beta_rec, beta_neg):
        out, e = ops.expelbo_fwd(L.contiguous(), KL.contiguous(), scale, beta_rec, beta_neg)
        ctx.save_for_backward(e)
        ctx.cfg = (scale, beta_rec, beta_neg)
        return out

    @staticmethod
    def backward(ctx, g):
        (e,) = ctx.saved_tensors
        scale, beta_rec, beta_neg = ctx.cfg
        dL, dKL = ops.expelbo_bwd(g.contiguous(), e, scale, beta_rec, beta_neg)
        return dL, dKL, None, None, None


class LinCombFn(torch.autograd.Function):
    """sum_i w_i * t_i over device scalars — the assembly of lossE / lossD (train_soft_intro_vae.py:583-586, :618-620)"""

    @staticmethod
    def forward(ctx, ws, *ts):
        ctx.ws = ws
        return ops.lincomb([t.detach().reshape(()).contiguous() for t in ts], ws)

    @staticmethod
    def backward(ctx, g):
        gs = ops.lincomb_bwd(g.contiguous(), ctx.ws)
        return (None,) + tuple(gs[i] if need else None for i, need in enumerate(ctx.needs_input_grad[1:]))


def lincomb(ts, ws):
    """differentiable sum_i ws[i] * ts[i] (python-float weights, 0-dim float32 device tensors), one launch each way"""
    ts = list(ts)
    if not 0 < len(ts) <= 6 or len(ts) != len(ws):
        raise ValueError("sivae_hip.lincomb: 1..6 terms with one weight each")
    for t in ts:
        if t.dim() != 0 or t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError("sivae_hip.lincomb: terms must be 0-dim float32 device scalars")
    return LinCombFn.apply(tuple(float(w) for w in ws), *ts)


def residual_block(x, w_exp, w1, g1, b1, w2, g2, b2, st1, st2, post=None, cache=None, x_up=False, nseg=1,
                   seg_rev=False, replay_update=True):
    return _apply(ResBlockFn, x, w_exp, w1, g1, b1, w2, g2, b2, st1, st2, post, cache, x_up, nseg, seg_rev,
                  replay_update)


def cache_segment(cache, g, nseg):
    """A view of segment g of a FILLED replay cache of a segmented pass (Decoder.forward(..., cache=, nseg=)): the same
    activations / statistics restricted to that pass's images, usable as the cache of an UNsegmented replay.  The
    engine builds the autograd graph of ONE pass of a pair this way (E-step: `rec` needs a data gradient, `fake` has no
    graph at all — reference :557-561) after the pair ran as one batch without a graph."""
    def cut(t, n_rows=None):
        if t is None:
            return None
        n = t.shape[0] // nseg
        return t[g * n:(g + 1) * n]
    out = {}
    for k, sub in cache.items():
        if not isinstance(sub, dict):
            continue
        if sub.get("y") is None:
            return None  # (not filled)
        v = {"tag": sub.get("tag"), "y": cut(sub["y"])}
        for name in ("a", "h", "c", "out", "mean1", "invstd1", "mean2", "invstd2"):
            if name in sub:
                v[name] = cut(sub[name])
        out[k] = v
    return out


def stem(x, w, g, b, st, nseg=1, seg_rev=False):
    return _apply(StemFn, x, w, g, b, st, nseg, seg_rev)


def conv_bias(x, w, bias, cache=None):
    return _apply(ConvBiasFn, x, w, bias, cache)


def linear(x, w, bias, relu=False):
    return _apply(LinearFn, x, w, bias, relu)


def reparameterize(mu, logvar, eps):
    return ReparamFn.apply(mu, logvar, eps)


def kl(logvar, mu, mu_o=0.0, logvar_o=0.0, reduce="none"):
    for t in (mu_o, logvar_o):
        if isinstance(t, torch.Tensor) and t.requires_grad:
            raise NotImplementedError("sivae_hip: calc_kl does not propagate gradients into tensor priors "
                                      "(mu_o / logvar_o); detach them")
    mu_o = mu_o if isinstance(mu_o, torch.Tensor) else float(mu_o)
    logvar_o = logvar_o if isinstance(logvar_o, torch.Tensor) else float(logvar_o)
    return KLFn.apply(logvar, mu, mu_o, logvar_o, reduce)


def expelbo(L, KL, scale, beta_rec, beta_neg):
    return ExpElboFn.apply(L, KL, float(scale), float(beta_rec), float(beta_neg))
