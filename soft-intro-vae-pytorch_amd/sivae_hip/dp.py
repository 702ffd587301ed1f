"""One-process-per-GPU data parallelism for the Soft-IntroVAE iteration.

The reference's hot path is single-device; its only DP precedent is the style variant (DDP over NCCL:
style_soft_intro_vae/launcher.py:26-29, train_style_soft_intro_vae.py:154-161).  Here the batch shards
per image, every rank keeps replicated weights + Adam state, and the ONLY exchange per backward is one
all-reduce of a flat fp32 gradient buffer (encoder after lossE.backward(), decoder after
lossD.backward()) — `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm.  The 1/world scaling
is folded into the fused Adam kernel (grad_scale), so the reduced buffer is touched exactly once.
BatchNorm statistics stay per-rank ("local BN", what DDP does to this model; broadcast_buffers=False in
the style precedent).  Works on any device the process group supports (gloo on CPU in the tests).
"""
import datetime
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(
        os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # rank 0 alone computes FID / writes checkpoints between iterations while the others already wait in the next
        # collective: the default watchdog limit (10 min) is shorter than an Inception pass over 50 000 images
        timeout = datetime.timedelta(seconds=int(os.environ.get("SIVAE_DP_TIMEOUT_S", "7200")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return world, rank, local


def _coll_device(device=None):
    """device a small helper tensor must live on for a collective of the current backend (RCCL: the GPU)"""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_int(value, src=0, device=None):
    """rank `src`'s python int on every rank (e.g. the unseeded run's stream seed: DistributedSampler only partitions
    the dataset when every rank permutes with the SAME seed)"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_coll_device(device))
    dist.broadcast(t, src=src)
    return int(t.item())


def any_rank(flag, device=None):
    """True on EVERY rank if `flag` is true on any rank: abort decisions (NaN loss, negative KL difference) must be
    collective — a rank that raises alone leaves the others hanging in the next all-reduce"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0.0)


def mean_over_ranks(values, device=None):
    """element-wise mean over the ranks of a short list of python floats (epoch statistics: every rank logs / decides
    on the GLOBAL figure, not on its own shard's)"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) / dist.get_world_size() for v in t.tolist()]


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def allreduce_sum_(flat):
    """in-place SUM all-reduce of a flat buffer (one collective per network per iteration)"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def broadcast_(tensors, src=0):
    """rank `src`'s values into every rank (initial weights / Adam state / buffers)"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shard_batch(global_batch, world, rank):
    """even per-image split of the global batch -> (start, count) of this rank"""
    if global_batch % world != 0:
        raise ValueError("global batch %d does not split evenly over %d ranks" % (global_batch, world))
    per = global_batch // world
    return rank * per, per


class GradSync:
    """callable handed to SoftIntroEngine: all-reduce(SUM) the flat gradient buffer; the mean's 1/world factor is
    applied by FlatAdam.step(grad_scale).

    Overlap with the backward ("launch on a side stream once the flat buffer's last segment is written"):
    the flat buffer is in module order (stem ... deepest block, fc) and a backward pass finalises gradients deepest
    first, so the TAIL of the buffer — fc and the 512-channel blocks, most of the bytes — is final while the expensive
    shallow layers of the last contributing pass are still running.  `arm(opt)` before `backward()` watches that tail
    with post-accumulate hooks (autograd sums the contributions of all passes in the accumulator's input buffer and
    runs it ONCE per backward, when the last one has arrived); when every tail parameter has been accumulated the tail
    is all-reduced asynchronously (RCCL runs it on its own stream, ordered after the accumulations already enqueued);
    the call after the backward reduces the head and waits for the tail.  Same values, same two buffers per iteration —
    only the issue time moves.  overlap=False (or SIVAE_DP_OVERLAP=0) keeps the single post-backward all-reduce."""

    # The early bucket is the second half of the PARAMETER LIST, i.e. the parameters whose gradients become final in the
    # first half of the backward.  Encoder (stem ... deep blocks, fc): that half holds the 512-channel blocks and fc —
    # > 90 % of the bytes — and it is final while the expensive shallow layers still run.  Decoder (fc, deep blocks ...
    # shallow blocks, predict): the same half is the shallow blocks — few bytes; its fc / 512-channel gradients only
    # become final at the very end of lossD.backward() (every pass reaches them last), so they cannot be overlapped
    # with anything and are reduced after the backward.  (Round 2 cut by bytes — 60 % from the tail — which for the
    # decoder reached into those late blocks and fired the "early" bucket at the end.)

    def __init__(self, overlap=None):
        self.world = world_size()
        self.grad_scale = 1.0 / self.world
        self.calls = 0
        self.bytes = 0
        if overlap is None:
            overlap = os.environ.get("SIVAE_DP_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and self.world > 1
        self._plans = {}   # id(opt) -> (split element offset, tail params)
        self._armed = None
        self.overlapped = 0  # all-reduces whose tail bucket was issued during the backward

    def _plan(self, opt):
        plan = self._plans.get(id(opt))
        if plan is None:
            total = opt.flat_grad.numel()
            split, tail = total, []
            off = total
            n_tail = (len(opt.params) + 1) // 2
            for p in list(reversed(opt.params))[:n_tail]:
                off -= p.numel()
                tail.append(p)
                split = off
            if split == 0:  # (a single giant tensor: nothing left to overlap with)
                tail, split = [], total
            plan = (split, tail)
            self._plans[id(opt)] = plan
            self._hooks_for(opt, tail)
        return plan

    def _hooks_for(self, opt, tail):
        def final(p, st):
            """parameter p's gradient is complete for this backward"""
            if id(p) in st["seen"]:
                return
            st["seen"].add(id(p))
            st["left"] -= 1
            if st["left"] == 0 and st["handle"] is None:
                if getattr(opt, "slabs", None):
                    opt.fold_slabs(st["split"], None)  # the tail's per-use slabs -> flat buffer, then reduce it
                    st["folded"] = True
                seg = opt.flat_grad[st["split"]:]
                st["handle"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)

        def make(p):
            def hook(_param):  # autograd accumulated p's gradient (runs once per backward per parameter)
                st = self._armed
                if st is None or st["opt"] is not opt:
                    return
                final(p, st)

            def wrote(_param):  # one USE of p wrote its gradient slab (functional._done): final after the last use
                st = self._armed
                if st is None or st["opt"] is not opt:
                    return
                c = st["writes"].get(id(p), 0) + 1
                st["writes"][id(p)] = c
                if c >= p.__dict__.get("_sivae_use", 0):
                    final(p, st)
            return hook, wrote
        for p in tail:
            hook, wrote = make(p)
            p.register_post_accumulate_grad_hook(hook)
            p.__dict__["_sivae_on_grad"] = wrote

    def arm(self, opt):
        """call right before the backward() that fills `opt`'s gradient buffer"""
        if not self.overlap:
            self._armed = None
            return
        split, tail = self._plan(opt)
        if not tail:
            self._armed = None
            return
        self._armed = dict(opt=opt, seen=set(), left=len(tail), split=split, handle=None, writes={}, folded=False)

    def __call__(self, opt_or_flat):
        """after backward(): `opt_or_flat` is the FlatAdam whose gradient buffer was just filled (its per-use slabs are
        folded here) or a bare flat gradient tensor"""
        opt = opt_or_flat if hasattr(opt_or_flat, "flat_grad") else None
        flat_grad = opt.flat_grad if opt is not None else opt_or_flat
        slabs = opt is not None and bool(getattr(opt, "slabs", None))
        self.calls += 1
        self.bytes += flat_grad.numel() * 4
        st, self._armed = self._armed, None
        if st is not None and st["handle"] is not None and st["opt"].flat_grad.data_ptr() == flat_grad.data_ptr():
            if st["split"] > 0:
                if slabs:
                    opt.fold_slabs(0, st["split"])
                allreduce_sum_(flat_grad[:st["split"]])
            st["handle"].wait()  # (the current stream waits for the collective; the host does not block on NCCL)
            self.overlapped += 1
            return
        if slabs:
            opt.fold_slabs()
        allreduce_sum_(flat_grad)


def enable_sync_bn(enable=True):
    """Opt-in synchronised BatchNorm: batch statistics and the two BatchNorm-backward means over the
    GLOBAL batch — ~340 tiny fp64 all-reduces per iteration at 256x256, latency-bound — so that N shards of B/N
    reproduce the single-process batch-B iteration.  Default is local BatchNorm."""
    from . import ops
    if not enable or not dist.is_initialized() or dist.get_world_size() == 1:
        ops.SYNC_BN = None
        return False

    def sync(sums):
        if sums is not None:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        return dist.get_world_size()

    ops.SYNC_BN = sync
    return True
