"""Flat-buffer Adam driven by one fused HIP kernel per network.

The parameters of a network are re-pointed at views of ONE contiguous fp32 buffer (values preserved) and
their .grad at views of one flat gradient buffer, so that
  * the optimizer step is a single launch (sivae_adam_step) instead of ~6 elementwise launches per tensor,
  * data-parallel gradient averaging is ONE RCCL all-reduce per network per iteration (see dp.py).
Semantics follow torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) as used at
soft_intro_vae/train_soft_intro_vae.py:450-454; MultiStepLR is `set_lr`/`MultiStepLR` below.
"""
import torch

from . import functional as SF
from . import ops


class FlatAdam:
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs ROCm-device parameters (no CPU fallback)")
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.t = 0
        self.dev_state = None
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                off += k
        self.param_groups = [{"lr": self.lr, "params": self.params}]  # torch-like surface for schedulers
        self.slabs = []        # per-use gradient slabs (enable_slabs): flat buffers laid out like flat_grad
        self._slabs_used = 0   # slabs that hold gradients of the current backward
        SF.bump_generation(self.params)

    def enable_slabs(self, k=4):
        """Give every parameter k gradient slabs (functional._claim / _dst): each USE of a parameter inside one
        backward writes its gradient straight into its own slab instead of going through autograd's accumulation
        (one torch add per tensor and use), and `fold_slabs` sums them into flat_grad with one launch.  The owner of
        the step (SoftIntroEngine) must call zero_grad() before and fold_slabs() after every backward."""
        if self.slabs:
            return
        n = self.flat_grad.numel()
        self.slabs = [torch.zeros(n, dtype=torch.float32, device=self.flat_grad.device) for _ in range(k)]
        off = 0
        for p in self.params:
            c = p.numel()
            p.__dict__["_sivae_slabs"] = [s[off:off + c].view(p.shape) for s in self.slabs]
            p.__dict__["_sivae_use"] = 0
            off += c

    def disable_slabs(self):
        for p in self.params:
            p.__dict__.pop("_sivae_slabs", None)
            p.__dict__.pop("_sivae_use", None)
        self.slabs = []

    def fold_slabs(self, lo=0, hi=None):
        """flat_grad[lo:hi] += the slabs the current backward wrote (call once per element range, after backward)"""
        k = self._slabs_used
        claimed = self._max_use()
        if claimed > k:
            # zero_grad() ran BEFORE the forward passes reserved their slabs: the slabs were not cleaned for this
            # backward, so folding them would add stale gradients — fail loudly instead of stepping on garbage
            raise RuntimeError("FlatAdam: %d gradient slabs were claimed but only %d prepared — with enable_slabs() "
                               "call zero_grad() AFTER the forward passes and fold_slabs() after backward()"
                               % (claimed, k))
        self._folded = True
        if k == 0:
            return
        hi = self.flat_grad.numel() if hi is None else hi
        if hi > lo:
            ops.sum_slabs(self.flat_grad[lo:hi], [s[lo:hi] for s in self.slabs[:k]])

    def _max_use(self):
        return min(len(self.slabs), max((p.__dict__.get("_sivae_use", 0) for p in self.params), default=0)) \
            if self.slabs else 0

    def _reset_uses(self):
        for p in self.params:
            if "_sivae_use" in p.__dict__:
                p.__dict__["_sivae_use"] = 0

    def zero_grad(self, set_to_none=False):
        # gradients must stay views of the flat buffer, so they are zeroed in place, never set to None
        self.flat_grad.zero_()
        if self.slabs:
            # (called after the forward passes reserved their slab indices and before backward() fills them:
            # only the slabs this backward will write need to be clean)
            self._slabs_used = self._max_use()
            self._folded = False
            for s_ in self.slabs[:self._slabs_used]:
                s_.zero_()
        for p, g in zip(self.params, self._grad_views()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def _grad_views(self):
        off = 0
        for p in self.params:
            k = p.numel()
            yield self.flat_grad[off:off + k].view(p.shape)
            off += k

    def step(self, grad_scale=1.0):
        if self.slabs and self._max_use() > 0 and not getattr(self, "_folded", False):
            raise RuntimeError("FlatAdam.step(): gradient slabs were written but never folded — call fold_slabs() "
                               "(or the engine's gradient sync) between backward() and step()")
        self.t += 1
        self.lr = float(self.param_groups[0]["lr"])
        if self.dev_state is not None:
            # step count / learning rate / bias corrections live on the device (HIP-graph mode): the launch
            # arguments are the same every iteration
            ops.adam_step_dev(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.dev_state,
                              self.betas[0], self.betas[1], self.eps, grad_scale)
        else:
            ops.adam_step(self.flat, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.t, self.lr, self.betas[0],
                          self.betas[1], self.eps, grad_scale)
        self._reset_uses()
        SF.bump_generation(self.params)
        SF.repack(self.params, self)  # every cached operand form of every weight, one launch per form
        from . import functional16 as SF16
        SF16.repack16(self.params, self)  # (bf16 mode: the bf16 operand slabs, one launch)

    def use_device_state(self):
        """Move the step counter and learning rate into a device tensor {t, lr, ., .} (float64).  Needed before the
        step is captured into a HIP graph; `sync_lr()` pushes a scheduler change to the device."""
        if self.dev_state is None:
            self.dev_state = torch.tensor([float(self.t), float(self.param_groups[0]["lr"]), 0.0, 0.0],
                                          dtype=torch.float64, device=self.flat.device)
        return self.dev_state

    def sync_lr(self):
        if self.dev_state is not None:
            self.dev_state[1:2].fill_(float(self.param_groups[0]["lr"]))

    def replayed(self, n=1):
        """bookkeeping after n HIP-graph replays of a captured step (the kernels ran, this python did not)"""
        self.t += n
        self._reset_uses()
        SF.bump_generation(self.params)


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR semantics for FlatAdam (milestones in scheduler steps)."""

    def __init__(self, optimizer, milestones, gamma=0.1):
        self.opt, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        k = sum(1 for m in self.milestones if m <= self.last_epoch)
        self.opt.param_groups[0]["lr"] = self.base_lr * (self.gamma ** k)
        if hasattr(self.opt, "sync_lr"):
            self.opt.sync_lr()
