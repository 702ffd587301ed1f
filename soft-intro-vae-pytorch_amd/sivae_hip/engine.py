"""The Soft-IntroVAE training iteration (E-step / D-step / vanilla-VAE step) on the HIP blocks.

One engine serves `train_soft_intro_vae` and `train_soft_intro_vae_bootstrap` (and bench.py):
  reference schedule   soft_intro_vae/train_soft_intro_vae.py:542-624 (vanilla branch :512-540)
  bootstrap deltas     soft_intro_vae_bootstrap/train_soft_intro_vae_bootstrap.py:576-652
The pass list, detach points and requires_grad toggles are the reference's; what changes is where the
work runs (fused HIP blocks), where the Gaussian draws come from (device Philox stream, or injected
tensors for parity tests), that gradients live in two flat buffers (one all-reduce per network under
data parallelism) and that all logged scalars come back in ONE device->host copy per iteration instead of
~12 `.item()` syncs.
"""
import os

import torch

from . import functional as SF
from . import rng

# The iteration contains four PAIRS of independent passes through unchanged weights: model(rec.detach()) /
# model(fake.detach()) = encoder pair + second-decoder pair (reference :567-568), encode(rec) / encode(fake) (:601-605),
# decode(z_rec) / decode(z_fake) (:607-608; bootstrap decode_target :635-636).  With SIVAE_PAIR_PASSES != 0 each pair
# runs as ONE segmented batch of 2B images (per-pass BatchNorm statistics, functional.py "segments"): half the launches
# and twice the work per launch for 8 of the 13 forward and 8 of the 12 backward passes — what the small per-GPU shards
# of the data-parallel configurations (16 / 8 images) need (256x256: 16-image shard 210 -> 221 img/s, bootstrap 8-image shard
# 182 -> 202; batch 128: +0.4 %).  "auto" (default) / "1": pair wherever the shapes allow; "0": never.
PAIR_PASSES = os.environ.get("SIVAE_PAIR_PASSES", "auto")


# ---- reference helper surface (same names / argument meaning / errors) ------------------------------
def reparameterize(mu, logvar, eps=None):
    """z = mu + eps * exp(0.5 logvar); eps ~ N(0, I) from the device Philox stream unless given."""
    if eps is None:
        eps = rng.randn(mu.shape, mu.device)
    return SF.reparameterize(mu, logvar, eps)


def calc_kl(logvar, mu, mu_o=0.0, logvar_o=0.0, reduce="sum"):
    """reduce in {'sum', 'mean'}; anything else returns the per-sample vector (reference :247-251).
    mu_o / logvar_o: numbers, or tensors broadcastable to [B, Z] (the reference wraps numbers into tensors, :237-243);
    tensors stay on the device — no .item() sync."""
    red = reduce if reduce in ("sum", "mean") else "none"
    return SF.kl(logvar, mu, mu_o, logvar_o, red)


def calc_reconstruction_loss(x, recon_x, loss_type="mse", reduction="sum"):
    if reduction not in ("sum", "mean", "none"):
        raise NotImplementedError
    if loss_type not in ("mse", "l1", "bce"):
        raise NotImplementedError
    B = x.size(0)
    D = x.numel() // B
    if loss_type == "mse":
        if reduction == "none":
            return SF.ReconFn.apply(x, recon_x, "mse", "rows", 1.0)
        return SF.ReconFn.apply(x, recon_x, "mse", "total", 1.0 if reduction == "sum" else 1.0 / B)
    if reduction == "none":
        return SF.ReconFn.apply(x, recon_x, loss_type, "elem", 1.0).view(B, D)
    return SF.ReconFn.apply(x, recon_x, loss_type, "total", 1.0 if reduction == "sum" else 1.0 / (B * D))


_HALVES_BY_SLICE = os.environ.get("SIVAE_HALVES_SLICE", "0") == "1"  # (A/B switch of the note in e_step)


def _halves(t):
    """the two passes of a paired batch"""
    if _HALVES_BY_SLICE:
        n = t.shape[0] // 2
        return t[:n], t[n:]
    return t.chunk(2)


def _recon_rows(x, recon_x, loss_type):
    """per-sample reconstruction sums [B] — what the reference builds for the exp-ELBO terms as
    calc_reconstruction_loss(..., reduction='none') followed by `while len(shape) > 1: sum(-1)` (:574-579).  For l1 / bce
    the 'none' result is the [B, D] element tensor; the engine goes straight to the row sums (one fused kernel, 2-stage
    fp64) instead of materialising it."""
    if loss_type not in ("mse", "l1", "bce"):
        raise NotImplementedError
    return SF.ReconFn.apply(x, recon_x, loss_type, "rows", 1.0)


STAT_NAMES = ("lossE", "lossD", "loss_rec", "kl_real", "kl_fake", "kl_rec", "expelbo_rec", "expelbo_fake")


class SoftIntroEngine:
    """Owns the two flat-buffer optimizers and runs iterations on a SoftIntroVAE model."""

    def __init__(self, model, opt_e, opt_d, beta_kl=1.0, beta_rec=1.0, beta_neg=1.0, gamma_r=1e-8,
                 recon_loss_type="mse", bootstrap=False, grad_sync=None, reuse_decoder_forward=True,
                 compute_dtype=None, pair_passes=None):
        """compute_dtype: None (leave the model as it is), "fp32" (the parity path) or "bf16" (BASELINE.json config 3's
        build-defined mode: bf16 activation storage and bf16 MFMA convs with fp32 accumulation — sivae_hip.nn
        .set_compute_dtype; weights, BatchNorm statistics, losses, gradients buffers and Adam stay fp32)."""
        if compute_dtype is not None:
            from .nn import set_compute_dtype
            set_compute_dtype(model, compute_dtype)
        self.model, self.opt_e, self.opt_d = model, opt_e, opt_d
        self.beta_kl, self.beta_rec, self.beta_neg, self.gamma_r = beta_kl, beta_rec, beta_neg, gamma_r
        self.loss_type = recon_loss_type
        self.bootstrap = bootstrap
        self.grad_sync = grad_sync  # dp.GradSync (all-reduce SUM of the flat gradient) or None
        self.grad_scale = getattr(grad_sync, "grad_scale", 1.0)  # 1/world, applied inside the fused Adam
        self.last_z = None
        # The D-step's `fake = sample(noise)` and `rec = decoder(z)` (reference :597-598) recompute, with an
        # UNCHANGED decoder (only the encoder was stepped in between), exactly what the E-step computed at
        # :557,:561.  With reuse on, the E-step passes fill an activation cache and the D-step passes replay it:
        # bit-identical outputs and saved tensors, BatchNorm running statistics still updated once per reference
        # pass, 2 of 8 decoder forwards (50.3 of 818.7 GFLOP per image at 256x256) not re-executed.
        self.reuse_decoder_forward = reuse_decoder_forward
        self._cache_fake, self._cache_rec = None, None
        self._cache_pair, self._zn, self._zn_src = None, None, None
        # pair_passes: None -> SIVAE_PAIR_PASSES ("auto" | "0" | "1"); True / False force it
        self.pair_passes = PAIR_PASSES if pair_passes is None else ("1" if pair_passes else "0")
        # parameter gradients of every pass go straight into per-use slabs of the flat optimizers and are folded with
        # one launch per network after each backward (functional._claim / FlatAdam.fold_slabs; SIVAE_DIRECT_GRADS=0:
        # autograd's per-tensor accumulation)
        from . import functional as _SF
        if _SF.DIRECT_GRADS:
            for opt in (opt_e, opt_d):
                if hasattr(opt, "enable_slabs"):
                    opt.enable_slabs(4)  # the decoder runs four times inside lossD (fake, rec, rec_rec, rec_fake)

    # -- whole-iteration HIP graph -------------------------------------------------------
    def capture(self, real_example, warmup=2):
        """Capture one full Soft-Intro iteration (both steps, both Adam updates, ~2-3 thousand launches) into a
        HIP graph.  Everything that changes between iterations lives in device memory — the Adam step counters
        and learning rates (`FlatAdam.use_device_state`), the Philox stream position, the BatchNorm counters, the
        input batch (a static buffer) — so a replay is one `hipGraphLaunch` instead of a python-driven launch
        sequence.  At 256x256 / batch 128 the GPU is the bottleneck either way; at 32x32 or batch 16 the host is.
        `warmup` eager iterations run first on the capture stream (they are REAL training iterations)."""
        if self.grad_sync is not None:
            raise RuntimeError("sivae_hip: HIP-graph capture of the data-parallel step is not supported")
        dev = real_example.device
        self.opt_e.use_device_state()
        self.opt_d.use_device_state()
        rng.default_stream().use_device_state(dev)
        self._g_real = real_example.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.soft_intro_step(self._g_real)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        SF.clear_pack_cache()  # every weight re-pack of an iteration must be part of the captured sequence
        torch.cuda.empty_cache()  # the graph gets its own pool: hand the warm-up's cached blocks back first
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            out = self.soft_intro_step(self._g_real)
        self._g_out = {"stats": out["stats"], "fake": out["fake"]}
        self.opt_e.t -= 1  # the capture pass launched nothing: undo its python-side counters
        self.opt_d.t -= 1
        return self

    def replay(self, real):
        """run the captured iteration on `real` (same shape as the example); returns the graph's static outputs"""
        self._g_real.copy_(real, non_blocking=True)
        self._graph.replay()
        self.opt_e.replayed()
        self.opt_d.replayed()
        return self._g_out

    # -- requires_grad toggles (reference :552-555, :592-595) -----------------------------------------
    def _train_encoder_only(self):
        for p in self.model.encoder.parameters():
            p.requires_grad = True
        for p in self.model.decoder.parameters():
            p.requires_grad = False
        if self.bootstrap:
            for p in self.model.target_decoder.parameters():
                p.requires_grad = False

    def _train_decoder_only(self):
        for p in self.model.encoder.parameters():
            p.requires_grad = False
        for p in self.model.decoder.parameters():
            p.requires_grad = True
        if self.bootstrap:
            for p in self.model.target_decoder.parameters():
                p.requires_grad = False

    def _second_decoder(self):
        return self.model.target_decoder if self.bootstrap else self.model.decoder

    def _paired(self, real):
        """run the same-weight pass pairs of this iteration as segmented batches?"""
        if self.pair_passes == "0":
            return False
        from .nn import segments_supported
        enc = self.model.encoder
        ok = segments_supported(real.size(2), real.size(0), getattr(enc, "compute_dtype", "fp32")) \
            and real.size(2) == real.size(3) and not getattr(enc, "conditional", False)
        return ok

    @staticmethod
    def _eps2(e0, e1, like):
        """the Gaussian draws of a pass pair, in the reference's order (two draws from the stream when not injected)"""
        if e0 is None:
            e0 = rng.randn(like.shape, like.device)
        if e1 is None:
            e1 = rng.randn(like.shape, like.device)
        return torch.cat([e0, e1])

    def _sync(self, opt):
        """after backward(): fold the per-use gradient slabs into the flat buffer, all-reduce it (data parallel)"""
        if self.grad_sync is not None:
            self.grad_sync(opt)
        else:
            opt.fold_slabs()

    def _arm(self, opt):
        """let the gradient synchroniser start reducing the early-final tail of `opt`'s flat gradient buffer while the
        coming backward is still running (dp.GradSync)"""
        if self.grad_sync is not None and hasattr(self.grad_sync, "arm"):
            self.grad_sync.arm(opt)

    # -- vanilla VAE step (reference :516-533) ----------------------------------------------------------
    def vae_step(self, real, eps=None, keep=False):
        m = self.model
        for p in m.encoder.parameters():
            p.requires_grad = True
        for p in m.decoder.parameters():
            p.requires_grad = True
        mu, logvar = m.encode(real)
        z = reparameterize(mu, logvar, eps)
        rec = self._second_decoder()(z) if self.bootstrap else m.decoder(z)
        loss_rec = calc_reconstruction_loss(real, rec, self.loss_type, "mean")
        loss_kl = calc_kl(logvar, mu, reduce="mean")
        loss = self.beta_rec * loss_rec + self.beta_kl * loss_kl
        self.opt_d.zero_grad()
        self.opt_e.zero_grad()
        loss.backward()
        self._sync(self.opt_e)
        self.opt_e.step(self.grad_scale)
        if not self.bootstrap:
            # (bootstrap: the reconstruction came from the frozen target decoder, :546 with target=True, so the decoder
            # has no gradient; torch.optim.Adam skips parameters whose .grad is None — neither the weights nor the
            # per-parameter step counts move — and so does this)
            self._sync(self.opt_d)
            self.opt_d.step(self.grad_scale)
        res = {"loss": loss.detach(), "loss_rec": loss_rec.detach(), "loss_kl": loss_kl.detach(), "rec": rec.detach()}
        if keep:
            res.update(mu=mu.detach(), logvar=logvar.detach(), z=z.detach())
        return res

    # -- Soft-Intro iteration (reference :547-624) --------------------------------------------------------
    def soft_intro_step(self, real, noise=None, eps=None, keep=False):
        """One full iteration: E-step + Adam(encoder) + D-step + Adam(decoder).
        eps: optional list of the five Gaussian draws in the reference's order. Returns a dict of
        detached device tensors (no host sync); `keep=True` also returns the image-sized intermediates."""
        if noise is None:
            noise = rng.randn((real.size(0), self.model.zdim), real.device)
        e = eps if eps is not None else [None] * 5
        out = {}
        es = self.e_step(real, noise, e[:3], keep)
        ds = self.d_step(real, noise, es["z"], e[3:], keep)
        if keep:
            out["E"], out["D"] = es["kept"], ds["kept"]
        # one small stats vector -> ONE device->host copy when the caller wants numbers
        out["stats"] = torch.stack([es["lossE"], ds["lossD"], ds["loss_rec"], es["kl_real"], ds["kl_fake"],
                                    ds["kl_rec"], es["expelbo_rec"], es["expelbo_fake"]])
        out["fake"] = ds["fake"]
        return out

    def e_step(self, real, noise, eps=(None, None, None), keep=False):
        """Encoder update (reference :551-589): forward passes, lossE, backward, [grad sync], Adam(encoder)."""
        m = self.model
        scale = 1.0 / (real.size(1) * real.size(2) * real.size(3))
        br, bk, bn, lt = self.beta_rec, self.beta_kl, self.beta_neg, self.loss_type
        dec2 = self._second_decoder()
        self._train_encoder_only()
        self._cache_fake = {} if self.reuse_decoder_forward else None
        self._cache_rec = {} if self.reuse_decoder_forward else None
        self._cache_pair, self._zn = None, None
        B = real.size(0)
        paired = self._paired(real)
        pair_dec = paired and self.reuse_decoder_forward
        if pair_dec:
            # sample(noise) and decoder(z) (:557,:561) as ONE segmented batch [z; noise] -> [rec; fake].  `fake` has no
            # graph (frozen decoder, leaf noise) and `rec` only needs a data gradient, so the pair runs WITHOUT a graph
            # and fills a replay cache; the graph of `rec` is then a replay of segment 0 of that cache (no kernels), and
            # the D-step replays the whole pair (:597-598).  The running statistics see `fake` first (seg_rev), as in
            # the reference's call order.
            real_mu, real_logvar = m.encode(real)
            z = reparameterize(real_mu, real_logvar, eps[0])
            self._cache_pair = {}
            with torch.no_grad():
                self._zn = torch.cat([z.detach(), noise])
                self._zn_src = (z.data_ptr(), z._version, noise.data_ptr(), noise._version)
                y2 = m.decoder(self._zn, cache=self._cache_pair, nseg=2, seg_rev=True)
            fake = y2[B:]
            view = SF.cache_segment(self._cache_pair, 0, 2)
            if view is None:
                raise RuntimeError("sivae_hip: the paired decoder pass did not fill its replay cache")
            rec = m.decoder(z, cache=view, replay_update=False, check_input=False)
        else:
            fake = m.decoder(noise, cache=self._cache_fake)
            real_mu, real_logvar = m.encode(real)
            z = reparameterize(real_mu, real_logvar, eps[0])
            rec = m.decoder(z, cache=self._cache_rec)
        loss_rec = calc_reconstruction_loss(real, rec, lt, "mean")
        kl_real = calc_kl(real_logvar, real_mu, reduce="mean")

        if paired:
            # model(rec.detach()) and model(fake.detach()) (:567-568) as one segmented batch [rec; fake]
            x2 = y2 if pair_dec else torch.cat([rec.detach(), fake.detach()])
            mu2, logvar2 = m.encoder(x2, nseg=2)
            z2 = reparameterize(mu2, logvar2, self._eps2(eps[1], eps[2], real_mu))
            rr2 = dec2(z2, nseg=2)
            # (chunk, not two slices: ONE autograd node per tensor whose backward is a single cat — a slice's backward
            # allocates a zero tensor of the whole pair, copies its half in and adds the two: ~35 small launches and
            # 3 GB of traffic per iteration at batch 128)
            (rec_mu, fake_mu), (rec_logvar, fake_logvar) = _halves(mu2), _halves(logvar2)
            rec_rec, rec_fake = _halves(rr2)
        else:
            rec_mu, rec_logvar = m.encode(rec.detach())
            z_rec = reparameterize(rec_mu, rec_logvar, eps[1])
            rec_rec = dec2(z_rec)
            fake_mu, fake_logvar = m.encode(fake.detach())
            z_fake = reparameterize(fake_mu, fake_logvar, eps[2])
            rec_fake = dec2(z_fake)

        kl_rec = calc_kl(rec_logvar, rec_mu, reduce="none")
        kl_fake = calc_kl(fake_logvar, fake_mu, reduce="none")
        l_rec_rec = _recon_rows(rec, rec_rec, lt)
        l_rec_fake = _recon_rows(fake, rec_fake, lt)
        expelbo_rec = SF.expelbo(l_rec_rec, kl_rec, scale, br, bn)
        expelbo_fake = SF.expelbo(l_rec_fake, kl_fake, scale, br, bn)
        # lossE = scale * (br * loss_rec + bk * kl_real) + 0.25 * (expelbo_rec + expelbo_fake)  (:583-586), one launch
        lossE = SF.lincomb([loss_rec, kl_real, expelbo_rec, expelbo_fake], [scale * br, scale * bk, 0.25, 0.25])
        self.opt_e.zero_grad()
        self._arm(self.opt_e)
        lossE.backward()
        self._sync(self.opt_e)
        self.opt_e.step(self.grad_scale)
        res = dict(z=z.detach(), lossE=lossE.detach(), kl_real=kl_real.detach(), expelbo_rec=expelbo_rec.detach(),
                   expelbo_fake=expelbo_fake.detach())
        if keep:
            res["kept"] = dict(fake=fake.detach(), real_mu=real_mu.detach(), real_logvar=real_logvar.detach(),
                               z=z.detach(), rec=rec.detach(), loss_rec=loss_rec.detach(),
                               kl_real=kl_real.detach(), rec_mu=rec_mu.detach(), rec_logvar=rec_logvar.detach(),
                               rec_rec=rec_rec.detach(), fake_mu=fake_mu.detach(),
                               fake_logvar=fake_logvar.detach(), rec_fake=rec_fake.detach(),
                               kl_rec=kl_rec.detach(), kl_fake=kl_fake.detach(),
                               expelbo_rec=expelbo_rec.detach(), expelbo_fake=expelbo_fake.detach(),
                               lossE=lossE.detach())
        self.last_z = res["z"]
        return res

    def d_step(self, real, noise, z, eps=(None, None), keep=False):
        """Decoder update (reference :591-624; bootstrap :620-652). z = the E-step latent (detached)."""
        m = self.model
        scale = 1.0 / (real.size(1) * real.size(2) * real.size(3))
        br, bk, gr, lt = self.beta_rec, self.beta_kl, self.gamma_r, self.loss_type
        dec2 = self._second_decoder()
        self._train_decoder_only()
        B = real.size(0)
        y2 = None
        if (self._cache_pair is not None and self._zn is not None and self._zn.shape[0] == 2 * B
                and self._zn_src == (z.data_ptr(), z._version, noise.data_ptr(), noise._version)):
            # replay of the E-step's [rec; fake] pair (same inputs, unchanged decoder), now WITH a graph: one segmented
            # backward instead of two
            y2 = m.decoder(self._zn, cache=self._cache_pair, nseg=2, seg_rev=True)
            rec, fake = _halves(y2)
        else:
            fake = m.decoder(noise, cache=self._cache_fake)
            rec = m.decoder(z.detach(), cache=self._cache_rec)
        self._cache_fake, self._cache_rec, self._cache_pair, self._zn = None, None, None, None
        loss_rec = calc_reconstruction_loss(real, rec, lt, "mean")
        if self._paired(real):
            # encode(rec) / encode(fake) (:601-605) and decode(z_rec) / decode(z_fake) (:607-608; bootstrap
            # decode_target :635-636) as segmented batches [rec; fake]
            mu2, logvar2 = m.encoder(y2 if y2 is not None else torch.cat([rec, fake]), nseg=2)
            z2 = reparameterize(mu2, logvar2, self._eps2(eps[0], eps[1], z))
            rr2 = dec2(z2, nseg=2) if self.bootstrap else m.decoder(z2.detach(), nseg=2)
            (rec_mu, fake_mu), (rec_logvar, fake_logvar) = _halves(mu2), _halves(logvar2)
            rec_rec, rec_fake = _halves(rr2)
        else:
            rec_mu, rec_logvar = m.encode(rec)
            z_rec = reparameterize(rec_mu, rec_logvar, eps[0])
            fake_mu, fake_logvar = m.encode(fake)
            z_fake = reparameterize(fake_mu, fake_logvar, eps[1])
            if self.bootstrap:
                rec_rec = dec2(z_rec)
                rec_fake = dec2(z_fake)
            else:
                rec_rec = m.decode(z_rec.detach())
                rec_fake = m.decode(z_fake.detach())
        if self.bootstrap:
            l_rr = calc_reconstruction_loss(rec, rec_rec, lt, "mean")
            l_fr = calc_reconstruction_loss(fake, rec_fake, lt, "mean")
        else:
            l_rr = calc_reconstruction_loss(rec.detach(), rec_rec, lt, "mean")
            l_fr = calc_reconstruction_loss(fake.detach(), rec_fake, lt, "mean")
        kl_rec = calc_kl(rec_logvar, rec_mu, reduce="mean")
        kl_fake = calc_kl(fake_logvar, fake_mu, reduce="mean")
        # lossD = scale * (loss_rec * br + (kl_rec + kl_fake) * 0.5 * bk + gr * 0.5 * br * (l_rr + l_fr))  (:618-620)
        lossD = SF.lincomb([loss_rec, kl_rec, kl_fake, l_rr, l_fr],
                           [scale * br, scale * 0.5 * bk, scale * 0.5 * bk, scale * gr * 0.5 * br, scale * gr * 0.5 * br])
        self.opt_d.zero_grad()
        self._arm(self.opt_d)
        lossD.backward()
        self._sync(self.opt_d)
        self.opt_d.step(self.grad_scale)
        res = dict(lossD=lossD.detach(), loss_rec=loss_rec.detach(), kl_rec=kl_rec.detach(),
                   kl_fake=kl_fake.detach(), fake=fake.detach())
        if keep:
            res["kept"] = dict(fake=fake.detach(), rec=rec.detach(), loss_rec=loss_rec.detach(),
                               rec_mu=rec_mu.detach(), rec_logvar=rec_logvar.detach(), fake_mu=fake_mu.detach(),
                               fake_logvar=fake_logvar.detach(), rec_rec=rec_rec.detach(),
                               rec_fake=rec_fake.detach(), loss_rec_rec=l_rr.detach(),
                               loss_fake_rec=l_fr.detach(), kl_rec=kl_rec.detach(), kl_fake=kl_fake.detach(),
                               lossD=lossD.detach())
        return res
