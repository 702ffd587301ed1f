"""Generate the golden fixtures from the REAL reference (runs only where /root/reference is mounted).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference has no tests of its own, so parity is pinned by importing it here (torchvision, which the
image lacks, is stubbed in sys.modules — only dataset/image-dump helpers use it) and recording inputs
and outputs as plain arrays:

  helpers.npz        calc_kl / calc_reconstruction_loss / reparameterize on seeded inputs
  step_<case>.npz    one Soft-IntroVAE iteration re-sequenced from the reference's own modules and
                     helpers: initial state_dict, inputs, the Gaussian draws, every intermediate
                     (mu, logvar, z, rec, fake, ...), all loss scalars, selected gradients
  loop_<case>.npz    the reference's UNMODIFIED train_soft_intro_vae() / train_soft_intro_vae_toy()
                     run for a few iterations on synthetic data with every RNG draw and every batch
                     recorded, plus initial and final state_dict — pins the whole E/D schedule, the
                     detach/requires_grad semantics and the Adam updates end to end.

Only narrow channel widths are used so the fixtures stay small; the layer pattern is the reference's.
Nothing from /root/reference is copied: fixtures are inputs/outputs only.
"""
import importlib
import importlib.machinery
import os
import sys
import tempfile
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_torchvision():
    for name in ["torchvision", "torchvision.utils", "torchvision.datasets", "torchvision.transforms",
                 "torchvision.models"]:
        m = mock.MagicMock(name=name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m


def _import_ref(subdir, module):
    for k in [k for k in sys.modules if k in ("dataset", "metrics", module) or k.startswith("metrics.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, subdir))
    try:
        return importlib.import_module(module)
    finally:
        sys.path.pop(0)


def _np(t):
    return t.detach().cpu().numpy().copy()


def _sd(model, prefix):
    return {prefix + k: _np(v) for k, v in model.state_dict().items()}


def _keep_grad(k):
    """gradients stored in the fixtures: first/last conv, fc, one mid block, a BatchNorm"""
    return (k.startswith(("main.0.", "main.1.", "fc.", "main.predict.")) or ".res_in_8." in k)


class RandnRecorder:
    """records every torch.randn / torch.randn_like result (in call order) while active"""

    def __init__(self):
        self.draws = []
        self._randn, self._randn_like = torch.randn, torch.randn_like

    def __enter__(self):
        rec = self

        def randn(*a, **k):
            out = rec._randn(*a, **k)
            rec.draws.append(_np(out))
            return out

        def randn_like(*a, **k):
            out = rec._randn_like(*a, **k)
            rec.draws.append(_np(out))
            return out

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


# --------------------------------------------------------------------------------------------------
def make_helpers(T):
    out = {}
    g = torch.Generator().manual_seed(7)
    mu = torch.randn(6, 10, generator=g)
    lv = torch.randn(6, 10, generator=g) * 0.5
    out["mu"], out["logvar"] = _np(mu), _np(lv)
    for red in ("sum", "mean", "none"):
        out["kl_%s" % red] = _np(T.calc_kl(lv, mu, reduce=red))
        out["kl_o_%s" % red] = _np(T.calc_kl(lv, mu, mu_o=0.3, logvar_o=-0.7, reduce=red))
    x = torch.rand(5, 3, 4, 4, generator=g)
    r = torch.rand(5, 3, 4, 4, generator=g) * 0.98 + 0.01
    out["x"], out["recon"] = _np(x), _np(r)
    for lt in ("mse", "l1", "bce"):
        for red in ("sum", "mean", "none"):
            out["rec_%s_%s" % (lt, red)] = _np(T.calc_reconstruction_loss(x, r, loss_type=lt, reduction=red))
    with RandnRecorder() as rr:
        z = T.reparameterize(mu, lv)
    out["reparam_eps"], out["reparam_z"] = rr.draws[0], _np(z)
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), **out)
    print("helpers.npz", len(out))


IMAGE_KEYS = ("fake", "rec", "rec_rec", "rec_fake")


def _thin(out, key, t):
    """large image tensors of the 128x128 / 256x256 fixtures: every 4th pixel in both directions + per-image,
    per-channel sums and sums of squares (keeps a 256x256 fixture at a few MB; the input batch is stored in full)"""
    a = _np(t)
    out[key + "@thin"] = a[:, :, ::4, ::4].copy()
    out[key + "@sum"] = a.astype(np.float64).sum((2, 3))
    out[key + "@sumsq"] = (a.astype(np.float64) ** 2).sum((2, 3))


def make_vae_step(T, name, cdim, zdim, channels, image_size, B, hp, bootstrap=False, seed=0):
    """One vanilla-VAE iteration (the `epoch < num_vae` branch, train_soft_intro_vae.py:512-533; the bootstrap file's
    branch decodes through the target decoder, train_soft_intro_vae_bootstrap.py:546 with target=True)."""
    torch.manual_seed(seed)
    model = T.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    model.train()
    out = {"meta_cdim": cdim, "meta_zdim": zdim, "meta_channels": np.array(channels), "meta_image_size": image_size,
           "meta_bootstrap": int(bootstrap)}
    for k, v in hp.items():
        out["hp_" + k] = v
    out.update(_sd(model, "init/"))
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, cdim, image_size, image_size, generator=g)
    out["real"] = _np(real)
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=hp["lr"])
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=hp["lr"])
    with RandnRecorder() as rr:
        real_mu, real_logvar, z, rec = model(real)
        loss_rec = T.calc_reconstruction_loss(real, rec, loss_type="mse", reduction="mean")
        loss_kl = T.calc_kl(real_logvar, real_mu, reduce="mean")
        loss = hp["beta_rec"] * loss_rec + hp["beta_kl"] * loss_kl
        opt_d.zero_grad()
        opt_e.zero_grad()
        loss.backward()
        for k, v in dict(real_mu=real_mu, real_logvar=real_logvar, z=z, rec=rec, loss_rec=loss_rec, loss_kl=loss_kl,
                         loss=loss).items():
            out["V/" + k] = _np(v)
        for net, pre in ((model.encoder, "encoder."), (model.decoder, "decoder.")):
            for k, p in net.named_parameters():
                if p.grad is not None:
                    out["V/grad/" + pre + k] = _np(p.grad)
        opt_e.step()
        opt_d.step()
    assert len(rr.draws) == 1
    out["eps0"] = rr.draws[0]
    out.update(_sd(model, "final/"))
    np.savez_compressed(os.path.join(OUT, "step_%s.npz" % name), **out)
    print("step_%s.npz" % name, "loss=%.6g" % float(loss))


def make_step(T, name, cdim, zdim, channels, image_size, B, hp, bootstrap=False, seed=0, thin=False, loss_type="mse"):
    """One iteration re-sequenced from the reference's modules (order of train_soft_intro_vae.py:547-624).
    loss_type: the `recon_loss_type` argument of train_soft_intro_vae (:288-291 and the `while len(shape) > 1: sum(-1)`
    consumers :574-578).  "bce": F.binary_cross_entropy needs reconstructions inside (0, 1) and the decoder ends in a bare
    conv, so the INITIAL predict layer is rescaled (weight * 0.02, bias = 0.5) before the state is recorded — the fixture
    carries that initial state like any other."""
    torch.manual_seed(seed)
    model = T.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    model.train()
    if loss_type == "bce":
        with torch.no_grad():
            for dec in [model.decoder] + ([model.target_decoder] if bootstrap else []):
                dec.main.predict.weight.mul_(0.02)
                dec.main.predict.bias.fill_(0.5)
    out = {"meta_cdim": cdim, "meta_zdim": zdim, "meta_channels": np.array(channels), "meta_image_size": image_size,
           "meta_bootstrap": int(bootstrap)}
    for k, v in hp.items():
        out["hp_" + k] = v
    out["meta_recon_loss_type"] = np.array(loss_type)
    out.update(_sd(model, "init/"))
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, cdim, image_size, image_size, generator=g)
    noise = torch.randn(B, zdim, generator=g)
    out["real"], out["noise"] = _np(real), _np(noise)
    scale = 1 / (cdim * image_size ** 2)
    br, bk, bn, gr = hp["beta_rec"], hp["beta_kl"], hp["beta_neg"], hp["gamma_r"]
    opt_e = torch.optim.Adam(model.encoder.parameters(), lr=hp["lr"])
    opt_d = torch.optim.Adam(model.decoder.parameters(), lr=hp["lr"])
    frozen = [model.decoder] + ([model.target_decoder] if bootstrap else [])

    with RandnRecorder() as rr:
        # ---- E step
        for p in model.encoder.parameters():
            p.requires_grad = True
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False
        fake = model.sample(noise)
        real_mu, real_logvar = model.encode(real)
        z = T.reparameterize(real_mu, real_logvar)
        rec = model.decoder(z)
        loss_rec = T.calc_reconstruction_loss(real, rec, loss_type=loss_type, reduction="mean")
        kl_real = T.calc_kl(real_logvar, real_mu, reduce="mean")
        rec_mu, rec_logvar, z_rec, rec_rec = model(rec.detach())
        fake_mu, fake_logvar, z_fake, rec_fake = model(fake.detach())
        kl_rec = T.calc_kl(rec_logvar, rec_mu, reduce="none")
        kl_fake = T.calc_kl(fake_logvar, fake_mu, reduce="none")
        l_rr = T.calc_reconstruction_loss(rec, rec_rec, loss_type=loss_type, reduction="none")
        while len(l_rr.shape) > 1:  # (:575-576)
            l_rr = l_rr.sum(-1)
        l_rf = T.calc_reconstruction_loss(fake, rec_fake, loss_type=loss_type, reduction="none")
        while len(l_rf.shape) > 1:  # (:578-579)
            l_rf = l_rf.sum(-1)
        expelbo_rec = (-2 * scale * (br * l_rr + bn * kl_rec)).exp().mean()
        expelbo_fake = (-2 * scale * (br * l_rf + bn * kl_fake)).exp().mean()
        lossE = scale * (br * loss_rec + bk * kl_real) + 0.25 * (expelbo_rec + expelbo_fake)
        opt_e.zero_grad()
        lossE.backward()
        e_tensors = dict(fake=fake, real_mu=real_mu, real_logvar=real_logvar, z=z, rec=rec, loss_rec=loss_rec,
                         kl_real=kl_real, rec_mu=rec_mu, rec_logvar=rec_logvar, rec_rec=rec_rec, fake_mu=fake_mu,
                         fake_logvar=fake_logvar, rec_fake=rec_fake, kl_rec=kl_rec, kl_fake=kl_fake,
                         expelbo_rec=expelbo_rec, expelbo_fake=expelbo_fake, lossE=lossE)
        for k, v in e_tensors.items():
            if thin and k in IMAGE_KEYS:
                _thin(out, "E/" + k, v)
            else:
                out["E/" + k] = _np(v)
        for k, p in model.encoder.named_parameters():
            if _keep_grad(k):
                out["E/grad/encoder." + k] = _np(p.grad)
        opt_e.step()
        # ---- D step
        for p in model.encoder.parameters():
            p.requires_grad = False
        for p in model.decoder.parameters():
            p.requires_grad = True
        fake = model.sample(noise)
        rec = model.decoder(z.detach())
        loss_rec = T.calc_reconstruction_loss(real, rec, loss_type=loss_type, reduction="mean")
        rec_mu, rec_logvar = model.encode(rec)
        z_rec = T.reparameterize(rec_mu, rec_logvar)
        fake_mu, fake_logvar = model.encode(fake)
        z_fake = T.reparameterize(fake_mu, fake_logvar)
        if bootstrap:
            rec_rec = model.decode_target(z_rec)
            rec_fake = model.decode_target(z_fake)
            l_rr = T.calc_reconstruction_loss(rec, rec_rec, loss_type=loss_type, reduction="mean")
            l_fr = T.calc_reconstruction_loss(fake, rec_fake, loss_type=loss_type, reduction="mean")
        else:
            rec_rec = model.decode(z_rec.detach())
            rec_fake = model.decode(z_fake.detach())
            l_rr = T.calc_reconstruction_loss(rec.detach(), rec_rec, loss_type=loss_type, reduction="mean")
            l_fr = T.calc_reconstruction_loss(fake.detach(), rec_fake, loss_type=loss_type, reduction="mean")
        kl_rec = T.calc_kl(rec_logvar, rec_mu, reduce="mean")
        kl_fake = T.calc_kl(fake_logvar, fake_mu, reduce="mean")
        lossD = scale * (loss_rec * br + (kl_rec + kl_fake) * 0.5 * bk + gr * 0.5 * br * (l_rr + l_fr))
        opt_d.zero_grad()
        lossD.backward()
        d_tensors = dict(fake=fake, rec=rec, loss_rec=loss_rec, rec_mu=rec_mu, rec_logvar=rec_logvar,
                         fake_mu=fake_mu, fake_logvar=fake_logvar, rec_rec=rec_rec, rec_fake=rec_fake,
                         loss_rec_rec=l_rr, loss_fake_rec=l_fr, kl_rec=kl_rec, kl_fake=kl_fake, lossD=lossD)
        for k, v in d_tensors.items():
            if thin and k in IMAGE_KEYS:
                _thin(out, "D/" + k, v)
            else:
                out["D/" + k] = _np(v)
        for k, p in model.decoder.named_parameters():
            if _keep_grad(k):
                out["D/grad/decoder." + k] = _np(p.grad)
        opt_d.step()
    assert len(rr.draws) == 5, len(rr.draws)
    for i, d in enumerate(rr.draws):
        out["eps%d" % i] = d
    out.update(_sd(model, "final/"))
    np.savez_compressed(os.path.join(OUT, "step_%s.npz" % name), **out)
    print("step_%s.npz" % name, "lossE=%.6g lossD=%.6g" % (float(lossE), float(lossD)))


def make_loop(T, name, cdim, image_size, narrow_channels, zdim, B, n_batches, kwargs, bootstrap=False,
              dataset_key="cifar10"):
    """Run the reference's own training function, unmodified, on synthetic data; record everything."""
    g = torch.Generator().manual_seed(99)
    data = torch.rand(B * n_batches, cdim, image_size, image_size, generator=g)
    labels = torch.zeros(B * n_batches, dtype=torch.long)
    captured = {}
    RealVAE = T.SoftIntroVAE

    def narrow_factory(cdim=3, zdim=512, channels=None, image_size=256, **kw):
        m = RealVAE(cdim=cdim, zdim=zdim, channels=narrow_channels, image_size=image_size, **kw)
        captured["model"] = m
        captured["init"] = {k: v.clone() for k, v in m.state_dict().items()}
        return m

    batches = []
    RealLoader = torch.utils.data.DataLoader

    class RecLoader(RealLoader):
        def __iter__(self):
            for b in super().__iter__():
                batches.append(_np(b[0]))
                yield b

    ds_name = {"cifar10": "CIFAR10", "mnist": "MNIST", "svhn": "SVHN"}[dataset_key]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            # the training function's code object is executed UNMODIFIED, with three of its global names
            # rebound (model factory -> narrow widths, dataset -> synthetic tensors, DataLoader -> recorder);
            # the reference classes keep their own globals, so their super(...) calls still resolve.
            g2 = dict(T.__dict__)
            g2["SoftIntroVAE"] = narrow_factory
            g2[ds_name] = lambda **kw: torch.utils.data.TensorDataset(data, labels)
            g2["DataLoader"] = RecLoader
            fn = types.FunctionType(T.train_soft_intro_vae.__code__, g2, "train_soft_intro_vae",
                                    T.train_soft_intro_vae.__defaults__)
            with RandnRecorder() as rr:
                fn(dataset=dataset_key, z_dim=zdim, batch_size=B, num_workers=0,
                   device=torch.device("cpu"), **kwargs)
        finally:
            os.chdir(cwd)
    model = captured["model"]
    out = {"meta_cdim": cdim, "meta_zdim": zdim, "meta_channels": np.array(narrow_channels),
           "meta_image_size": image_size, "meta_bootstrap": int(bootstrap), "meta_n_iters": len(batches),
           "meta_batches_per_epoch": n_batches}
    for k, v in kwargs.items():
        if isinstance(v, (int, float)):
            out["hp_" + k] = v
    for k, v in captured["init"].items():
        out["init/" + k] = _np(v)
    for k, v in model.state_dict().items():
        out["final/" + k] = _np(v)
    for i, b in enumerate(batches):
        out["batch%d" % i] = b
    # per iteration: noise_batch (randn) then 5 randn_like draws; the epoch-end dump adds one more randn
    for i, d in enumerate(rr.draws):
        out["draw%d" % i] = d
    out["meta_n_draws"] = len(rr.draws)
    np.savez_compressed(os.path.join(OUT, "loop_%s.npz" % name), **out)
    print("loop_%s.npz" % name, "iters", len(batches), "draws", len(rr.draws))


def make_loop_2d(T2):
    batches = []
    RealDS = T2.ToyDataset

    class RecDS(RealDS):
        def next_batch(self, batch_size=64, device=None, sig=0.02):
            b = super().next_batch(batch_size=batch_size, device=device, sig=sig)
            batches.append(_np(b))
            return b

    captured = {}
    RealVAE = T2.SoftIntroVAESimple

    def factory(**kw):
        kw["num_hidden"] = 48  # narrow MLP (the reference hard-codes 256) to keep the fixture small
        m = RealVAE(**kw)
        captured["init"] = {k: v.clone() for k, v in m.state_dict().items()}
        return m

    n_iter, num_vae, B = 6, 2, 64
    kwargs = dict(z_dim=2, lr_e=2e-4, lr_d=2e-4, batch_size=B, n_iter=n_iter, num_vae=num_vae, save_interval=5000,
                  recon_loss_type="mse", beta_kl=0.3, beta_rec=0.2, beta_neg=0.9, test_iter=5000, seed=92, scale=1,
                  dataset="8Gaussians")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            # the end-of-training evaluation (density plots, grid ELBO, histogram KL/JSD) is not part of the
            # hot path: stub it out so the run only executes the training iterations
            g2 = dict(T2.__dict__)
            g2.update(ToyDataset=RecDS, SoftIntroVAESimple=factory,
                      plot_samples_density=lambda *a, **k: None, plot_vae_density=lambda *a, **k: None,
                      calculate_sample_kl=lambda *a, **k: 0.0, calculate_elbo_with_grid=lambda *a, **k: 0.0)
            fn = types.FunctionType(T2.train_soft_intro_vae_toy.__code__, g2, "train_soft_intro_vae_toy",
                                    T2.train_soft_intro_vae_toy.__defaults__)
            with mock.patch.object(T2.plt, "savefig", lambda *a, **k: None), RandnRecorder() as rr:
                model = fn(device=torch.device("cpu"), **kwargs)
        finally:
            os.chdir(cwd)
    out = {"meta_n_iter": n_iter, "meta_num_vae": num_vae, "meta_B": B, "meta_num_hidden": 48}
    for k, v in kwargs.items():
        if isinstance(v, (int, float)):
            out["hp_" + k] = v
    for k, v in captured["init"].items():
        out["init/" + k] = _np(v)
    for k, v in model.state_dict().items():
        out["final/" + k] = _np(v)
    # training batches are the first n_iter next_batch calls (later calls come from the plotting code)
    for i in range(n_iter):
        out["batch%d" % i] = batches[i]
    for i, d in enumerate(rr.draws):
        out["draw%d" % i] = d
    out["meta_n_draws"] = len(rr.draws)
    np.savez_compressed(os.path.join(OUT, "loop_2d.npz"), **out)
    print("loop_2d.npz draws", len(rr.draws), "batches", len(batches))


def make_cond(T, name, cdim, zdim, channels, image_size, B, cond_dim, seed=0, min_margin=5e-6):
    """The conditional model (train_soft_intro_vae.py:106-107,118-119,138-143,162-165,187-196): SoftIntroVAE(conditional=
    True) forward with o_cond, its sample(z, y_cond), and the gradients of a scalar of (mu, logvar, rec) for the two fc
    layers (the only layers the condition touches) + the first and last convs.

    Conditioning of the fixture: a LeakyReLU pre-activation within fp32 rounding of 0 flips its mask between any two fp32
    evaluations and moves EVERY upstream gradient by O(1/sqrt(#elements)) (with seed 13 the decoder's res_in_16 output
    holds one at 5e-7 of the tensor's scale: an MI355X evaluation flips it, the reference's CPU one does not, and all
    decoder / encoder gradients differ by 6e-3).  The seed is therefore advanced until no LeakyReLU input of any pass lies
    within `min_margin` of the kink relative to its tensor's largest value; seed and margin are stored."""
    while True:
        torch.manual_seed(seed)
        model = T.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size, conditional=True,
                               cond_dim=cond_dim)
        model.train()
        init = _sd(model, "init/")
        margins = []
        hooks = [m.register_forward_pre_hook(
            lambda mod, inp: margins.append(float(inp[0].detach().abs().min() / inp[0].detach().abs().max())))
            for m in model.modules() if isinstance(m, torch.nn.LeakyReLU)]
        g = torch.Generator().manual_seed(4321 + seed)
        real = torch.rand(B, cdim, image_size, image_size, generator=g)
        cond = torch.zeros(B, cond_dim)
        cond[torch.arange(B), torch.randint(0, cond_dim, (B,), generator=g)] = 1.0  # one-hot labels
        noise = torch.randn(B, zdim, generator=g)
        with RandnRecorder() as rr:
            mu, logvar, z, rec = model(real, o_cond=cond)
            fake = model.sample(noise, y_cond=cond)
            loss = (T.calc_reconstruction_loss(real, rec, loss_type="mse", reduction="mean")
                    + T.calc_kl(logvar, mu, reduce="mean") + fake.pow(2).mean())
            loss.backward()
        for h in hooks:
            h.remove()
        if min(margins) >= min_margin:
            break
        print("  seed %d: LeakyReLU kink margin %.2e < %.1e, next seed" % (seed, min(margins), min_margin))
        seed += 1
    out = {"meta_cdim": cdim, "meta_zdim": zdim, "meta_channels": np.array(channels), "meta_image_size": image_size,
           "meta_cond_dim": cond_dim, "meta_seed": seed, "meta_kink_margin": min(margins)}
    out.update(init)
    out["real"], out["cond"], out["noise"] = _np(real), _np(cond), _np(noise)
    assert len(rr.draws) == 1
    out["eps0"] = rr.draws[0]
    for k, v in dict(mu=mu, logvar=logvar, z=z, rec=rec, fake=fake, loss=loss).items():
        out["C/" + k] = _np(v)
    for k, p in model.named_parameters():
        if k in ("encoder.fc.weight", "encoder.fc.bias", "decoder.fc.0.weight", "decoder.fc.0.bias",
                 "encoder.main.0.weight", "decoder.main.predict.weight"):
            out["C/grad/" + k] = _np(p.grad)
    out.update(_sd(model, "final/"))  # (BatchNorm buffers after the two decoder passes / one encoder pass)
    np.savez_compressed(os.path.join(OUT, "%s.npz" % name), **out)
    print("%s.npz" % name, "seed %d margin %.2e loss=%.6g" % (seed, min(margins), float(loss)))


def main_round4():
    """fixtures added in round 4: full iterations with recon_loss_type l1 and bce (plain + bootstrap for l1), and the
    conditional model"""
    torch.set_num_threads(4)
    _stub_torchvision()
    T = _import_ref("soft_intro_vae", "train_soft_intro_vae")
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8, lr=2e-4)
    make_step(T, "l1_narrow", 3, 16, [8, 16, 32], 32, 4, hp, seed=11, loss_type="l1")
    make_step(T, "bce_narrow", 3, 16, [8, 16, 32], 32, 4, hp, seed=12, loss_type="bce")
    make_cond(T, "cond_narrow", 3, 16, [8, 16, 32], 32, 4, 10, seed=13)
    TB = _import_ref("soft_intro_vae_bootstrap", "train_soft_intro_vae_bootstrap")
    hpb = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1.0, lr=2e-4)
    make_step(TB, "bootstrap_l1_narrow", 3, 16, [8, 16, 32], 32, 4, hpb, bootstrap=True, seed=14, loss_type="l1")


def main_round2():
    """fixtures added in round 2 (the round-1 files are not regenerated): a vanilla-VAE iteration (plain and
    bootstrap), and full iterations on the reduced-width 128x128 (5-level) and 256x256 (6-level) topologies incl. the
    bootstrap variant on the 6-level one (SURVEY 8c list)"""
    torch.set_num_threads(4)
    _stub_torchvision()
    T = _import_ref("soft_intro_vae", "train_soft_intro_vae")
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8, lr=2e-4)
    hp2 = dict(beta_rec=0.5, beta_kl=1.0, beta_neg=1024.0, gamma_r=1e-8, lr=2e-4)
    make_vae_step(T, "vae_narrow", 3, 16, [8, 16, 32], 32, 4, hp, seed=8)
    make_step(T, "celeb128_narrow", 3, 32, [8, 16, 32, 64, 64], 128, 2, hp2, seed=4, thin=True)
    make_step(T, "celeb256_narrow", 3, 32, [8, 16, 32, 64, 64, 64], 256, 2, hp2, seed=5, thin=True)
    TB = _import_ref("soft_intro_vae_bootstrap", "train_soft_intro_vae_bootstrap")
    hpb = dict(beta_rec=0.5, beta_kl=1.0, beta_neg=1024.0, gamma_r=1.0, lr=2e-4)
    make_vae_step(TB, "vae_bootstrap_narrow", 3, 16, [8, 16, 32], 32, 4, dict(hp, gamma_r=1.0), bootstrap=True, seed=9)
    make_step(TB, "bootstrap256_narrow", 3, 32, [8, 16, 32, 64, 64, 64], 256, 2, hpb, bootstrap=True, seed=6, thin=True)


def main():
    torch.set_num_threads(4)
    _stub_torchvision()
    T = _import_ref("soft_intro_vae", "train_soft_intro_vae")
    make_helpers(T)
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8, lr=2e-4)
    make_step(T, "cifar_narrow", 3, 16, [8, 16, 32], 32, 4, hp)
    hp2 = dict(beta_rec=0.5, beta_kl=1.0, beta_neg=1024.0, gamma_r=1e-8, lr=2e-4)
    make_step(T, "deep64_narrow", 3, 12, [4, 8, 16, 16], 64, 2, hp2, seed=1)
    make_step(T, "mnist_narrow", 1, 8, [8, 16], 28, 6, hp, seed=2)
    make_loop(T, "cifar_narrow", 3, 32, [8, 16, 32], 16, 8, 3,
              dict(lr_e=2e-4, lr_d=2e-4, num_vae=0, beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=5,
                   test_iter=1000, save_interval=50, start_epoch=0, num_epochs=1))
    # epoch 0 = vanilla-VAE branch, epoch 1 = Soft-Intro branch (num_epochs=1 with num_vae=1 trips a latent
    # reference bug: `b_size` is unbound in the end-of-training dump, train_soft_intro_vae.py:679)
    make_loop(T, "vae_branch", 3, 32, [8, 16, 32], 16, 8, 2,
              dict(lr_e=2e-4, lr_d=2e-4, num_vae=1, beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=6,
                   test_iter=1000, save_interval=50, start_epoch=0, num_epochs=2))
    TB = _import_ref("soft_intro_vae_bootstrap", "train_soft_intro_vae_bootstrap")
    hpb = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1.0, lr=2e-4)
    make_step(TB, "bootstrap_narrow", 3, 16, [8, 16, 32], 32, 4, hpb, bootstrap=True, seed=3)
    make_loop(TB, "bootstrap_narrow", 3, 32, [8, 16, 32], 16, 8, 3,
              dict(lr_e=2e-4, lr_d=2e-4, num_vae=0, beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=7,
                   test_iter=1000, save_interval=50, start_epoch=0, copy_to_target_freq=1, num_epochs=2),
              bootstrap=True)
    T2 = _import_ref("soft_intro_vae_2d", "train_soft_intro_vae_2d")
    make_loop_2d(T2)


if __name__ == "__main__":
    if "--round2" in sys.argv:
        main_round2()
    elif "--round4" in sys.argv:
        main_round4()
    else:
        main()
        main_round2()
        main_round4()
