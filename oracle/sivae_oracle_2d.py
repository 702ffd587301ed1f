"""CPU oracle for the 2-D toy variant (config 1, "CPU plumbing") — TEST INFRASTRUCTURE ONLY.

Functional restatement of soft_intro_vae_2d/train_soft_intro_vae_2d.py: the MLP encoder/decoder
(EncoderSimple :402-421, DecoderSimple :424-444), the simple calc_kl (:290-308) and the
iteration of train_soft_intro_vae_toy (:517-660): vanilla-VAE branch :526-543, Soft-Intro branch
:549-642 (note the fake-before-rec encode order at :584-586 and dim_scale = 0.5 at :515).

Parity status: PINNED by tests/golden/loop_2d.npz (the reference's own training function, unmodified,
with every batch and Gaussian draw recorded) — see tests/test_oracle_golden.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .sivae_oracle import Adam, calc_reconstruction_loss, reparameterize  # noqa: F401

DIM_SCALE = 0.5


def layer_names(n_layers):
    return ["input"] + ["hidden_%d" % (i + 1) for i in range(n_layers)] + ["output"]


def init_params(x_dim=2, zdim=2, n_layers=3, num_hidden=256, seed=0):
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()

    def lin(name, fin, fout):
        b = 1.0 / math.sqrt(fin)
        P[name + ".weight"] = (torch.rand(fout, fin, generator=g) * 2 - 1) * b
        P[name + ".bias"] = (torch.rand(fout, generator=g) * 2 - 1) * b

    names = layer_names(n_layers)
    dims_e = [x_dim] + [num_hidden] * (n_layers + 1) + [2 * zdim]
    dims_d = [zdim] + [num_hidden] * (n_layers + 1) + [x_dim]
    for i, n in enumerate(names):
        lin("encoder.main." + n, dims_e[i], dims_e[i + 1])
    P["decoder.loggamma"] = torch.tensor(0.0)
    for i, n in enumerate(names):
        lin("decoder.main." + n, dims_d[i], dims_d[i + 1])
    return P


def _mlp(P, prefix, x, n_layers):
    names = layer_names(n_layers)
    h = x
    for n in names[:-1]:
        h = F.relu(F.linear(h, P[prefix + n + ".weight"], P[prefix + n + ".bias"]))
    return F.linear(h, P[prefix + names[-1] + ".weight"], P[prefix + names[-1] + ".bias"])


def encode(P, x, n_layers=3):
    y = _mlp(P, "encoder.main.", x, n_layers)
    z = y.shape[1] // 2
    return y[:, :z], y[:, z:]


def decode(P, z, n_layers=3):
    return _mlp(P, "decoder.main.", z.reshape(z.shape[0], -1), n_layers)


def calc_kl(logvar, mu, reduce="sum"):
    """:290-308 (non-outlier form)"""
    per = -0.5 * (1 + logvar - mu.pow(2) - logvar.exp()).sum(1)
    if reduce == "sum":
        return per.sum()
    if reduce == "mean":
        return per.mean()
    return per


def _grads(P, prefix, flag):
    for k in P:
        if k.startswith(prefix):
            P[k].requires_grad_(flag)
            if flag:
                P[k].grad = None


def vae_iteration(P, opt_e, opt_d, batch, eps0, hp, n_layers=3):
    _grads(P, "encoder.", True)
    _grads(P, "decoder.", True)
    mu, logvar = encode(P, batch, n_layers)
    z = reparameterize(mu, logvar, eps0)
    rec = decode(P, z, n_layers)
    loss_rec = calc_reconstruction_loss(batch, rec, hp.get("recon_loss_type", "mse"), "mean")
    loss_kl = calc_kl(logvar, mu, "mean")
    loss = hp["beta_rec"] * loss_rec + hp["beta_kl"] * loss_kl
    loss.backward()
    opt_e.step()
    opt_d.step()
    return dict(loss=loss, loss_rec=loss_rec, loss_kl=loss_kl)


def soft_intro_iteration(P, opt_e, opt_d, batch, noise, eps5, hp, n_layers=3):
    """eps5 in the reference's draw order: real (:570), fake (:584), rec (:586), rec (:621), fake (:624)."""
    br, bk, bn, gr = hp["beta_rec"], hp["beta_kl"], hp["beta_neg"], hp.get("gamma_r", 1e-8)
    lt = hp.get("recon_loss_type", "mse")
    # ---- E
    _grads(P, "encoder.", True)
    _grads(P, "decoder.", False)
    fake = decode(P, noise, n_layers)
    real_mu, real_logvar = encode(P, batch, n_layers)
    z = reparameterize(real_mu, real_logvar, eps5[0])
    rec = decode(P, z, n_layers)
    loss_rec = calc_reconstruction_loss(batch, rec, lt, "mean")
    kl_real = calc_kl(real_logvar, real_mu, "mean")
    fake_mu, fake_logvar = encode(P, fake.detach(), n_layers)
    rec_fake = decode(P, reparameterize(fake_mu, fake_logvar, eps5[1]), n_layers)
    rec_mu, rec_logvar = encode(P, rec.detach(), n_layers)
    rec_rec = decode(P, reparameterize(rec_mu, rec_logvar, eps5[2]), n_layers)
    fake_kl = calc_kl(fake_logvar, fake_mu, "none")
    rec_kl = calc_kl(rec_logvar, rec_mu, "none")
    l_fake = calc_reconstruction_loss(fake, rec_fake, lt, "none")
    l_rec = calc_reconstruction_loss(rec, rec_rec, lt, "none")
    e_fake = (-2 * DIM_SCALE * (br * l_fake + bn * fake_kl)).exp().mean()
    e_rec = (-2 * DIM_SCALE * (br * l_rec + bn * rec_kl)).exp().mean()
    lossE = DIM_SCALE * (bk * kl_real + br * loss_rec) + 0.25 * (e_fake + e_rec)
    lossE.backward()
    opt_e.step()
    # ---- D
    _grads(P, "encoder.", False)
    _grads(P, "decoder.", True)
    fake = decode(P, noise, n_layers)
    rec = decode(P, z.detach(), n_layers)
    loss_rec_d = calc_reconstruction_loss(batch, rec, lt, "mean")
    rec_mu, rec_logvar = encode(P, rec, n_layers)
    z_rec = reparameterize(rec_mu, rec_logvar, eps5[3])
    fake_mu, fake_logvar = encode(P, fake, n_layers)
    z_fake = reparameterize(fake_mu, fake_logvar, eps5[4])
    rec_rec = decode(P, z_rec.detach(), n_layers)
    rec_fake = decode(P, z_fake.detach(), n_layers)
    l_rr = calc_reconstruction_loss(rec.detach(), rec_rec, lt, "mean")
    l_rf = calc_reconstruction_loss(fake.detach(), rec_fake, lt, "mean")
    fkl = calc_kl(fake_logvar, fake_mu, "mean")
    rkl = calc_kl(rec_logvar, rec_mu, "mean")
    lossD = DIM_SCALE * (br * loss_rec_d + 0.5 * bk * (fkl + rkl) + gr * 0.5 * br * (l_rr + l_rf))
    lossD.backward()
    opt_d.step()
    return dict(lossE=lossE, lossD=lossD, loss_rec=loss_rec_d, kl_real=kl_real, expelbo_fake=e_fake,
                expelbo_rec=e_rec, kl_fake=fkl, kl_rec=rkl)
