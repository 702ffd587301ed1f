"""CPU oracle for the Soft-IntroVAE hot path — TEST INFRASTRUCTURE ONLY.

A from-scratch functional restatement (stock torch CPU ops, explicit parameter dictionaries, explicit
noise injection) of the reference's image model, loss helpers and one training iteration:

    ResidualBlock / Encoder / Decoder / SoftIntroVAE   soft_intro_vae/train_soft_intro_vae.py:38-223
    calc_kl / reparameterize / calc_reconstruction_loss :231-294
    E-step / D-step of train_soft_intro_vae             :542-624   (vanilla-VAE branch :512-540)
    bootstrap deltas                                    soft_intro_vae_bootstrap/train_soft_intro_vae_bootstrap.py:193-246,576-652
    Adam / MultiStepLR                                  :450-454

Parity status: PINNED — tests/test_oracle_golden.py checks every function here against golden vectors
captured from the imported reference (tests/golden/make_golden.py, run in the build container where
/root/reference is mounted).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (soft-intro-vae-pytorch_amd/) never does.

Parameters live in a flat dict keyed exactly like the reference's state_dict
("encoder.main.0.weight", "decoder.main.res_in_4.bn1.running_var", ...), so a reference checkpoint
IS an oracle parameter set.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

SLOPE = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------------
# architecture description (names follow train_soft_intro_vae.py:88-101,153-159)
# --------------------------------------------------------------------------------------------------
def encoder_plan(channels, image_size):
    """-> list of ("stem",) | ("res", name, cin, cout) | ("pool",) in execution order + final (C, h, w)"""
    plan = [("stem", "main", channels[0])]
    cc, sz = channels[0], image_size // 2
    for ch in channels[1:]:
        plan.append(("res", "main.res_in_%d" % sz, cc, ch))
        plan.append(("pool",))
        cc, sz = ch, sz // 2
    plan.append(("res", "main.res_in_%d" % sz, cc, cc))
    # spatial size after the stem pool and len(channels)-1 further pools, floor semantics
    s = image_size // 2
    for _ in channels[1:]:
        s = s // 2
    return plan, (cc, s, s)


def decoder_plan(channels, conv_input_size):
    cc, sz = channels[-1], 4
    plan = []
    for ch in channels[::-1]:
        plan.append(("res", "main.res_in_%d" % sz, cc, ch))
        plan.append(("up",))
        cc, sz = ch, sz * 2
    plan.append(("res", "main.res_in_%d" % sz, cc, cc))
    plan.append(("predict", "main.predict", cc))
    return plan


def init_params(cdim, zdim, channels, image_size, seed=0, bootstrap=False, dtype=torch.float32, cond_dim=0):
    """cond_dim > 0: the conditional variant (Encoder.fc takes num_fc_features + cond_dim inputs, Decoder.fc zdim +
    cond_dim: train_soft_intro_vae.py:106-107,138-143).
    Random parameters with the reference's shapes and torch's default init distributions
    (kaiming-uniform(a=sqrt 5) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases).
    Encoder BatchNorm buffers carry the side effect of the constructor's dummy forward
    (train_soft_intro_vae.py:102,111-114): running_var = 0.9, num_batches_tracked = 1."""
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)

    def conv(name, co, ci, k, bias=False):
        P[name + ".weight"] = uni((co, ci, k, k), ci * k * k)
        if bias:
            P[name + ".bias"] = uni((co,), ci * k * k)

    def bn(name, c, touched):
        P[name + ".weight"] = torch.ones(c, dtype=dtype)
        P[name + ".bias"] = torch.zeros(c, dtype=dtype)
        P[name + ".running_mean"] = torch.zeros(c, dtype=dtype)
        P[name + ".running_var"] = torch.full((c,), 0.9 if touched else 1.0, dtype=dtype)
        P[name + ".num_batches_tracked"] = torch.tensor(1 if touched else 0, dtype=torch.int64)

    def res(name, ci, co, touched):
        if ci != co:
            conv(name + ".conv_expand", co, ci, 1)
        conv(name + ".conv1", co, ci, 3)
        bn(name + ".bn1", co, touched)
        conv(name + ".conv2", co, co, 3)
        bn(name + ".bn2", co, touched)

    eplan, feat = encoder_plan(channels, image_size)
    for item in eplan:
        if item[0] == "stem":
            conv("encoder.main.0", item[2], cdim, 5)
            bn("encoder.main.1", item[2], True)
        elif item[0] == "res":
            res("encoder." + item[1], item[2], item[3], True)
    nfeat = feat[0] * feat[1] * feat[2]
    P["encoder.fc.weight"] = uni((2 * zdim, nfeat + cond_dim), nfeat + cond_dim)
    P["encoder.fc.bias"] = uni((2 * zdim,), nfeat + cond_dim)
    for dec in (["decoder", "target_decoder"] if bootstrap else ["decoder"]):
        P[dec + ".fc.0.weight"] = uni((nfeat, zdim + cond_dim), zdim + cond_dim)
        P[dec + ".fc.0.bias"] = uni((nfeat,), zdim + cond_dim)
        for item in decoder_plan(channels, feat):
            if item[0] == "res":
                res(dec + "." + item[1], item[2], item[3], False)
            elif item[0] == "predict":
                conv(dec + "." + item[1], cdim, item[2], 5, bias=True)
    return P


def trainable_keys(P, prefix):
    return [k for k in P if k.startswith(prefix) and not k.endswith(("running_mean", "running_var",
                                                                        "num_batches_tracked"))]


# --------------------------------------------------------------------------------------------------
# forward passes (training-mode BatchNorm, in-place buffer updates like nn.BatchNorm2d)
# --------------------------------------------------------------------------------------------------
def _bn(P, name, x, training=True):
    if training:
        P[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"],
                        P[name + ".bias"], training=training, momentum=BN_MOMENTUM, eps=BN_EPS)


def residual_block(P, name, x, training=True):
    """train_soft_intro_vae.py:65-75"""
    identity = F.conv2d(x, P[name + ".conv_expand.weight"]) if (name + ".conv_expand.weight") in P else x
    h = F.leaky_relu(_bn(P, name + ".bn1", F.conv2d(x, P[name + ".conv1.weight"], padding=1), training), SLOPE)
    c = _bn(P, name + ".bn2", F.conv2d(h, P[name + ".conv2.weight"], padding=1), training)
    return F.leaky_relu(c + identity, SLOPE)


def encode(P, x, channels, image_size, training=True, prefix="encoder.", o_cond=None):
    """Encoder.forward :116-122 -> (mu, logvar); o_cond [B, cond_dim] (conditional model, :118-119): concatenated to the
    flattened features in front of fc"""
    plan, _ = encoder_plan(channels, image_size)
    h = x
    for item in plan:
        if item[0] == "stem":
            h = F.conv2d(h, P[prefix + "main.0.weight"], padding=2)
            h = F.avg_pool2d(F.leaky_relu(_bn(P, prefix + "main.1", h, training), SLOPE), 2)
        elif item[0] == "res":
            h = residual_block(P, prefix + item[1], h, training)
        else:
            h = F.avg_pool2d(h, 2)
    h = h.reshape(h.shape[0], -1)
    if o_cond is not None:
        h = torch.cat([h, o_cond], dim=1)
    y = F.linear(h, P[prefix + "fc.weight"], P[prefix + "fc.bias"])
    zdim = y.shape[1] // 2
    return y[:, :zdim], y[:, zdim:]


def decode(P, z, channels, image_size, training=True, prefix="decoder.", y_cond=None):
    """Decoder.forward :161-169; y_cond [B, cond_dim] (conditional model, :162-165): concatenated to z in front of fc"""
    _, feat = encoder_plan(channels, image_size)
    z = z.reshape(z.shape[0], -1)
    if y_cond is not None:
        z = torch.cat([z, y_cond.reshape(y_cond.shape[0], -1)], dim=1)
    h = F.relu(F.linear(z, P[prefix + "fc.0.weight"], P[prefix + "fc.0.bias"]))
    h = h.reshape(z.shape[0], *feat)
    for item in decoder_plan(channels, feat):
        if item[0] == "res":
            h = residual_block(P, prefix + item[1], h, training)
        elif item[0] == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
        else:
            h = F.conv2d(h, P[prefix + item[1] + ".weight"], P[prefix + item[1] + ".bias"], padding=2)
    return h


# --------------------------------------------------------------------------------------------------
# loss helpers
# --------------------------------------------------------------------------------------------------
def reparameterize(mu, logvar, eps):
    """:254-265 with the Gaussian draw made explicit"""
    return mu + eps * torch.exp(0.5 * logvar)


def calc_kl(logvar, mu, mu_o=0.0, logvar_o=0.0, reduce="sum"):
    """:231-251"""
    mu_o = torch.as_tensor(mu_o, dtype=mu.dtype)
    logvar_o = torch.as_tensor(logvar_o, dtype=mu.dtype)
    per = -0.5 * (1 + logvar - logvar_o - logvar.exp() / logvar_o.exp() - (mu - mu_o) ** 2 / logvar_o.exp()).sum(1)
    if reduce == "sum":
        return per.sum()
    if reduce == "mean":
        return per.mean()
    return per


def calc_reconstruction_loss(x, recon_x, loss_type="mse", reduction="sum"):
    """:268-294 (note: 'mean' of mse is the batch mean of per-sample sums; l1/bce follow F.* reductions)"""
    if reduction not in ("sum", "mean", "none"):
        raise NotImplementedError
    r = recon_x.reshape(recon_x.shape[0], -1)
    t = x.reshape(x.shape[0], -1)
    if loss_type == "mse":
        per = ((r - t) ** 2).sum(1)
        if reduction == "sum":
            return per.sum()
        if reduction == "mean":
            return per.mean()
        return per
    if loss_type == "l1":
        el = (r - t).abs()
    elif loss_type == "bce":
        el = F.binary_cross_entropy(r, t, reduction="none")
    else:
        raise NotImplementedError
    if reduction == "sum":
        return el.sum()
    if reduction == "mean":
        return el.mean()
    return el


def _per_sample(v):
    while v.dim() > 1:
        v = v.sum(-1)
    return v


# --------------------------------------------------------------------------------------------------
# one Soft-IntroVAE iteration: gradients only (the optimizer is applied by the caller)
# --------------------------------------------------------------------------------------------------
def _set_requires_grad(P, prefix, flag):
    for k in trainable_keys(P, prefix):
        P[k].requires_grad_(flag)
        if flag:
            P[k].grad = None


def e_step(P, real, noise, eps, hp, channels, image_size, bootstrap=False):
    """Encoder update, :551-588. eps = (eps_real, eps_rec, eps_fake) in the reference's draw order
    (reparameterize(real) :560, inside model(rec) :567, inside model(fake) :568).
    Returns dict of tensors; encoder gradients are left in P[k].grad."""
    scale = 1.0 / (real.shape[1] * real.shape[2] * real.shape[3])
    br, bk, bn_, lt = hp["beta_rec"], hp["beta_kl"], hp["beta_neg"], hp.get("recon_loss_type", "mse")
    _set_requires_grad(P, "encoder.", True)
    _set_requires_grad(P, "decoder.", False)
    if bootstrap:
        _set_requires_grad(P, "target_decoder.", False)
    second = "target_decoder." if bootstrap else "decoder."

    fake = decode(P, noise, channels, image_size)
    real_mu, real_logvar = encode(P, real, channels, image_size)
    z = reparameterize(real_mu, real_logvar, eps[0])
    rec = decode(P, z, channels, image_size)
    loss_rec = calc_reconstruction_loss(real, rec, lt, "mean")
    kl_real = calc_kl(real_logvar, real_mu, reduce="mean")

    rec_mu, rec_logvar = encode(P, rec.detach(), channels, image_size)
    z_rec = reparameterize(rec_mu, rec_logvar, eps[1])
    rec_rec = decode(P, z_rec, channels, image_size, prefix=second)
    fake_mu, fake_logvar = encode(P, fake.detach(), channels, image_size)
    z_fake = reparameterize(fake_mu, fake_logvar, eps[2])
    rec_fake = decode(P, z_fake, channels, image_size, prefix=second)

    kl_rec = calc_kl(rec_logvar, rec_mu, reduce="none")
    kl_fake = calc_kl(fake_logvar, fake_mu, reduce="none")
    l_rec_rec = _per_sample(calc_reconstruction_loss(rec, rec_rec, lt, "none"))
    l_rec_fake = _per_sample(calc_reconstruction_loss(fake, rec_fake, lt, "none"))
    expelbo_rec = (-2 * scale * (br * l_rec_rec + bn_ * kl_rec)).exp().mean()
    expelbo_fake = (-2 * scale * (br * l_rec_fake + bn_ * kl_fake)).exp().mean()
    lossE = scale * (br * loss_rec + bk * kl_real) + 0.25 * (expelbo_rec + expelbo_fake)
    lossE.backward()
    return dict(fake=fake, real_mu=real_mu, real_logvar=real_logvar, z=z, rec=rec, loss_rec=loss_rec,
                kl_real=kl_real, rec_mu=rec_mu, rec_logvar=rec_logvar, rec_rec=rec_rec, fake_mu=fake_mu,
                fake_logvar=fake_logvar, rec_fake=rec_fake, kl_rec=kl_rec, kl_fake=kl_fake,
                expelbo_rec=expelbo_rec, expelbo_fake=expelbo_fake, lossE=lossE)


def d_step(P, real, noise, z, eps, hp, channels, image_size, bootstrap=False):
    """Decoder update, :591-623. z is the E-step latent (detached); eps = (eps_rec, eps_fake)
    (draw order :602, :605). Bootstrap: target decoder + un-detached paths, bootstrap file :623-652."""
    scale = 1.0 / (real.shape[1] * real.shape[2] * real.shape[3])
    br, bk, gr, lt = hp["beta_rec"], hp["beta_kl"], hp["gamma_r"], hp.get("recon_loss_type", "mse")
    _set_requires_grad(P, "encoder.", False)
    _set_requires_grad(P, "decoder.", True)
    if bootstrap:
        _set_requires_grad(P, "target_decoder.", False)

    fake = decode(P, noise, channels, image_size)
    rec = decode(P, z.detach(), channels, image_size)
    loss_rec = calc_reconstruction_loss(real, rec, lt, "mean")
    rec_mu, rec_logvar = encode(P, rec, channels, image_size)
    z_rec = reparameterize(rec_mu, rec_logvar, eps[0])
    fake_mu, fake_logvar = encode(P, fake, channels, image_size)
    z_fake = reparameterize(fake_mu, fake_logvar, eps[1])
    if bootstrap:
        rec_rec = decode(P, z_rec, channels, image_size, prefix="target_decoder.")
        rec_fake = decode(P, z_fake, channels, image_size, prefix="target_decoder.")
        l_rec_rec = calc_reconstruction_loss(rec, rec_rec, lt, "mean")
        l_fake_rec = calc_reconstruction_loss(fake, rec_fake, lt, "mean")
    else:
        rec_rec = decode(P, z_rec.detach(), channels, image_size)
        rec_fake = decode(P, z_fake.detach(), channels, image_size)
        l_rec_rec = calc_reconstruction_loss(rec.detach(), rec_rec, lt, "mean")
        l_fake_rec = calc_reconstruction_loss(fake.detach(), rec_fake, lt, "mean")
    kl_rec = calc_kl(rec_logvar, rec_mu, reduce="mean")
    kl_fake = calc_kl(fake_logvar, fake_mu, reduce="mean")
    lossD = scale * (loss_rec * br + (kl_rec + kl_fake) * 0.5 * bk + gr * 0.5 * br * (l_rec_rec + l_fake_rec))
    lossD.backward()
    return dict(fake=fake, rec=rec, loss_rec=loss_rec, rec_mu=rec_mu, rec_logvar=rec_logvar, fake_mu=fake_mu,
                fake_logvar=fake_logvar, rec_rec=rec_rec, rec_fake=rec_fake, loss_rec_rec=l_rec_rec,
                loss_fake_rec=l_fake_rec, kl_rec=kl_rec, kl_fake=kl_fake, lossD=lossD)


def vae_step(P, real, eps0, hp, channels, image_size, bootstrap=False):
    """Vanilla-VAE branch :512-533 (bootstrap decodes with the target decoder, bootstrap file :546)."""
    _set_requires_grad(P, "encoder.", True)
    _set_requires_grad(P, "decoder.", True)
    mu, logvar = encode(P, real, channels, image_size)
    z = reparameterize(mu, logvar, eps0)
    rec = decode(P, z, channels, image_size, prefix="target_decoder." if bootstrap else "decoder.")
    loss_rec = calc_reconstruction_loss(real, rec, hp.get("recon_loss_type", "mse"), "mean")
    loss_kl = calc_kl(logvar, mu, reduce="mean")
    loss = hp["beta_rec"] * loss_rec + hp["beta_kl"] * loss_kl
    loss.backward()
    return dict(mu=mu, logvar=logvar, z=z, rec=rec, loss_rec=loss_rec, loss_kl=loss_kl, loss=loss)


# --------------------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)
# --------------------------------------------------------------------------------------------------
class Adam:
    def __init__(self, P, keys, lr, betas=(0.9, 0.999), eps=1e-8):
        self.P, self.keys, self.lr, self.betas, self.eps = P, list(keys), lr, betas, eps
        self.t = 0  # number of step() calls
        self.steps = {k: 0 for k in self.keys}  # torch.optim.Adam keeps the step count PER PARAMETER ...
        self.m = {k: torch.zeros_like(P[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(P[k]) for k in self.keys}

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        for k in self.keys:
            g = self.P[k].grad
            if g is None:
                continue  # ... and skips (state untouched, count not advanced) parameters without a gradient
            self.steps[k] += 1
            bc1, bc2 = 1 - b1 ** self.steps[k], 1 - b2 ** self.steps[k]
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            self.P[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def train_iteration(P, opt_e, opt_d, real, noise, eps5, hp, channels, image_size, bootstrap=False):
    """One full Soft-IntroVAE iteration (:547-624): E-step, Adam(encoder), D-step, Adam(decoder).
    eps5 = the five reparameterisation draws in the reference's order."""
    e = e_step(P, real, noise, eps5[:3], hp, channels, image_size, bootstrap)
    opt_e.step()
    d = d_step(P, real, noise, e["z"], eps5[3:], hp, channels, image_size, bootstrap)
    opt_d.step()
    return e, d
