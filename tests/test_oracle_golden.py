"""T0/T1 — the CPU oracle (oracle/) against the golden vectors captured from the imported reference.

These tests pin the oracle: every helper, one full iteration with all intermediates and gradients, and
the reference's own (unmodified) training loops for several iterations. They run on CPU everywhere
(no reference import, fixtures only).
"""
import os

import numpy as np
import pytest
import torch

from oracle import sivae_oracle as O
from oracle import sivae_oracle_2d as O2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _t(a):
    return torch.from_numpy(np.array(a))


def _close(a, b, rtol, what):
    a = a.detach().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = torch.as_tensor(np.array(b)).double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert err <= rtol, "%s: rel err %.3e > %.1e" % (what, err, rtol)


def _close_fx(v, fx, key, rtol, what):
    """compare with a fixture entry stored in full, or thinned (every 4th pixel + per-image channel sums / sums of
    squares: the 128x128 / 256x256 fixtures, tests/golden/make_golden.py::_thin)"""
    if key in fx.files:
        return _close(v, fx[key], rtol, what)
    v = v.detach().double()
    _close(v[:, :, ::4, ::4], fx[key + "@thin"], rtol, what + " (thin)")
    _close(v.sum((2, 3)), fx[key + "@sum"], rtol, what + " (sum)")
    _close((v * v).sum((2, 3)), fx[key + "@sumsq"], rtol, what + " (sumsq)")


def _params_from(fx, prefix="init/"):
    P = {}
    for k in fx.files:
        if k.startswith(prefix):
            P[k[len(prefix):]] = _t(fx[k]).clone()
    return P


def _meta(fx):
    return (int(fx["meta_cdim"]), int(fx["meta_zdim"]), [int(c) for c in fx["meta_channels"]],
            int(fx["meta_image_size"]), bool(int(fx["meta_bootstrap"])))


# ---------------------------------------------------------------------------------------------- helpers
def test_helpers_match_reference():
    fx = _load("helpers.npz")
    mu, lv = _t(fx["mu"]), _t(fx["logvar"])
    for red in ("sum", "mean", "none"):
        _close(O.calc_kl(lv, mu, reduce=red), fx["kl_%s" % red], 1e-6, "kl " + red)
        _close(O.calc_kl(lv, mu, mu_o=0.3, logvar_o=-0.7, reduce=red), fx["kl_o_%s" % red], 1e-6, "kl_o " + red)
    x, r = _t(fx["x"]), _t(fx["recon"])
    for lt in ("mse", "l1", "bce"):
        for red in ("sum", "mean", "none"):
            _close(O.calc_reconstruction_loss(x, r, lt, red), fx["rec_%s_%s" % (lt, red)], 1e-6,
                   "recon %s %s" % (lt, red))
    _close(O.reparameterize(mu, lv, _t(fx["reparam_eps"])), fx["reparam_z"], 1e-7, "reparameterize")
    with pytest.raises(NotImplementedError):
        O.calc_reconstruction_loss(x, r, "mse", "bogus")
    with pytest.raises(NotImplementedError):
        O.calc_reconstruction_loss(x, r, "huber", "sum")


# ---------------------------------------------------------------------------------------------- structure
@pytest.mark.parametrize("name", ["step_cifar_narrow", "step_deep64_narrow", "step_mnist_narrow",
                                  "step_bootstrap_narrow"])
def test_init_params_structure(name):
    """oracle.init_params reproduces the reference state_dict: key set, shapes, and the encoder BatchNorm
    buffers mutated by the constructor's dummy forward (running_var 0.9, num_batches_tracked 1)."""
    fx = _load(name + ".npz")
    cdim, zdim, channels, image_size, boot = _meta(fx)
    ref = _params_from(fx)
    mine = O.init_params(cdim, zdim, channels, image_size, seed=0, bootstrap=boot)
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")) or ".bn" in k or "main.1." in k:
            _close(mine[k].double(), ref[k].double(), 1e-7, k)
    # same init distribution: weights bounded by 1/sqrt(fan_in)
    for k in ref:
        if k.endswith("conv1.weight"):
            bound = 1.0 / np.sqrt(ref[k].shape[1] * 9)
            assert float(ref[k].abs().max()) <= bound * (1 + 1e-6)
            assert float(mine[k].abs().max()) <= bound * (1 + 1e-6)


# ---------------------------------------------------------------------------------------------- one iteration
@pytest.mark.parametrize("name", ["step_cifar_narrow", "step_deep64_narrow", "step_mnist_narrow",
                                  "step_bootstrap_narrow", "step_celeb128_narrow", "step_celeb256_narrow",
                                  "step_bootstrap256_narrow", "step_l1_narrow", "step_bce_narrow",
                                  "step_bootstrap_l1_narrow"])
def test_one_iteration_matches_reference(name):
    """(round 4: step_l1 / step_bce / step_bootstrap_l1 run the iteration with recon_loss_type l1 / bce — reference
    :288-291 and the per-sample `while len(shape) > 1: sum(-1)` consumers :574-578)"""
    fx = _load(name + ".npz")
    cdim, zdim, channels, image_size, boot = _meta(fx)
    P = _params_from(fx)
    hp = dict(beta_rec=float(fx["hp_beta_rec"]), beta_kl=float(fx["hp_beta_kl"]), beta_neg=float(fx["hp_beta_neg"]),
              gamma_r=float(fx["hp_gamma_r"]))
    if "meta_recon_loss_type" in fx.files:
        hp["recon_loss_type"] = str(fx["meta_recon_loss_type"])
    lr = float(fx["hp_lr"])
    real, noise = _t(fx["real"]), _t(fx["noise"])
    eps = [_t(fx["eps%d" % i]) for i in range(5)]
    opt_e = O.Adam(P, O.trainable_keys(P, "encoder."), lr)
    opt_d = O.Adam(P, O.trainable_keys(P, "decoder."), lr)
    e = O.e_step(P, real, noise, eps[:3], hp, channels, image_size, boot)
    for k, v in e.items():
        _close_fx(v, fx, "E/" + k, 2e-5, "%s E/%s" % (name, k))
    ng = 0
    for k in fx.files:
        if k.startswith("E/grad/"):
            _close(P[k[len("E/grad/"):]].grad, fx[k], 2e-3, k)
            ng += 1
    assert ng >= 4
    opt_e.step()
    if "E/rec@thin" in fx.files:
        # deep nets: start the D-step from the REFERENCE's post-E-step encoder (Adam's sign-like first step turns fp32
        # summation-order noise in ~1e-8 gradients into O(lr) weight differences that depend on the thread count)
        for k in O.trainable_keys(P, "encoder."):
            P[k].data.copy_(_t(fx["final/" + k]))
    d = O.d_step(P, real, noise, e["z"], eps[3:], hp, channels, image_size, boot)
    for k, v in d.items():
        _close_fx(v, fx, "D/" + k, 2e-5, "%s D/%s" % (name, k))
    for k in fx.files:
        if k.startswith("D/grad/"):
            _close(P[k[len("D/grad/"):]].grad, fx[k], 2e-3, k)
    opt_d.step()
    # (bce fixture: the rescaled predict layer makes the decoder gradients ~1e-9, inside Adam's eps = 1e-8 regime where the
    # update is no longer scale-free — post-step weights are judged by drift in units of lr like the deep nets)
    thin = ("E/rec@thin" in fx.files) or hp.get("recon_loss_type") == "bce"
    for k in fx.files:
        if k.startswith("final/") and (not thin or k.endswith(("running_mean", "running_var", "num_batches_tracked"))):
            _close(P[k[len("final/"):]].double(), fx[k], 1e-4, k)
    if thin:
        # deep (5- / 6-level) nets: the first Adam step is sign-like on gradients of ~1e-8, so a few weights move by
        # up to ~lr between any two fp32 evaluations -> post-step weights in units of lr (weight_drift below)
        dmax, dmed, dfrac = weight_drift(P, fx, lr)
        assert dmed <= 0.1 and dfrac <= 0.01, "weight drift in lr units: max %.3f median %.3e frac>lr %.4f" % (
            dmax, dmed, dfrac)



def test_conditional_model_matches_reference():
    """the conditional branch of Encoder / Decoder / SoftIntroVAE (:106-107,118-119,138-143,162-165,187-196): o_cond is
    concatenated to the flattened features in front of Encoder.fc, y_cond to z in front of Decoder.fc"""
    fx = _load("cond_narrow.npz")
    cdim, zdim, image_size = int(fx["meta_cdim"]), int(fx["meta_zdim"]), int(fx["meta_image_size"])
    channels, cond_dim = [int(c) for c in fx["meta_channels"]], int(fx["meta_cond_dim"])
    P = _params_from(fx)
    mine = O.init_params(cdim, zdim, channels, image_size, seed=0, cond_dim=cond_dim)
    assert list(mine.keys()) == list(P.keys())
    assert all(tuple(mine[k].shape) == tuple(P[k].shape) for k in P)
    for k in O.trainable_keys(P, ""):
        P[k].requires_grad_(True)
    real, cond, noise = _t(fx["real"]), _t(fx["cond"]), _t(fx["noise"])
    mu, logvar = O.encode(P, real, channels, image_size, o_cond=cond)
    z = O.reparameterize(mu, logvar, _t(fx["eps0"]))
    rec = O.decode(P, z, channels, image_size, y_cond=cond)
    fake = O.decode(P, noise, channels, image_size, y_cond=cond)
    loss = (O.calc_reconstruction_loss(real, rec, "mse", "mean") + O.calc_kl(logvar, mu, reduce="mean")
            + fake.pow(2).mean())
    loss.backward()
    for k, v in dict(mu=mu, logvar=logvar, z=z, rec=rec, fake=fake, loss=loss).items():
        _close(v, fx["C/" + k], 2e-5, "cond " + k)
    ng = 0
    for k in fx.files:
        if k.startswith("C/grad/"):
            _close(P[k[len("C/grad/"):]].grad, fx[k], 2e-3, k)
            ng += 1
    assert ng == 6
    for k in fx.files:
        if k.startswith("final/") and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            _close(P[k[len("final/"):]].double(), fx[k], 1e-4, k)


@pytest.mark.parametrize("name", ["step_vae_narrow", "step_vae_bootstrap_narrow"])
def test_vanilla_vae_iteration_matches_reference(name):
    """the `epoch < num_vae` branch (:512-533; bootstrap: decodes through the target decoder, so the decoder gets no
    gradient and torch.optim.Adam leaves it alone): every output, every gradient, post-Adam weights"""
    fx = _load(name + ".npz")
    cdim, zdim, channels, image_size, boot = _meta(fx)
    P = _params_from(fx)
    hp = dict(beta_rec=float(fx["hp_beta_rec"]), beta_kl=float(fx["hp_beta_kl"]))
    lr = float(fx["hp_lr"])
    opt_e = O.Adam(P, O.trainable_keys(P, "encoder."), lr)
    opt_d = O.Adam(P, O.trainable_keys(P, "decoder."), lr)
    v = O.vae_step(P, _t(fx["real"]), _t(fx["eps0"]), hp, channels, image_size, boot)
    for mine, theirs in (("mu", "real_mu"), ("logvar", "real_logvar"), ("z", "z"), ("rec", "rec"),
                         ("loss_rec", "loss_rec"), ("loss_kl", "loss_kl"), ("loss", "loss")):
        _close(v[mine], fx["V/" + theirs], 2e-5, "%s V/%s" % (name, theirs))
    grads = [k for k in fx.files if k.startswith("V/grad/")]
    assert any(k.startswith("V/grad/encoder.") for k in grads)
    assert boot != any(k.startswith("V/grad/decoder.") for k in grads)  # bootstrap: no decoder gradients at all
    for k in grads:
        _close(P[k[len("V/grad/"):]].grad, fx[k], 2e-3, k)
    if boot:
        for k in O.trainable_keys(P, "decoder."):
            assert P[k].grad is None, k
    opt_e.step()
    opt_d.step()
    for k in fx.files:
        if k.startswith("final/"):
            _close(P[k[len("final/"):]].double(), fx[k], 1e-4, k)


# ---------------------------------------------------------------------------------------------- whole loops
def run_image_loop(fx, step_fn=None):
    """Replays the reference train loop on the recorded batches / draws with the oracle.
    Returns the parameter dict after the last iteration (and epoch-end side effects)."""
    cdim, zdim, channels, image_size, boot = _meta(fx)
    P = _params_from(fx)
    hp = dict(beta_rec=float(fx["hp_beta_rec"]), beta_kl=float(fx["hp_beta_kl"]), beta_neg=float(fx["hp_beta_neg"]),
              gamma_r=1.0 if boot else 1e-8)
    lr = float(fx["hp_lr_e"])
    num_vae, test_iter = int(fx["hp_num_vae"]), int(fx["hp_test_iter"])
    n_iters, per_epoch = int(fx["meta_n_iters"]), int(fx["meta_batches_per_epoch"])
    opt_e = O.Adam(P, O.trainable_keys(P, "encoder."), lr)
    opt_d = O.Adam(P, O.trainable_keys(P, "decoder."), lr)
    draws = [_t(fx["draw%d" % i]) for i in range(int(fx["meta_n_draws"]))]
    di = 0
    det_prefix = "target_decoder." if boot else "decoder."
    for it in range(n_iters):
        epoch = it // per_epoch
        real = _t(fx["batch%d" % it])
        if epoch < num_vae:
            O.vae_step(P, real, draws[di], hp, channels, image_size, boot)
            di += 1
            opt_e.step()
            opt_d.step()
        else:
            noise = draws[di]
            O.train_iteration(P, opt_e, opt_d, real, noise, draws[di + 1:di + 6], hp, channels, image_size, boot)
            di += 6
            if it % test_iter == 0:
                # train_soft_intro_vae.py:641-642 — a deterministic forward IN TRAIN MODE (BatchNorm side effects)
                with torch.no_grad():
                    mu, _ = O.encode(P, real, channels, image_size)
                    O.decode(P, mu, channels, image_size, prefix=det_prefix)
        if (it + 1) % per_epoch == 0 and boot:
            # bootstrap file :680-682 — copy decoder -> target_decoder (parameters and buffers)
            for k in list(P.keys()):
                if k.startswith("decoder."):
                    P["target_" + k] = P[k].detach().clone()
    # end of training, :676-680: deterministic forward + one sample, in train mode, under no_grad
    with torch.no_grad():
        mu, _ = O.encode(P, real, channels, image_size)
        O.decode(P, mu, channels, image_size, prefix=det_prefix)
        O.decode(P, draws[di], channels, image_size)
        di += 1
    assert di == len(draws)
    return P


def weight_drift(P, fx, lr):
    """|w - w_ref| / lr over all trainable tensors -> (max, median, fraction above one lr).

    Adam turns fp32 rounding noise in tiny gradients into O(lr) differences on a few elements (its
    first steps are sign-like: lr*g/(|g|+eps)), so post-step weights are compared in units of lr:
    a semantic error (wrong detach, wrong schedule) moves EVERY element by ~lr per step."""
    ds = []
    for k in fx.files:
        if k.startswith("final/") and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            ds.append((P[k[len("final/"):]].detach().double().numpy() - fx[k].astype(np.float64)).ravel())
    d = np.abs(np.concatenate(ds)) / lr
    return float(d.max()), float(np.median(d)), float((d > 1.0).mean())


@pytest.mark.parametrize("name", ["loop_cifar_narrow", "loop_vae_branch", "loop_bootstrap_narrow"])
def test_reference_training_loop_is_reproduced(name):
    fx = _load(name + ".npz")
    nthreads = torch.get_num_threads()
    torch.set_num_threads(4)  # the fixtures were generated with 4 threads: same partitioning -> bit-identical
    try:
        P = run_image_loop(fx)
    finally:
        torch.set_num_threads(nthreads)
    dmax, dmed, dfrac = weight_drift(P, fx, float(fx["hp_lr_e"]))
    # bit-exact with matching thread partitioning (dmax == 0 here); on other hosts allow fp32-noise drift
    assert dmed <= 0.1 and dfrac <= 0.01, "weight drift in lr units: max %.3f median %.3e frac>lr %.4f" % (
        dmax, dmed, dfrac)
    # BatchNorm buffers are not touched by Adam: tight
    checked = 0
    for k in fx.files:
        if k.startswith("final/") and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            _close(P[k[len("final/"):]].double(), fx[k], 2e-4, "%s %s" % (name, k))
            checked += 1
    assert checked > 20


def test_reference_2d_loop_is_reproduced():
    fx = _load("loop_2d.npz")
    P = _params_from(fx)
    hp = dict(beta_rec=float(fx["hp_beta_rec"]), beta_kl=float(fx["hp_beta_kl"]), beta_neg=float(fx["hp_beta_neg"]))
    lr = float(fx["hp_lr_e"])
    opt_e = O.Adam(P, [k for k in P if k.startswith("encoder.")], lr)
    opt_d = O.Adam(P, [k for k in P if k.startswith("decoder.")], lr)
    draws = [_t(fx["draw%d" % i]) for i in range(int(fx["meta_n_draws"]))]
    di = 0
    for it in range(int(fx["meta_n_iter"])):
        batch = _t(fx["batch%d" % it])
        if it < int(fx["meta_num_vae"]):
            O2.vae_iteration(P, opt_e, opt_d, batch, draws[di], hp)
            di += 1
        else:
            O2.soft_intro_iteration(P, opt_e, opt_d, batch, draws[di], draws[di + 1:di + 6], hp)
            di += 6
    assert di == len(draws) - 1  # the last draw is the plotting noise (:666)
    for k in fx.files:
        if k.startswith("final/"):
            _close(P[k[len("final/"):]].double(), fx[k], 1e-5, "2d " + k)
