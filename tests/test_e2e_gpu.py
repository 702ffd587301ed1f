"""T3 — the HIP product path end to end on the MI355X, against (a) the golden vectors captured from the
imported reference and (b) the CPU oracle run live on seeded inputs at the reference's real widths.

North-star tolerance: encoder mu/logvar, reconstructions and loss values within 1e-4 relative (fp32).
Gradients / post-Adam weights are fp32-noise amplified (see tests/test_oracle_golden.py::weight_drift) and
are checked with the drift criterion in units of the learning rate.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _rel(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.array(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.array(b)).double()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _build(fx, device):
    """product model + engine initialised from a fixture's `init/` state_dict"""
    import train_soft_intro_vae as T
    import train_soft_intro_vae_bootstrap as TB
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    cdim, zdim = int(fx["meta_cdim"]), int(fx["meta_zdim"])
    channels, image_size = [int(c) for c in fx["meta_channels"]], int(fx["meta_image_size"])
    boot = bool(int(fx["meta_bootstrap"]))
    model = (TB if boot else T).SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    sd = {k[len("init/"):]: torch.from_numpy(np.array(fx[k])) for k in fx.files if k.startswith("init/")}
    model.load_state_dict(sd, strict=True)
    model = model.to(device).train()
    return model, boot


def _engine(model, boot, hp, lr):
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    oe, od = FlatAdam(model.encoder.parameters(), lr=lr), FlatAdam(model.decoder.parameters(), lr=lr)
    return SoftIntroEngine(model, oe, od, beta_kl=hp["beta_kl"], beta_rec=hp["beta_rec"], beta_neg=hp["beta_neg"],
                           gamma_r=hp["gamma_r"], bootstrap=boot), oe, od


@pytest.mark.parametrize("name", ["step_cifar_narrow", "step_deep64_narrow", "step_mnist_narrow",
                                  "step_bootstrap_narrow"])
def test_iteration_matches_reference_fixture(name):
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    model, boot = _build(fx, dev)
    hp = {k: float(fx["hp_" + k]) for k in ("beta_rec", "beta_kl", "beta_neg", "gamma_r")}
    eng, oe, od = _engine(model, boot, hp, float(fx["hp_lr"]))
    grads = {}
    for tag, opt, net in (("E", oe, model.encoder), ("D", od, model.decoder)):
        orig = opt.step

        def step(grad_scale=1.0, _orig=orig, _net=net, _tag=tag):
            grads[_tag] = {k: p.grad.detach().clone() for k, p in _net.named_parameters()}
            _orig(grad_scale)
        opt.step = step
    real = torch.from_numpy(fx["real"]).to(dev)
    noise = torch.from_numpy(fx["noise"]).to(dev)
    eps = [torch.from_numpy(fx["eps%d" % i]).to(dev) for i in range(5)]
    out = eng.soft_intro_step(real, noise=noise, eps=eps, keep=True)
    torch.cuda.synchronize()
    bad = []
    for part in ("E", "D"):
        for k, v in out[part].items():
            err = _rel(v, fx["%s/%s" % (part, k)])
            if err > TOL:
                bad.append(("%s/%s" % (part, k), err))
    assert not bad, "forward/loss parity vs reference: %s" % bad
    gbad = []
    for k in fx.files:
        for part, pre in (("E", "E/grad/encoder."), ("D", "D/grad/decoder.")):
            if k.startswith(pre):
                err = _rel(grads[part][k[len(pre):]], fx[k])
                if err > 5e-3:
                    gbad.append((k, err))
    assert not gbad, "gradient parity vs reference: %s" % gbad
    # BatchNorm buffers after the iteration: exact semantics (5 encoder / 8 decoder updates), tight tolerance
    sd = model.state_dict()
    for k in fx.files:
        if k.startswith("final/") and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert _rel(sd[k[len("final/"):]], fx[k]) <= 2e-4, k
    # weights after one Adam step: drift in lr units
    lr = float(fx["hp_lr"])
    ds = []
    for k in fx.files:
        if k.startswith("final/") and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            ds.append((sd[k[len("final/"):]].double().cpu().numpy() - fx[k].astype(np.float64)).ravel())
    d = np.abs(np.concatenate(ds)) / lr
    assert np.median(d) <= 0.1 and (d > 1.0).mean() <= 0.01, (float(d.max()), float(np.median(d)))


@pytest.mark.parametrize("name", ["loop_cifar_narrow", "loop_vae_branch", "loop_bootstrap_narrow"])
def test_reference_training_loop_on_hip(name):
    """replays the reference's own training run (recorded batches + Gaussian draws) on the HIP engine"""
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    model, boot = _build(fx, dev)
    hp = dict(beta_rec=float(fx["hp_beta_rec"]), beta_kl=float(fx["hp_beta_kl"]), beta_neg=float(fx["hp_beta_neg"]),
              gamma_r=1.0 if boot else 1e-8)
    lr = float(fx["hp_lr_e"])
    eng, oe, od = _engine(model, boot, hp, lr)
    num_vae, test_iter = int(fx["hp_num_vae"]), int(fx["hp_test_iter"])
    n_iters, per_epoch = int(fx["meta_n_iters"]), int(fx["meta_batches_per_epoch"])
    draws = [torch.from_numpy(fx["draw%d" % i]).to(dev) for i in range(int(fx["meta_n_draws"]))]
    di = 0
    for it in range(n_iters):
        real = torch.from_numpy(fx["batch%d" % it]).to(dev)
        if it // per_epoch < num_vae:
            eng.vae_step(real, eps=draws[di])
            di += 1
        else:
            eng.soft_intro_step(real, noise=draws[di], eps=draws[di + 1:di + 6])
            di += 6
            if it % test_iter == 0:
                with torch.no_grad():
                    model(real, deterministic=True)
        if (it + 1) % per_epoch == 0 and boot:
            model.target_decoder.load_state_dict(model.decoder.state_dict())
    with torch.no_grad():
        model(real, deterministic=True)
        model.sample(draws[di])
    torch.cuda.synchronize()
    sd = model.state_dict()
    ds = []
    for k in fx.files:
        if not k.startswith("final/"):
            continue
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert _rel(sd[k[len("final/"):]], fx[k]) <= 5e-4, k
        else:
            ds.append((sd[k[len("final/"):]].double().cpu().numpy() - fx[k].astype(np.float64)).ravel())
    d = np.abs(np.concatenate(ds)) / lr
    assert np.median(d) <= 0.1 and (d > 1.0).mean() <= 0.01, (float(d.max()), float(np.median(d)), name)


# ---------------------------------------------------------------------------------------------- vs live oracle
def _oracle_vs_hip(cdim, zdim, channels, image_size, B, hp, boot=False, seed=0):
    from oracle import sivae_oracle as O
    import train_soft_intro_vae as T
    import train_soft_intro_vae_bootstrap as TB
    dev = torch.device("cuda:0")
    P = O.init_params(cdim, zdim, channels, image_size, seed=seed, bootstrap=boot)
    model = (TB if boot else T).SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    model = model.to(dev).train()
    eng, oe, od = _engine(model, boot, hp, 2e-4)
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, cdim, image_size, image_size, generator=g)
    noise = torch.randn(B, zdim, generator=g)
    eps = [torch.randn(B, zdim, generator=g) for _ in range(5)]
    out = eng.soft_intro_step(real.to(dev), noise=noise.to(dev), eps=[e.to(dev) for e in eps], keep=True)
    opt_e = O.Adam(P, O.trainable_keys(P, "encoder."), 2e-4)
    opt_d = O.Adam(P, O.trainable_keys(P, "decoder."), 2e-4)
    e, d = O.train_iteration(P, opt_e, opt_d, real, noise, eps, hp, channels, image_size, boot)
    bad = []
    for part, ref in (("E", e), ("D", d)):
        for k, v in ref.items():
            err = _rel(out[part][k], v)
            if err > TOL:
                bad.append(("%s/%s" % (part, k), err))
    return bad


def test_cifar_full_width_vs_oracle():
    """config 2 network (32x32, channels [64,128,256], z 128) at B = 16, beta_neg 256"""
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8)
    assert not _oracle_vs_hip(3, 128, [64, 128, 256], 32, 16, hp)


def test_celeb128_topology_vs_oracle():
    """config 3 topology (128x128, 5 levels) at reduced width, B = 4, CelebA betas"""
    hp = dict(beta_rec=0.5, beta_kl=1.0, beta_neg=1024.0, gamma_r=1e-8)
    assert not _oracle_vs_hip(3, 64, [16, 32, 64, 128, 128], 128, 4, hp, seed=1)


def test_celeb256_full_config_vs_oracle():
    """config 4 network exactly (256x256, [64,128,256,512,512,512], z 512) at B = 2"""
    hp = dict(beta_rec=0.5, beta_kl=1.0, beta_neg=1024.0, gamma_r=1e-8)
    assert not _oracle_vs_hip(3, 512, [64, 128, 256, 512, 512, 512], 256, 2, hp, seed=2)


def test_bootstrap_full_width_vs_oracle():
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1.0)
    assert not _oracle_vs_hip(3, 128, [64, 128, 256], 32, 8, hp, boot=True, seed=3)


def test_size_independent_properties_at_full_batch_shapes():
    """Properties that need no oracle, at the headline layer shapes: dgrad is the adjoint of fwd
    (<conv(x), y> == <x, conv^T(y)>), wgrad is linear in dy, BatchNorm output has zero mean / unit variance."""
    from sivae_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for (B, Ci, Co, H, ks) in [(16, 64, 64, 256, 3), (16, 256, 512, 32, 3), (16, 512, 512, 4, 3), (16, 64, 3, 256, 5)]:
        x = torch.randn(B, Ci, H, H, generator=g).to(dev)
        y = torch.randn(B, Co, H, H, generator=g).to(dev)
        w = (torch.randn(Co, Ci, ks, ks, generator=g) / (Ci * ks * ks) ** 0.5).to(dev)
        fx = ops.conv2d_fwd(x, ops.pack_weight(w, 0), Co, ks)
        bty = ops.conv2d_fwd(y, ops.pack_weight(w, 1), Ci, ks)
        lhs = float((fx.double() * y.double()).sum())
        rhs = float((x.double() * bty.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * (abs(lhs) + abs(rhs) + 1.0), (B, Ci, Co, H, ks, lhs, rhs)
        dw = ops.conv2d_wgrad(x, y, ks)
        wsum = float((dw.double() * w.double()).sum())
        assert abs(wsum - lhs) <= 1e-5 * (abs(lhs) + 1.0), (wsum, lhs)
        fx2, part = ops.conv2d_fwd(x, ops.pack_weight(w, 0), Co, ks, want_stats=True)
        mean, invstd = ops.bn_stats_from_conv(part, B, Co, H * H)
        ones = torch.ones(Co, device=dev)
        out = ops.bn_apply_act(fx2, None, mean, invstd, ones, torch.zeros(Co, device=dev), 1.0)
        assert float(out.mean((0, 2, 3)).abs().max()) < 1e-4
        assert float((out.var((0, 2, 3), unbiased=False) - 1).abs().max()) < 1e-3
