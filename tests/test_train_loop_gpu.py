"""GPU: the drop-in entry points themselves (train_soft_intro_vae / _bootstrap / _toy) and the 2-D variant
against the recorded run of the reference's own 2-D training function."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_train_soft_intro_vae_entry_point_runs(tmp_path, monkeypatch):
    import train_soft_intro_vae as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SIVAE_SYNTHETIC_IMAGES", "48")
    model = T.train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=32, batch_size=16, num_workers=0, num_epochs=2,
                                   num_vae=1, beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=3, test_iter=2,
                                   save_interval=1, device=torch.device("cuda:0"))
    sd = model.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
    # constructor probe (1) + epoch 0: 3 vanilla-VAE iterations (1 encoder pass each) + epoch 1: 3 Soft-Intro
    # iterations (5 encoder passes each) + the deterministic dump at cur_iter 4 (test_iter=2) + the final dump
    assert int(sd["encoder.main.1.num_batches_tracked"]) == 1 + 3 + 15 + 1 + 1
    ckpts = os.listdir(tmp_path / "saves")
    assert any(c.endswith(".pth") for c in ckpts)
    # checkpoint round trip in the reference's format {"epoch", "model"}
    m2 = T.SoftIntroVAE(cdim=3, zdim=32, channels=[64, 128, 256], image_size=32).to("cuda:0")
    T.load_model(m2, str(tmp_path / "saves" / sorted(ckpts)[-1]), torch.device("cuda:0"))
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_device_prefetcher_matches_host_pipeline():
    """uint8 batches through sivae_hip.data.DevicePrefetcher == ToTensor()-style host conversion, same order;
    float batches pass through unchanged"""
    from sivae_hip.data import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    imgs = torch.randint(0, 256, (20, 3, 16, 16), generator=g, dtype=torch.uint8)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(imgs), batch_size=8, shuffle=False)
    got = [b.clone() for b in DevicePrefetcher(loader, "cuda:0", take_first=True)]
    assert [b.shape[0] for b in got] == [8, 8, 4] and all(b.is_cuda and b.dtype == torch.float32 for b in got)
    assert torch.equal(torch.cat(got).cpu(), imgs.float() * (1.0 / 255.0))
    # mirror: every sample equals the source or its horizontal flip
    got = torch.cat([b.clone() for b in DevicePrefetcher(loader, "cuda:0", take_first=True, hflip=True, seed=1)]).cpu()
    ref = imgs.float() * (1.0 / 255.0)
    same = (got == ref).flatten(1).all(1)
    flipped = (got == ref.flip(3)).flatten(1).all(1)
    assert bool((same | flipped).all()) and bool(flipped.any()) and bool(same.any())
    fl = torch.rand(10, 3, 8, 8, generator=g)
    got = torch.cat([b.clone() for b in DevicePrefetcher(torch.utils.data.DataLoader(fl, batch_size=4), "cuda:0")])
    assert torch.equal(got.cpu(), fl)


def test_train_entry_point_with_uint8_input_pipeline(tmp_path, monkeypatch):
    import train_soft_intro_vae as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SIVAE_SYNTHETIC_IMAGES", "16")
    monkeypatch.setenv("SIVAE_SYNTHETIC_U8", "1")
    model = T.train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=16, batch_size=8, num_workers=0, num_epochs=1,
                                   beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=5, device=torch.device("cuda:0"))
    assert all(torch.isfinite(v).all() for v in model.state_dict().values() if v.is_floating_point())


def test_bootstrap_entry_point_runs(tmp_path, monkeypatch):
    import train_soft_intro_vae_bootstrap as TB
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SIVAE_SYNTHETIC_IMAGES", "16")
    model = TB.train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=16, batch_size=8, num_workers=0, num_epochs=1,
                                    beta_kl=1.0, beta_rec=1.0, beta_neg=256, seed=4, device=torch.device("cuda:0"))
    sd = model.state_dict()
    # after the epoch the target decoder is a copy of the decoder (bootstrap file :680-682)
    for k, v in sd.items():
        if k.startswith("decoder.") and "num_batches" not in k and "running" not in k:
            assert torch.equal(v, sd["target_" + k]), k


def test_pretrained_and_start_epoch_through_the_entry_point(tmp_path, monkeypatch):
    """`pretrained=` + `start_epoch=` (reference train_soft_intro_vae.py:443-444,471): the second call loads the first call's
    checkpoint (reference format) and runs epochs [start_epoch, num_epochs) only.  With lr 0 nothing but the BatchNorm
    buffers may move: the weights stay bit-equal to the checkpoint, and the BatchNorm pass counter shows exactly one
    Soft-Intro epoch on top of the loaded one."""
    import train_soft_intro_vae as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SIVAE_SYNTHETIC_IMAGES", "32")
    dev = torch.device("cuda:0")
    kw = dict(dataset="synthetic-cifar10", z_dim=16, batch_size=16, num_workers=0, beta_kl=1.0, beta_rec=1.0, beta_neg=256,
              seed=7, test_iter=1000, device=dev)
    m1 = T.train_soft_intro_vae(num_epochs=1, save_interval=1, **kw)
    sd1 = {k: v.clone() for k, v in m1.state_dict().items()}
    ckpts = sorted(c for c in os.listdir(tmp_path / "saves") if c.endswith(".pth"))
    assert ckpts
    path = str(tmp_path / "saves" / ckpts[-1])
    saved = torch.load(path, map_location="cpu")
    assert set(saved) == {"epoch", "model"}
    m2 = T.train_soft_intro_vae(num_epochs=2, start_epoch=1, pretrained=path, lr_e=0.0, lr_d=0.0, save_interval=50, **kw)
    sd2 = m2.state_dict()
    for k, v in saved["model"].items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        assert torch.equal(sd2[k].cpu(), v), k  # (started from the checkpoint, lr 0: unchanged)
    # the constructor's probe pass does not survive load_model (the counter is loaded): checkpoint value + one epoch of
    # 2 iterations x 5 encoder passes + the deterministic dump at cur_iter 0 (test_iter) + the final dump
    n0 = int(saved["model"]["encoder.main.1.num_batches_tracked"])
    assert n0 == int(sd1["encoder.main.1.num_batches_tracked"])
    assert int(sd2["encoder.main.1.num_batches_tracked"]) == n0 + 2 * 5 + 1 + 1
    # start_epoch == num_epochs: no epoch runs at all, the loaded model comes back as is
    m3 = T.train_soft_intro_vae(num_epochs=1, start_epoch=1, pretrained=path, save_interval=50, **kw)
    for k, v in saved["model"].items():
        assert torch.equal(m3.state_dict()[k].cpu(), v), k


def test_exit_on_negative_diff_through_the_entry_point(tmp_path, monkeypatch):
    """`exit_on_negative_diff` (reference :652-657): after epoch 50 an epoch whose mean kl_fake - kl_real is below -1
    aborts with SystemError("Negative KL Difference"); with the flag off the same run completes.  The statistics come from
    the real engine (one real iteration per epoch on the GPU) with kl_fake shifted down by 5 on the device, so the abort
    is decided by the loop's own epoch means."""
    import train_soft_intro_vae as T
    from sivae_hip import engine as E
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SIVAE_SYNTHETIC_IMAGES", "8")
    orig = E.SoftIntroEngine.soft_intro_step
    i_fake, i_real = E.STAT_NAMES.index("kl_fake"), E.STAT_NAMES.index("kl_real")
    calls = []

    def shifted(self, real, *a, **k):
        out = orig(self, real, *a, **k)
        st = out["stats"].clone()
        st[i_fake] = st[i_real] - 5.0
        out["stats"] = st
        calls.append(1)
        return out
    monkeypatch.setattr(E.SoftIntroEngine, "soft_intro_step", shifted)
    kw = dict(dataset="synthetic-cifar10", z_dim=16, batch_size=8, num_workers=0, beta_kl=1.0, beta_rec=1.0, beta_neg=256,
              seed=9, test_iter=100000, save_interval=1000, device=torch.device("cuda:0"))
    with pytest.raises(SystemError, match="Negative KL Difference"):
        T.train_soft_intro_vae(num_epochs=60, exit_on_negative_diff=True, **kw)
    assert len(calls) == 52  # epochs 0..51: the check needs epoch > 50 (reference :652)
    calls.clear()
    T.train_soft_intro_vae(num_epochs=53, exit_on_negative_diff=False, **kw)
    assert len(calls) == 53


def test_nan_raises_system_error(tmp_path, monkeypatch):
    import train_soft_intro_vae as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SIVAE_SYNTHETIC_IMAGES", "8")
    with pytest.raises(SystemError):
        T.train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=16, batch_size=8, num_workers=0, num_epochs=1,
                               beta_kl=float("nan"), device=torch.device("cuda:0"))


def test_2d_loop_matches_reference_recording():
    """replays loop_2d.npz (the reference's unmodified train_soft_intro_vae_toy, recorded) on the HIP 2-D path"""
    import train_soft_intro_vae_2d as T2
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(GOLD, "loop_2d.npz"))
    model = T2.SoftIntroVAESimple(x_dim=2, zdim=2, n_layers=3, num_hidden=int(fx["meta_num_hidden"]))
    model.load_state_dict({k[len("init/"):]: torch.from_numpy(np.array(fx[k])) for k in fx.files
                           if k.startswith("init/")}, strict=True)
    model = model.to(dev).train()
    lr = float(fx["hp_lr_e"])
    opt_e = FlatAdam(model.encoder.parameters(), lr=lr)
    opt_d = FlatAdam([p for n, p in model.decoder.named_parameters() if n != "loggamma"], lr=lr)
    hp = dict(beta_rec=float(fx["hp_beta_rec"]), beta_kl=float(fx["hp_beta_kl"]), beta_neg=float(fx["hp_beta_neg"]))
    draws = [torch.from_numpy(fx["draw%d" % i]).to(dev) for i in range(int(fx["meta_n_draws"]))]
    di = 0
    for it in range(int(fx["meta_n_iter"])):
        batch = torch.from_numpy(fx["batch%d" % it]).to(dev)
        if it < int(fx["meta_num_vae"]):
            T2.vae_iteration_2d(model, opt_e, opt_d, batch, hp, eps=draws[di])
            di += 1
        else:
            res = T2.soft_intro_iteration_2d(model, opt_e, opt_d, batch, hp, noise=draws[di], eps=draws[di + 1:di + 6])
            di += 6
    torch.cuda.synchronize()
    assert torch.isfinite(res["lossE"]) and torch.isfinite(res["lossD"])
    sd = model.state_dict()
    ds = []
    for k in fx.files:
        if k.startswith("final/"):
            ds.append((sd[k[len("final/"):]].double().cpu().numpy() - fx[k].astype(np.float64)).ravel())
    d = np.abs(np.concatenate(ds)) / lr
    assert np.median(d) <= 0.05 and (d > 1.0).mean() <= 0.01, (float(d.max()), float(np.median(d)))


def test_2d_toy_entry_point_runs(tmp_path, monkeypatch):
    import train_soft_intro_vae_2d as T2
    monkeypatch.chdir(tmp_path)
    model = T2.train_soft_intro_vae_toy(z_dim=2, batch_size=64, n_iter=6, num_vae=2, save_interval=5000, beta_kl=0.3,
                                        beta_rec=0.2, beta_neg=0.9, test_iter=3, seed=92, device=torch.device("cuda:0"),
                                        dataset="8Gaussians")
    assert all(torch.isfinite(v).all() for v in model.state_dict().values())
    assert os.path.exists(tmp_path / "results_log_soft_intro_vae.txt")


def test_hip_graph_replay_matches_eager():
    """whole-iteration HIP graph (device-side Adam step count / lr, device-side Philox position) == the eager loop"""
    import train_soft_intro_vae as T
    from sivae_hip import rng
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    batches = [torch.rand(8, 3, 32, 32, generator=g).to(dev) for _ in range(5)]

    def build():
        torch.manual_seed(11)
        m = T.SoftIntroVAE(cdim=3, zdim=16, channels=[32, 64], image_size=32).to(dev).train()
        oe, od = FlatAdam(m.encoder.parameters(), lr=2e-4), FlatAdam(m.decoder.parameters(), lr=2e-4)
        return m, SoftIntroEngine(m, oe, od, beta_kl=1.0, beta_rec=1.0, beta_neg=256.0)

    rng.manual_seed(5)
    m_a, e_a = build()
    for b in batches:
        stats_a = e_a.soft_intro_step(b)["stats"].clone()
    rng.manual_seed(5)
    m_b, e_b = build()
    e_b.soft_intro_step(batches[0])
    e_b.capture(batches[1], warmup=1)          # warm-up iteration = batch 1
    for b in batches[2:]:
        stats_b = e_b.replay(b)["stats"].clone()
    torch.cuda.synchronize()
    assert e_b.opt_e.t == e_a.opt_e.t == 5
    assert int(e_b.opt_e.dev_state[0].item()) == 5
    # Same arithmetic up to the last bit of the two Adam bias-correction factors (device pow() vs python's); the
    # first Adam steps are sign-like, so a noise-level gradient element can still end up a few lr apart
    # -> drift criterion in units of lr, as for the recorded reference loops
    lr = 2e-4
    drift = []
    for (k, va), (_, vb) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        if va.is_floating_point() and "running" not in k:
            drift.append(((va - vb).abs() / lr).flatten().cpu())
        elif not va.is_floating_point():
            assert torch.equal(va, vb), k
    d = torch.cat(drift)
    assert float(d.median()) <= 0.05 and float((d > 1.0).float().mean()) <= 0.01, (float(d.median()), float(d.max()))
    assert torch.allclose(stats_a, stats_b, rtol=2e-2, atol=1e-6), (stats_a, stats_b)
