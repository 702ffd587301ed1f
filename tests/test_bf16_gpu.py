"""bf16 mode (BASELINE.json configs[2]: "CelebA 128x128 SoftIntroVAE zdim=256, batch 128, 1xMI355X bf16") on the MI355X.

The reference trains in fp32 only, so the mode is build-defined and so is its tolerance.  What is promised and checked
here, against the fp32 CPU oracle (the pinned restatement of the reference) on identical seeded inputs and weights:
  * latents of one iteration — encoder mu / logvar, z of every pass — within BF16_TOL = 4e-2 of the oracle's, max-norm
    relative (bf16 keeps 8 mantissa bits: 2^-9 = 2e-3 per rounding, ~25 roundings deep through an encoder or decoder
    stack, amplified by BatchNorm); images (reconstructions, fakes, reconstructions of those) within BF16_IMG_TOL =
    1e-1 max-norm (measured 2-7e-2: the worst pixel of B*3*H*W after two network passes);
  * scalar losses within BF16_LOSS_TOL = 1.6e-2 relative (measured <= 8e-3);
  * gradients (fp32, flat buffers): cosine similarity >= 0.95 between each network's flat gradient and the oracle's
    fp32 gradient (measured 0.98 encoder / 0.9998 decoder at B = 8), every tensor within 0.4 relative L2 (bf16
    gradient storage; BatchNorm backward over B*H*W = 128 samples per channel is ill-conditioned);
  * the fp32 path is untouched by the switch (bit-identical outputs before / after a bf16 excursion).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16_TOL = 4e-2
BF16_IMG_TOL = 1e-1
BF16_LOSS_TOL = 1.6e-2
IMAGES = ("fake", "rec", "rec_rec", "rec_fake")
SCALARS = ("loss_rec", "kl_real", "lossE", "lossD", "expelbo_rec", "expelbo_fake", "loss_rec_rec", "loss_fake_rec")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _tol(k):
    return BF16_LOSS_TOL if k in SCALARS else (BF16_IMG_TOL if k in IMAGES else BF16_TOL)


def _rel2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _setup(cdim, zdim, channels, image_size, B, hp, dtype, seed=0):
    from oracle import sivae_oracle as O
    import train_soft_intro_vae as T
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    P = O.init_params(cdim, zdim, channels, image_size, seed=seed)
    model = T.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    model = model.to(dev).train()
    oe, od = FlatAdam(model.encoder.parameters(), lr=2e-4), FlatAdam(model.decoder.parameters(), lr=2e-4)
    eng = SoftIntroEngine(model, oe, od, beta_kl=hp["beta_kl"], beta_rec=hp["beta_rec"], beta_neg=hp["beta_neg"],
                          gamma_r=hp["gamma_r"], compute_dtype=dtype)
    grads = {}
    for tag, opt, net in (("E", oe, model.encoder), ("D", od, model.decoder)):
        orig = opt.step

        def step(grad_scale=1.0, _orig=orig, _net=net, _tag=tag, _opt=opt):
            grads[_tag] = {k: p.grad.detach().clone() for k, p in _net.named_parameters()}
            grads[_tag + "_flat"] = _opt.flat_grad.detach().clone()
            _orig(grad_scale)
        opt.step = step
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, cdim, image_size, image_size, generator=g)
    noise = torch.randn(B, zdim, generator=g)
    eps = [torch.randn(B, zdim, generator=g) for _ in range(5)]
    return O, P, model, eng, grads, real, noise, eps, dev


def _bf16_vs_oracle(cdim, zdim, channels, image_size, B, hp, seed=0, report=None):
    O, P, model, eng, grads, real, noise, eps, dev = _setup(cdim, zdim, channels, image_size, B, hp, "bf16", seed)
    deps = [e.to(dev) for e in eps]
    problems, table = [], []
    opt_e = O.Adam(P, O.trainable_keys(P, "encoder."), 2e-4)
    e = O.e_step(P, real, noise, eps[:3], hp, channels, image_size, False)
    g32 = {k: P[k].grad.clone() for k in O.trainable_keys(P, "encoder.")}
    opt_e.step()
    es = eng.e_step(real.to(dev), noise.to(dev), deps[:3], keep=True)
    for k, v in e.items():
        err = _rel(es["kept"][k], v)
        table.append(("E/" + k, err))
        if err > _tol(k):
            problems.append(("E/" + k, err))
    flat_ref = torch.cat([g32[k].flatten() for k in O.trainable_keys(P, "encoder.")])
    flat_hip = torch.cat([grads["E"][k[len("encoder."):]].flatten().cpu() for k in O.trainable_keys(P, "encoder.")])
    cos = float(torch.nn.functional.cosine_similarity(flat_ref.double(), flat_hip.double(), dim=0))
    table.append(("E/grad cosine", cos))
    if cos < 0.95:
        problems.append(("E/grad cosine", cos))
    worst = max((_rel2(grads["E"][k[len("encoder."):]], g32[k]), k) for k in g32)
    table.append(("E/grad worst rel-L2 (%s)" % worst[1], worst[0]))
    if worst[0] > 0.4:
        problems.append(("E/grad rel-L2", worst))
    # D-step from the oracle's post-E-step encoder weights
    with torch.no_grad():
        for k, p in model.encoder.named_parameters():
            p.copy_(P["encoder." + k].to(dev))
    from sivae_hip import functional as SF
    SF.bump_generation(model.encoder.parameters())
    d = O.d_step(P, real, noise, e["z"], eps[3:], hp, channels, image_size, False)
    ds = eng.d_step(real.to(dev), noise.to(dev), e["z"].to(dev), deps[3:], keep=True)
    for k, v in d.items():
        err = _rel(ds["kept"][k], v)
        table.append(("D/" + k, err))
        if err > _tol(k):
            problems.append(("D/" + k, err))
    dk = O.trainable_keys(P, "decoder.")
    flat_ref = torch.cat([P[k].grad.flatten() for k in dk])
    flat_hip = torch.cat([grads["D"][k[len("decoder."):]].flatten().cpu() for k in dk])
    cos = float(torch.nn.functional.cosine_similarity(flat_ref.double(), flat_hip.double(), dim=0))
    table.append(("D/grad cosine", cos))
    if cos < 0.95:
        problems.append(("D/grad cosine", cos))
    worst = max((_rel2(grads["D"][k[len("decoder."):]], P[k].grad), k) for k in dk)
    table.append(("D/grad worst rel-L2 (%s)" % worst[1], worst[0]))
    if worst[0] > 0.4:
        problems.append(("D/grad rel-L2", worst))
    print("\nbf16 mode vs fp32 oracle, %dx%d channels %s z%d B=%d" % (image_size, image_size, channels, zdim, B))
    for k, v in table:
        print("  %-58s %.3e" % (k, v))
    if report is not None:
        report.extend(table)
    return problems


def test_bf16_cifar_net_vs_fp32_oracle():
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8)
    assert not _bf16_vs_oracle(3, 128, [64, 128, 256], 32, 16, hp)


def test_bf16_mnist_net_vs_fp32_oracle():
    """the single-channel 28x28 network (reference :376-379: channels [64, 128], z 32): odd 7x7 maps, and the kw-packed
    5x5 layers with ONE image channel (5 of the 16 k-step channels used)"""
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8)
    assert not _bf16_vs_oracle(1, 32, [64, 128], 28, 16, hp, seed=2)


def test_bf16_celeb128_config_vs_fp32_oracle():
    """config 3's network exactly: 128x128, channels [64,128,256,512,512], z 256 (reference :381-386), at B = 8"""
    hp = dict(beta_rec=0.5, beta_kl=1.0, beta_neg=1024.0, gamma_r=1e-8)
    assert not _bf16_vs_oracle(3, 256, [64, 128, 256, 512, 512], 128, 8, hp, seed=1)


def test_fp32_path_untouched_by_bf16_switch():
    """an fp32 iteration, a bf16 iteration, then the same fp32 iteration on a fresh model: bit-identical to the first"""
    from sivae_hip.nn import set_compute_dtype
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8)
    outs = []
    for rep in range(2):
        O, P, model, eng, grads, real, noise, eps, dev = _setup(3, 32, [16, 32, 64], 32, 8, hp, "fp32")
        es = eng.e_step(real.to(dev), noise.to(dev), [t.to(dev) for t in eps[:3]], keep=True)
        outs.append({k: v.clone() for k, v in es["kept"].items()})
        if rep == 0:
            set_compute_dtype(model, "bf16")
            eng.soft_intro_step(real.to(dev))
            set_compute_dtype(model, "fp32")
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_bf16_training_tracks_fp32_training():
    """20 iterations of each mode from the same weights / batches / noise: finite, and the reconstruction loss and
    KL curves of the bf16 run stay within 5 % of the fp32 run's"""
    from sivae_hip import rng
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1e-8)
    curves = {}
    for dtype in ("fp32", "bf16"):
        O, P, model, eng, grads, real, noise, eps, dev = _setup(3, 64, [32, 64, 128], 32, 32, hp, dtype)
        rng.manual_seed(7)
        rows = []
        x = real.to(dev)
        for _ in range(20):
            rows.append(eng.soft_intro_step(x)["stats"])
        curves[dtype] = torch.stack(rows).cpu()
        assert torch.isfinite(curves[dtype]).all()
    a, b = curves["fp32"], curves["bf16"]
    # columns: lossE lossD loss_rec kl_real kl_fake kl_rec expelbo_rec expelbo_fake
    for col, name in ((2, "loss_rec"), (3, "kl_real")):
        rel = ((a[:, col] - b[:, col]).abs() / a[:, col].abs()).max()
        print("bf16 vs fp32 training, %s: max relative deviation over 20 iterations %.3e" % (name, float(rel)))
        assert rel < 0.05, (name, float(rel))


def test_bf16_bootstrap_iteration_vs_fp32_oracle():
    """config 5's variant (frozen target decoder, gamma_r = 1, un-detached D-step targets) in the bf16 mode: one full
    iteration on the 32x32 net, same tolerances as the plain variant"""
    from oracle import sivae_oracle as O
    import train_soft_intro_vae_bootstrap as TB
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    cdim, zdim, channels, image_size, B = 3, 64, [32, 64, 128], 32, 16
    hp = dict(beta_rec=1.0, beta_kl=1.0, beta_neg=256.0, gamma_r=1.0)
    P = O.init_params(cdim, zdim, channels, image_size, seed=3, bootstrap=True)
    model = TB.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    model = model.to(dev).train()
    eng = SoftIntroEngine(model, FlatAdam(model.encoder.parameters(), lr=2e-4),
                          FlatAdam(model.decoder.parameters(), lr=2e-4), beta_kl=1.0, beta_rec=1.0, beta_neg=256.0,
                          gamma_r=1.0, bootstrap=True, compute_dtype="bf16")
    g = torch.Generator().manual_seed(1234)
    real = torch.rand(B, cdim, image_size, image_size, generator=g)
    noise = torch.randn(B, zdim, generator=g)
    eps = [torch.randn(B, zdim, generator=g) for _ in range(5)]
    e = O.e_step(P, real, noise, eps[:3], hp, channels, image_size, True)
    es = eng.e_step(real.to(dev), noise.to(dev), [t.to(dev) for t in eps[:3]], keep=True)
    bad = [(k, _rel(es["kept"][k], v)) for k, v in e.items() if _rel(es["kept"][k], v) > _tol(k)]
    assert not bad, bad
    target_before = {k: v.clone() for k, v in model.target_decoder.state_dict().items() if "running" not in k and "num_b" not in k}
    ds = eng.d_step(real.to(dev), noise.to(dev), es["z"], [t.to(dev) for t in eps[3:]])
    torch.cuda.synchronize()
    assert torch.isfinite(ds["lossD"]).all()
    for k, v in target_before.items():  # the target decoder is frozen in both steps
        assert torch.equal(model.target_decoder.state_dict()[k], v), k


def test_bf16_eval_mode_inference_vs_fp32_oracle():
    """eval-mode BatchNorm (running statistics) through the bf16 kernels: deterministic encode / decode / sample vs the
    fp32 oracle with training=False at the bf16 tolerances; no buffer moves"""
    import train_soft_intro_vae as T
    from oracle import sivae_oracle as O
    from sivae_hip.nn import set_compute_dtype
    dev = torch.device("cuda:0")
    channels, image_size, zdim, B = [32, 64, 128], 32, 24, 6
    torch.manual_seed(3)
    model = set_compute_dtype(T.SoftIntroVAE(cdim=3, zdim=zdim, channels=channels, image_size=image_size), "bf16")
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for _ in range(3):  # move the running statistics away from their initial values
            model(torch.rand(B, 3, image_size, image_size, generator=g).to(dev))
    model.eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = torch.rand(B, 3, image_size, image_size, generator=g)
    z = torch.randn(B, zdim, generator=g)
    with torch.no_grad():
        mu, logvar, zz, rec = model(x.to(dev), deterministic=True)
        smp = model.sample(z.to(dev))
        mu_o, logvar_o = O.encode(P, x, channels, image_size, training=False)
        rec_o = O.decode(P, mu_o, channels, image_size, training=False)
        smp_o = O.decode(P, z, channels, image_size, training=False)
    assert _rel(mu, mu_o) <= BF16_TOL and _rel(logvar, logvar_o) <= BF16_TOL, (_rel(mu, mu_o), _rel(logvar, logvar_o))
    assert _rel(rec, rec_o) <= BF16_IMG_TOL and _rel(smp, smp_o) <= BF16_IMG_TOL, (_rel(rec, rec_o), _rel(smp, smp_o))
    for k, v in model.state_dict().items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert torch.equal(v.cpu(), P[k]), k


# tolerances of the bf16 mode against the REFERENCE's fp32 arrays on the reduced-width 128x128 five-level topology at B = 2
# (BatchNorm statistics over 2 images: the deepest levels normalise over 32 samples) — 2x the measured worst case:
# latents 3.2e-2 (rec_logvar), images 5.9e-2 (rec_fake), scalar losses 7.7e-3 (kl_real)
BF16_FIX_LATENT_TOL = 6.4e-2
BF16_FIX_IMG_TOL = 1.2e-1
BF16_FIX_LOSS_TOL = 1.6e-2


def test_bf16_iteration_vs_reference_fixture_celeb128_narrow():
    """config 3's topology (128x128, five levels) in bf16 mode against the arrays captured from the imported reference
    (tests/golden/step_celeb128_narrow.npz, fp32): E-step forward quantities and losses at the stated tolerances"""
    import os
    import numpy as np
    import test_e2e_gpu as E
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(E.GOLD, "step_celeb128_narrow.npz"))
    model, boot = E._build(fx, dev)
    hp = {k: float(fx["hp_" + k]) for k in ("beta_rec", "beta_kl", "beta_neg", "gamma_r")}
    lr = float(fx["hp_lr"])
    eng = SoftIntroEngine(model, FlatAdam(model.encoder.parameters(), lr=lr), FlatAdam(model.decoder.parameters(), lr=lr),
                          beta_kl=hp["beta_kl"], beta_rec=hp["beta_rec"], beta_neg=hp["beta_neg"], gamma_r=hp["gamma_r"],
                          compute_dtype="bf16")
    real = torch.from_numpy(fx["real"]).to(dev)
    noise = torch.from_numpy(fx["noise"]).to(dev)
    eps = [torch.from_numpy(fx["eps%d" % i]).to(dev) for i in range(5)]
    es = eng.e_step(real, noise, eps[:3], keep=True)
    table, bad = [], []
    for k, v in es["kept"].items():
        err = E._rel_fx(v, fx, "E/" + k)
        tol = BF16_FIX_LOSS_TOL if k in SCALARS else (BF16_FIX_IMG_TOL if k in IMAGES else BF16_FIX_LATENT_TOL)
        table.append((k, err, tol))
        if not err <= tol:
            bad.append((k, err, tol))
    print("\nbf16 mode vs reference fixture step_celeb128_narrow (B = 2):")
    for k, err, tol in table:
        print("  E/%-16s %.3e  (tol %.1e)" % (k, err, tol))
    assert not bad, bad


@pytest.mark.gpu
def test_batched_repack16_rebuilds_every_bf16_slab_in_place():
    """bf16 mode: FlatAdam.step() rebuilds every cached bf16 operand slab of a network with ONE launch (sivae_pack_batch,
    form 6).  After three iterations every slab must equal a fresh per-weight pack of the CURRENT weight bit for bit, its
    cache entry must be valid for the current weight (no lazy rebuild left; the slabs of the virtual 5 x 1 weights — permuted
    copies — are dropped by the step and rebuilt on demand), and training must agree bit for bit with the unbatched path."""
    import train_soft_intro_vae as T
    from sivae_hip import functional as SF
    from sivae_hip import functional16 as SF16
    from sivae_hip import ops16
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    real = torch.rand(8, 3, 64, 64, generator=g).to(dev)
    noise = torch.randn(8, 32, generator=g).to(dev)
    eps = [torch.randn(8, 32, generator=g).to(dev) for _ in range(5)]
    finals = {}
    for batch in (True, False):
        SF.PACK_BATCH = batch
        try:
            torch.manual_seed(11)
            model = T.SoftIntroVAE(cdim=3, zdim=32, channels=[32, 64, 64], image_size=64).to(dev).train()
            oe, od = FlatAdam(model.encoder.parameters(), lr=2e-4), FlatAdam(model.decoder.parameters(), lr=2e-4)
            eng = SoftIntroEngine(model, oe, od, beta_kl=1.0, beta_rec=0.5, beta_neg=256.0, compute_dtype="bf16")
            for _ in range(3):
                out = eng.soft_intro_step(real, noise, eps)
            torch.cuda.synchronize()
            finals[batch] = ({k: v.detach().clone() for k, v in model.state_dict().items()}, out["stats"].clone())
            if not batch:
                continue
            assert "_sivae_pack16_plan" in oe.__dict__ and "_sivae_pack16_plan" in od.__dict__
            n = 0
            for p in list(model.encoder.parameters()) + list(model.decoder.parameters()):
                for (mode, virt), (tag, obj) in p.__dict__.get("_sivae_pack16", {}).items():
                    assert tag == (p._version, getattr(p, "_sivae_gen", 0), p.data_ptr(), SF.cache_epoch()), \
                        "cache entry not re-validated by the batched repack"
                    if virt is not None:
                        # (a virtual-weight slab is dropped by its optimizer's step and rebuilt on demand: the encoder's is
                        # back, with a current tag, because the decoder step runs the encoder after the encoder's update)
                        fresh = ops16.PackedW16(SF16._virtual(p.detach(), virt), mode)
                        assert torch.equal(obj.data.view(torch.int16), fresh.data.view(torch.int16)), (mode, virt)
                        continue
                    fresh = ops16.PackedW16(p.detach(), mode)
                    assert torch.equal(obj.data.view(torch.int16), fresh.data.view(torch.int16)), (mode, tuple(p.shape))
                    assert SF16.packed16(p, mode) is obj  # (a hit: no lazy rebuild)
                    n += 1
            assert n >= 20, n
        finally:
            SF.PACK_BATCH = True
    for k, v in finals[True][0].items():
        assert torch.equal(v, finals[False][0][k]), k  # same kernels, same operands: bit-identical training
    assert torch.equal(finals[True][1], finals[False][1])
