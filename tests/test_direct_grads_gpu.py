"""GPU: parameter gradients written straight into the optimizer's per-use slabs (functional._claim / FlatAdam.fold_slabs)
give the same flat gradient buffers as autograd's accumulation (`p.grad += g` per tensor and use — what the reference's
`lossE.backward()` / `lossD.backward()` do, train_soft_intro_vae.py:571, :619), in both arithmetic modes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(direct, dtype):
    from oracle import sivae_oracle as O
    import train_soft_intro_vae as T
    from sivae_hip import functional as SF
    from sivae_hip.engine import SoftIntroEngine
    from sivae_hip.optim import FlatAdam
    dev = torch.device("cuda:0")
    cdim, zdim, channels, image_size, B = 3, 32, [32, 64, 128], 32, 8
    old = SF.DIRECT_GRADS
    SF.DIRECT_GRADS = direct
    try:
        P = O.init_params(cdim, zdim, channels, image_size, seed=3)
        model = T.SoftIntroVAE(cdim=cdim, zdim=zdim, channels=channels, image_size=image_size)
        model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
        model = model.to(dev).train()
        oe, od = FlatAdam(model.encoder.parameters(), lr=2e-4), FlatAdam(model.decoder.parameters(), lr=2e-4)
        eng = SoftIntroEngine(model, oe, od, beta_kl=1.0, beta_rec=1.0, beta_neg=256.0, gamma_r=1e-8, compute_dtype=dtype)
        assert bool(oe.slabs) == direct and bool(od.slabs) == direct
        grads, used = {}, {}
        for tag, opt in (("E", oe), ("D", od)):
            orig = opt.step

            def step(grad_scale=1.0, _orig=orig, _tag=tag, _opt=opt):
                grads[_tag] = _opt.flat_grad.detach().clone()
                used[_tag] = _opt._slabs_used
                _orig(grad_scale)
            opt.step = step
        g = torch.Generator().manual_seed(11)
        real = torch.rand(B, cdim, image_size, image_size, generator=g).to(dev)
        noise = torch.randn(B, zdim, generator=g).to(dev)
        eps = [torch.randn(B, zdim, generator=g).to(dev) for _ in range(5)]
        eng.soft_intro_step(real, noise, eps)
        torch.cuda.synchronize()
        uses_after = max(p.__dict__.get("_sivae_use", 0) for p in list(model.parameters()))
        return grads, used, uses_after, (oe.flat.detach().clone(), od.flat.detach().clone()), eng._paired(real)
    finally:
        SF.DIRECT_GRADS = old


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_slab_gradients_equal_autograd_accumulation(dtype):
    gd, used, uses_after, wd, paired = _run(True, dtype)
    ga, used_a, _, wa, _ = _run(False, dtype)
    # the encoder runs three times inside lossE (:566-571), the decoder four times inside lossD (fake, rec and the two
    # reconstructions of their re-encodings, :601-619); with the same-weight pass pairs run as segmented batches (fp32
    # mode, engine.PAIR_PASSES) that is two uses each: real + [rec; fake], and [rec; fake] + [rec_rec; rec_fake]
    assert paired == (dtype == "fp32")
    assert used == ({"E": 2, "D": 2} if paired else {"E": 3, "D": 4})
    assert used_a == {"E": 0, "D": 0}
    assert uses_after == 0                   # the use counters are reset by the optimizer step
    for k in ("E", "D"):
        a, b = gd[k].double(), ga[k].double()
        assert torch.isfinite(a).all()
        # same per-use gradients, folded in forward order instead of autograd's accumulation order: fp32 rounding only.
        # (The decoder gradient is taken AFTER the encoder's Adam step: where the two encoder gradients differ in their
        # last bits — they do, by ~1e-7 relative — a near-zero element can change sign and Adam, sign-like on its first
        # steps, moves that weight by lr the other way; the D-step then sees a slightly different encoder.  Bit-equal
        # encoder gradients keep the 1e-6 gate, otherwise the decoder gradient is held to 1e-3.)
        tol = 1e-6 if (k == "E" or torch.equal(gd["E"], ga["E"])) else 1e-3
        assert float((a - b).abs().max() / b.abs().max()) <= tol, (k, tol)
    for a, b in zip(wd, wa):
        assert float((a.double() - b.double()).abs().max()) <= 1e-6 * float(b.abs().max()) + 4e-4  # (<= 2 Adam lr steps)
