"""Output side of the hot path: batch generation with the training kernels.

The reference feeds its FID network from `model.sample(noise)` in a loop (soft_intro_vae/metrics/fid_score.py:241-250:
`torch.randn(batch, zdim)` -> `model_s.sample` -> `clip(images*255, 0, 255).astype(uint8)`), with the VAE left in
whatever mode it is in (its `model.eval()` is commented out, :233).  `generate` is that loop on the device: Philox
noise, the decoder through the same HIP kernels, the uint8 quantisation as one kernel — no host round trip per batch.

Inference uses the conv kernels' fusions as they are: in eval mode BatchNorm-1 is applied by conv2's load prologue
from the running statistics (the "BN folded into the conv" of SURVEY 8f-4 at no extra pass) and BatchNorm-2 + residual
+ LeakyReLU is the one streaming pass the residual add needs anyway.
"""
import torch

from . import ops, rng


@torch.no_grad()
def generate(model, num_images, batch_size=50, as_uint8=True, eval_mode=None, stream=None):
    """Yield ceil(num_images / batch_size) batches of `model.sample(noise)` on the model's device.

    as_uint8: quantise like the reference's FID feed (uint8 NCHW); False yields the raw fp32 images.
    eval_mode: None keeps the model's current mode (the reference's behaviour: train-mode BatchNorm, running
        statistics keep moving); True / False switch to eval / train for the duration of the loop.
    stream: a sivae_hip.rng.PhiloxStream for the noise (default: the process-wide stream)."""
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("sivae_hip.infer.generate: the model must be on a ROCm device (no CPU path)")
    draw = (stream or rng.default_stream()).randn
    was_training = model.training
    if eval_mode is not None:
        model.train(not eval_mode)
    try:
        for _ in range(0, int(num_images), int(batch_size)):
            noise = draw((int(batch_size), model.zdim), dev)
            images = model.sample(noise)
            yield ops.f32_to_u8(images.contiguous()) if as_uint8 else images
    finally:
        model.train(was_training)


@torch.no_grad()
def reconstruct(model, x, eval_mode=None):
    """deterministic reconstruction dump (train_soft_intro_vae.py:641-646,676-684): decode(mu(x))"""
    was_training = model.training
    if eval_mode is not None:
        model.train(not eval_mode)
    try:
        _, _, _, rec = model(x, deterministic=True)
        return rec
    finally:
        model.train(was_training)
