"""Device-side Gaussian stream: Philox4x32-10 keyed by (seed, offset), one kernel per draw.

Replaces torch.randn / torch.randn_like on the hot path (reference: noise_batch at
soft_intro_vae/train_soft_intro_vae.py:547 and the five reparameterize() draws :264). The stream is
keyed by (seed, rank, draw counter), so data-parallel ranks draw independent noise and a run is
reproducible for a given (seed, world size).
"""
import torch

from . import ops


class PhiloxStream:
    def __init__(self, seed=0, rank=0):
        self.reseed(seed, rank)

    def reseed(self, seed, rank=0):
        self.seed = (int(seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03 + 0x1234567) & (2 ** 64 - 1)
        self.offset = 0
        self.offset_dev = None

    def use_device_state(self, device):
        """Keep the stream position in a device scalar from now on (same sequence of normals): a captured HIP
        graph then draws fresh noise on every replay."""
        if self.offset_dev is None:
            self.offset_dev = torch.tensor([self.offset], dtype=torch.int64, device=device)
        return self.offset_dev

    def randn(self, shape, device):
        n = 1
        for s in shape:
            n *= int(s)
        if self.offset_dev is not None:
            return ops.randn_dev(tuple(shape), self.seed, self.offset_dev, device)
        out = ops.randn(tuple(shape), self.seed, self.offset, device)
        self.offset += (n + 3) // 4  # one Philox counter per 4 normals
        return out


_default = PhiloxStream(0, 0)


def default_stream():
    return _default


def manual_seed(seed, rank=0):
    _default.reseed(seed, rank)


def randn(shape, device):
    return _default.randn(shape, torch.device(device))
