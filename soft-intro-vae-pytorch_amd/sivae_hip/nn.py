"""Module-level mirror of the reference networks, backed by the fused HIP blocks in `functional`.

Class names, constructor signatures, attribute names and state_dict keys are the reference's
(soft_intro_vae/train_soft_intro_vae.py:38-223, bootstrap variant
soft_intro_vae_bootstrap/train_soft_intro_vae_bootstrap.py:44-246) so checkpoints are interchangeable and
`main.py` / `metrics/fid_score.py` work unchanged.  The stock torch.nn layers below are used only as
parameter/buffer containers (they give the reference's default initialisation and key names) — their
forward() is never called; Encoder/Decoder/ResidualBlock.forward dispatch to the HIP blocks.
"""
import os

import torch
import torch.nn as nn

from . import functional as SF
from . import functional16 as SF16


# SIVAE_DEFER_UPSAMPLE=0 materialises every nn.Upsample output as the reference does (A/B measurements)
DEFER_UPSAMPLE = os.environ.get("SIVAE_DEFER_UPSAMPLE", "1") != "0"


class ResidualBlock(nn.Module):
    """reference: train_soft_intro_vae.py:38-75"""

    def __init__(self, inc=64, outc=64, groups=1, scale=1.0):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("sivae_hip: grouped convolutions are not used by Soft-IntroVAE")
        midc = int(outc * scale)
        if inc != outc:
            self.conv_expand = nn.Conv2d(inc, outc, kernel_size=1, stride=1, padding=0, groups=1, bias=False)
        else:
            self.conv_expand = None
        self.conv1 = nn.Conv2d(inc, midc, kernel_size=3, stride=1, padding=1, groups=groups, bias=False)
        self.bn1 = nn.BatchNorm2d(midc)
        self.relu1 = nn.LeakyReLU(0.2, inplace=True)
        self.conv2 = nn.Conv2d(midc, outc, kernel_size=3, stride=1, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(outc)
        self.relu2 = nn.LeakyReLU(0.2, inplace=True)

    def forward(self, x, post=None, cache=None, x_up=False, nseg=1, seg_rev=False, replay_update=True):
        """post in {None, 'pool', 'up', 'up_deferred'} fuses the AvgPool2d / Upsample that follows the block in the
        nets ('up_deferred': the next block reads this block's output through upsample addressing, x_up=True there).
        cache: see functional.ResBlockFn (activation cache for replaying an identical forward pass).
        A blocked bf16 input (bf16 mode, `set_compute_dtype`) runs the bf16 twin of the block.
        nseg > 1: x is a SEGMENTED batch (nseg passes laid end to end, per-pass BatchNorm statistics — functional.py)."""
        args = (x, None if self.conv_expand is None else self.conv_expand.weight, self.conv1.weight, self.bn1.weight,
                self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias, SF.BNState(self.bn1),
                SF.BNState(self.bn2), post, cache, x_up)
        if x.dtype == torch.bfloat16:
            return SF16.residual_block(*args, nseg, seg_rev, replay_update)
        return SF.residual_block(*args, nseg, seg_rev, replay_update)


def set_compute_dtype(module, dtype):
    """Select the arithmetic of every Encoder / Decoder under `module`: "fp32" (the parity path, default) or "bf16"
    (build-defined mixed precision for BASELINE.json config 3: bf16 activation storage + bf16 MFMA convs with fp32
    accumulation, fp32 BatchNorm statistics, master weights, losses and Adam — functional16.py)."""
    if dtype not in ("fp32", "bf16"):
        raise ValueError("compute dtype must be 'fp32' or 'bf16', got %r" % (dtype,))
    for m in module.modules():
        if isinstance(m, (Encoder, Decoder)):
            m.compute_dtype = dtype
    return module


def _block_weights(m):
    """the weight tuple a block's replay-cache tag is built from (same order as functional.ResBlockFn / ConvBiasFn)"""
    if isinstance(m, ResidualBlock):
        return (None if m.conv_expand is None else m.conv_expand.weight, m.conv1.weight, m.bn1.weight, m.bn1.bias,
                m.conv2.weight, m.bn2.weight, m.bn2.bias)
    if isinstance(m, nn.Conv2d):
        return (m.weight, m.bias)
    return ()


def segments_supported(image_size, seg_images, compute_dtype="fp32"):
    """can `nseg` passes of seg_images images each run as ONE segmented batch through these networks?  (power-of-two
    maps, and no row of conv-epilogue statistics may straddle two passes: fp32 — a pass must be a whole number of the
    4-image tile blocks of the 4x4 maps; bf16 mode — its pixel tiles hold up to 16 whole images of a 4x4 map, and the
    BatchNorm prologue must be off, functional16.MATERIALIZE_H)"""
    if compute_dtype == "bf16":
        return (SF16.MATERIALIZE_H and image_size >= 32 and (image_size & (image_size - 1)) == 0
                and seg_images % 16 == 0 and SF.ops.SYNC_BN is None)
    return (compute_dtype == "fp32" and image_size >= 32 and (image_size & (image_size - 1)) == 0
            and seg_images % 4 == 0 and SF.ops.SYNC_BN is None)


def _run_main(main, x, cache=None, bf16=None, nseg=1, seg_rev=False, replay_update=True):
    """Walk a reference-shaped nn.Sequential, dispatching each group of layers to its fused HIP block.
    x: fp32 NCHW, or a blocked bf16 activation (bf16 mode): the blocks dispatch on the input dtype unless `bf16` says
    otherwise (the bf16 encoder hands its stem the fp32 image)."""
    if bf16 is None:
        bf16 = x.dtype == torch.bfloat16
    F_ = SF16 if bf16 else SF
    mods = list(main.children())
    i, n = 0, len(mods)
    x_up = False  # x currently stands for Upsample(2,'nearest')(x): the consumers read it through upsample addressing
    stale = False  # an earlier block of this pass missed its replay cache
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        sub = None if cache is None else cache.setdefault(i, {})
        if sub is not None and sub.get("y") is not None:
            # a cached block is replayed only if its own weights are unchanged AND every block before it was replayed
            # too (a recomputed block hands its successors a new input)
            if stale or sub.get("tag") != SF.cache_tag(_block_weights(m)):
                sub.clear()
                stale = True
        if isinstance(m, ResidualBlock):
            if isinstance(nxt, nn.AvgPool2d):
                x = m(x, post="pool", cache=sub, x_up=x_up, nseg=nseg, seg_rev=seg_rev, replay_update=replay_update)
                x_up = False
                i += 2
            elif isinstance(nxt, nn.Upsample):
                # leave the Upsample to the next block when that is a ResidualBlock (its kernels read through
                # h>>1, w>>1; the residual add needs the upsampled width to be a multiple of 4)
                w_here = x.shape[3] * (2 if x_up else 1)  # (dim 3 is W in both layouts)
                defer = DEFER_UPSAMPLE and i + 2 < n and isinstance(mods[i + 2], ResidualBlock) and w_here % 2 == 0
                x = m(x, post="up_deferred" if defer else "up", cache=sub, x_up=x_up, nseg=nseg, seg_rev=seg_rev,
                      replay_update=replay_update)
                x_up = defer
                i += 2
            else:
                x = m(x, cache=sub, x_up=x_up, nseg=nseg, seg_rev=seg_rev, replay_update=replay_update)
                x_up = False
                i += 1
        elif isinstance(m, nn.Conv2d) and isinstance(nxt, nn.BatchNorm2d):
            # encoder stem: conv5x5 -> BN -> LeakyReLU -> AvgPool2d
            assert isinstance(mods[i + 2], nn.LeakyReLU) and isinstance(mods[i + 3], nn.AvgPool2d)
            x = F_.stem(x, m.weight, nxt.weight, nxt.bias, SF.BNState(nxt), nseg, seg_rev)
            i += 4
        elif isinstance(m, nn.Conv2d):
            x = F_.conv_bias(x, m.weight, m.bias, sub)
            i += 1
        else:
            raise RuntimeError("sivae_hip: unexpected layer %s in network" % type(m).__name__)
    return x


class Encoder(nn.Module):
    """reference: train_soft_intro_vae.py:78-122"""

    def __init__(self, cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256, conditional=False,
                 cond_dim=10):
        super().__init__()
        self.zdim = zdim
        self.cdim = cdim
        self.image_size = image_size
        self.conditional = conditional
        self.cond_dim = cond_dim
        self.compute_dtype = "fp32"
        cc = channels[0]
        self.main = nn.Sequential(
            nn.Conv2d(cdim, cc, 5, 1, 2, bias=False),
            nn.BatchNorm2d(cc),
            nn.LeakyReLU(0.2),
            nn.AvgPool2d(2),
        )
        sz = image_size // 2
        for ch in channels[1:]:
            self.main.add_module("res_in_{}".format(sz), ResidualBlock(cc, ch, scale=1.0))
            self.main.add_module("down_to_{}".format(sz // 2), nn.AvgPool2d(2))
            cc, sz = ch, sz // 2
        self.main.add_module("res_in_{}".format(sz), ResidualBlock(cc, cc, scale=1.0))
        self.conv_output_size = self.calc_conv_output_size()
        num_fc_features = self.conv_output_size[0] * self.conv_output_size[1] * self.conv_output_size[2]
        print("conv shape: ", self.conv_output_size)
        print("num fc features: ", num_fc_features)
        if self.conditional:
            self.fc = nn.Linear(num_fc_features + self.cond_dim, 2 * zdim)
        else:
            self.fc = nn.Linear(num_fc_features, 2 * zdim)

    def calc_conv_output_size(self):
        """The reference discovers the feature size by pushing zeros through `main` IN TRAINING MODE
        (train_soft_intro_vae.py:111-114), which also mutates every encoder BatchNorm buffer:
        running_mean <- 0, running_var <- 0.9*1 + 0.1*0 = 0.9, num_batches_tracked <- 1.
        Both effects are reproduced analytically here (no kernel launch at construction time)."""
        s = self.image_size
        n_pools = sum(1 for m in self.main.children() if isinstance(m, nn.AvgPool2d))
        for _ in range(n_pools):
            s = s // 2
        last = [m for m in self.main.children() if isinstance(m, ResidualBlock)][-1]
        with torch.no_grad():
            for m in self.main.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.running_mean.zero_()
                    m.running_var.fill_(0.9)
                    m.num_batches_tracked.fill_(1)
        return torch.Size([last.conv2.out_channels, s, s])

    def forward(self, x, o_cond=None, nseg=1, seg_rev=False):
        """nseg > 1: x is nseg independent batches laid end to end (a SEGMENTED batch): one pass of the kernels, per-segment
        BatchNorm statistics, running buffers updated once per segment in order (seg_rev: last first) — numerically the
        reference's nseg separate calls (train_soft_intro_vae.py:567-568, :601-605)."""
        if self.compute_dtype == "bf16":
            # (the stem takes the fp32 image itself: it chooses the operand layout of the 5x5 conv)
            y = SF16.from_blocked(_run_main(self.main, x, bf16=True, nseg=nseg, seg_rev=seg_rev),
                                  self.conv_output_size[0])
            y = y.reshape(x.size(0), -1)
        else:
            y = _run_main(self.main, x, nseg=nseg, seg_rev=seg_rev).reshape(x.size(0), -1)
        if self.conditional and o_cond is not None:
            y = torch.cat([y, o_cond], dim=1)
        y = SF.linear(y, self.fc.weight, self.fc.bias)
        mu, logvar = y.chunk(2, dim=1)
        return mu, logvar


class Decoder(nn.Module):
    """reference: train_soft_intro_vae.py:125-169"""

    def __init__(self, cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256, conditional=False,
                 conv_input_size=None, cond_dim=10):
        super().__init__()
        self.cdim = cdim
        self.image_size = image_size
        self.conditional = conditional
        self.compute_dtype = "fp32"
        cc = channels[-1]
        self.conv_input_size = conv_input_size
        if conv_input_size is None:
            num_fc_features = cc * 4 * 4
        else:
            num_fc_features = conv_input_size[0] * conv_input_size[1] * conv_input_size[2]
        self.cond_dim = cond_dim
        if self.conditional:
            self.fc = nn.Sequential(nn.Linear(zdim + self.cond_dim, num_fc_features), nn.ReLU(True))
        else:
            self.fc = nn.Sequential(nn.Linear(zdim, num_fc_features), nn.ReLU(True))
        sz = 4
        self.main = nn.Sequential()
        for ch in channels[::-1]:
            self.main.add_module("res_in_{}".format(sz), ResidualBlock(cc, ch, scale=1.0))
            self.main.add_module("up_to_{}".format(sz * 2), nn.Upsample(scale_factor=2, mode="nearest"))
            cc, sz = ch, sz * 2
        self.main.add_module("res_in_{}".format(sz), ResidualBlock(cc, cc, scale=1.0))
        self.main.add_module("predict", nn.Conv2d(cc, cdim, 5, 1, 2))

    def forward(self, z, y_cond=None, cache=None, nseg=1, seg_rev=False, replay_update=True, check_input=True):
        """cache (optional dict): filled by the first call, replayed by a second call with the SAME z and
        unchanged decoder weights — see SoftIntroEngine (the reference recomputes `fake` and `rec` in the
        D-step although the decoder has not changed since the E-step computed them).
        nseg > 1: z is a SEGMENTED batch (see Encoder.forward; the reference's pairs :607-608, bootstrap :635-636).
        replay_update=False / check_input=False: replay a `functional.cache_segment` view (one pass of a pair that already
        ran as a segmented batch): the running statistics were counted by that pass, and the cache belongs to the pair's
        input tensor, not to this z."""
        z = z.reshape(z.size(0), -1)
        if self.conditional and y_cond is not None:
            y_cond = y_cond.reshape(y_cond.size(0), -1)
            z = torch.cat([z, y_cond], dim=1)
        if cache is not None and check_input:
            # a filled cache replays the pass only for the SAME input tensor (storage, version and shape); anything
            # else starts a fresh fill.  (The weights are checked block by block: functional.cache_tag.)
            key = (z.data_ptr(), z._version, tuple(z.shape), self.compute_dtype, nseg, seg_rev)
            if cache.get("in_key") != key:
                cache.clear()
                cache["in_key"] = key
        y = SF.linear(z, self.fc[0].weight, self.fc[0].bias, relu=True)
        y = y.view(z.size(0), *self.conv_input_size)
        if self.compute_dtype == "bf16":
            y = SF16.to_blocked(y)
        return _run_main(self.main, y, cache, nseg=nseg, seg_rev=seg_rev, replay_update=replay_update)
