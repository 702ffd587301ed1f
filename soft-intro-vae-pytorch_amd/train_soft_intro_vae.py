"""MI355X-native drop-in for the reference module `soft_intro_vae/train_soft_intro_vae.py`.

Put this directory on PYTHONPATH ahead of the reference's and its `main.py` (`from train_soft_intro_vae
import train_soft_intro_vae`, main.py:8,47-52) runs unchanged: same public names, same signatures, same
state_dict keys, same error behaviour — but every network pass, the Gaussian sampler and the KL /
reconstruction / exp-ELBO losses run in hand-written HIP kernels for gfx950 (libsivae_hip.so), gradients
live in two flat buffers, Adam is one fused launch per network and logging costs one device->host copy per
iteration.  There is no CPU path: a CPU device raises (the reference's CPU semantics live in oracle/, which
only tests and the bench's cpu_baseline leg use).

Extra dataset keys (not in the reference): "synthetic-<name>" (e.g. synthetic-cifar10, synthetic-celeb256)
trains on uniform-random images of <name>'s shape so the loop can be exercised without datasets.
"""
import os
import pickle
import random
import time

import numpy as np
import torch
import torch.nn as nn

from sivae_hip import data as _data
from sivae_hip import dp as _dp
from sivae_hip import engine as _engine
from sivae_hip import ops as _ops
from sivae_hip import rng as _rng
from sivae_hip.engine import calc_kl, calc_reconstruction_loss, reparameterize  # noqa: F401  (reference API)
from sivae_hip.nn import Decoder, Encoder, ResidualBlock  # noqa: F401
from sivae_hip.optim import FlatAdam, MultiStepLR


class SoftIntroVAE(nn.Module):
    """reference: train_soft_intro_vae.py:172-223"""

    def __init__(self, cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256, conditional=False,
                 cond_dim=10):
        super().__init__()
        self.zdim = zdim
        self.conditional = conditional
        self.cond_dim = cond_dim
        self.encoder = Encoder(cdim, zdim, channels, image_size, conditional=conditional, cond_dim=cond_dim)
        self.decoder = Decoder(cdim, zdim, channels, image_size, conditional=conditional,
                               conv_input_size=self.encoder.conv_output_size, cond_dim=cond_dim)

    def forward(self, x, o_cond=None, deterministic=False):
        cond = o_cond if (self.conditional and o_cond is not None) else None
        mu, logvar = self.encode(x, o_cond=cond)
        z = mu if deterministic else reparameterize(mu, logvar)
        y = self.decode(z, y_cond=cond)
        return mu, logvar, z, y

    def sample(self, z, y_cond=None):
        return self.decode(z, y_cond=y_cond)

    def sample_with_noise(self, num_samples=1, device=torch.device("cpu"), y_cond=None):
        z = _rng.randn((num_samples, self.zdim), device)
        return self.decode(z, y_cond=y_cond)

    def encode(self, x, o_cond=None):
        if self.conditional and o_cond is not None:
            return self.encoder(x, o_cond=o_cond)
        return self.encoder(x)

    def decode(self, z, y_cond=None):
        if self.conditional and y_cond is not None:
            return self.decoder(z, y_cond=y_cond)
        return self.decoder(z)


# ---- small helpers of the reference surface -------------------------------------------------------------
def str_to_list(x):
    return [int(xi) for xi in x.split(",")]


def is_image_file(filename):
    return any(filename.endswith(ext) for ext in (".jpg", ".png", ".jpeg", ".bmp"))


def record_scalar(writer, scalar_list, scalar_name_list, cur_iter):
    names = scalar_name_list[1:-1].split(",")
    for idx, item in enumerate(scalar_list):
        writer.add_scalar(names[idx].strip(" "), item, cur_iter)


def record_image(writer, image_list, cur_iter, num_rows=8):
    from torchvision.utils import make_grid
    writer.add_image("visualization", make_grid(torch.cat(image_list, dim=0), nrow=num_rows), cur_iter)


def load_model(model, pretrained, device):
    weights = torch.load(pretrained, map_location=device)
    model.load_state_dict(weights["model"], strict=False)


def save_checkpoint(model, epoch, iteration, prefix=""):
    model_out_path = "./saves/" + prefix + "model_epoch_{}_iter_{}.pth".format(epoch, iteration)
    state = {"epoch": epoch, "model": model.state_dict()}
    os.makedirs("./saves/", exist_ok=True)
    torch.save(state, model_out_path)
    print("model checkpoint saved @ {}".format(model_out_path))


# ---- dataset table (reference :376-440) ------------------------------------------------------------------
_ARCH = {
    "cifar10": (32, [64, 128, 256], 3),
    "svhn": (32, [64, 128, 256], 3),
    "mnist": (28, [64, 128], 1),
    "fmnist": (28, [64, 128], 1),
    "celeb128": (128, [64, 128, 256, 512, 512], 3),
    "monsters128": (128, [64, 128, 256, 512, 512], 3),
    "celeb256": (256, [64, 128, 256, 512, 512, 512], 3),
    "celeb1024": (1024, [16, 32, 64, 128, 256, 512, 512, 512], 3),
}
_TUPLE_DATASETS = ("cifar10", "svhn", "fmnist", "mnist")


class _SyntheticImages(torch.utils.data.Dataset):
    """U[0,1) fp32 images (what ToTensor() yields); SIVAE_SYNTHETIC_U8=1: uint8 images as a decoder would hand them
    over — the device-side input pipeline (sivae_hip/data.py) then does the mirror + /255."""

    def __init__(self, n, ch, size, seed=1234):
        g = torch.Generator().manual_seed(seed)
        if os.environ.get("SIVAE_SYNTHETIC_U8", "0") == "1":
            self.data = torch.randint(0, 256, (n, ch, size, size), generator=g, dtype=torch.uint8)
        else:
            self.data = torch.rand(n, ch, size, size, generator=g)

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        return self.data[i]


def _build_dataset(dataset, image_size, ch):
    """-> torch Dataset. torchvision / the reference's dataset.py are imported lazily: only real datasets need them."""
    if dataset.startswith("synthetic-"):
        n = int(os.environ.get("SIVAE_SYNTHETIC_IMAGES", "1024"))
        return _SyntheticImages(n, ch, image_size)
    if dataset in ("cifar10", "svhn", "fmnist", "mnist"):
        from torchvision import transforms
        from torchvision.datasets import CIFAR10, MNIST, SVHN, FashionMNIST
        tf = transforms.ToTensor()
        if dataset == "cifar10":
            return CIFAR10(root="./cifar10_ds", train=True, download=True, transform=tf)
        if dataset == "svhn":
            return SVHN(root="./svhn", split="train", transform=tf, download=True)
        if dataset == "fmnist":
            return FashionMNIST(root="./fmnist_ds", train=True, download=True, transform=tf)
        return MNIST(root="./mnist_ds", train=True, download=True, transform=tf)
    from dataset import DigitalMonstersDataset, ImageDatasetFromFile  # the reference's dataset.py
    if dataset == "monsters128":
        return DigitalMonstersDataset(root_path="./monsters_ds/", output_height=image_size)
    train_size = 29000 if dataset == "celeb1024" else 162770
    data_root = "./" + dataset if dataset == "celeb1024" else "../data/celeb256/img_align_celeba"
    image_list = [x for x in os.listdir(data_root) if is_image_file(x)]
    train_list = image_list[:train_size]
    assert len(train_list) > 0
    return ImageDatasetFromFile(train_list, data_root, input_height=None, crop_height=None,
                                output_height=image_size, is_mirror=True)


def _save_image(tensor, path, nrow):
    try:
        import torchvision.utils as vutils
        vutils.save_image(tensor, path, nrow=nrow)
    except Exception:  # torchvision is optional here (absent in the build image): image dumps are skipped
        pass


def _require_device(device, local_rank=None):
    """-> the ROCm device this process trains on, made CURRENT (the kernels launch on the current stream of the
    tensors' device; making it current keeps every helper that asks torch for "the" stream in agreement).
    Under torchrun (WORLD_SIZE > 1) each rank takes the GPU of its LOCAL_RANK, the one-process-per-GPU convention of the
    reference's only DP precedent (style_soft_intro_vae/launcher.py:126-129); SIVAE_DP_SAME_DEVICE=1 keeps the given
    device on every rank (single-GPU testing of the DP path)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("train_soft_intro_vae (MI355X build): device %s is not a ROCm device; this engine has no "
                           "CPU path" % device)
    if local_rank is not None and os.environ.get("SIVAE_DP_SAME_DEVICE", "0") != "1":
        device = torch.device("cuda", local_rank)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    torch.cuda.set_device(device)
    return device


def train_soft_intro_vae(dataset="cifar10", z_dim=128, lr_e=2e-4, lr_d=2e-4, batch_size=128, num_workers=4,
                         start_epoch=0, exit_on_negative_diff=False,
                         num_epochs=250, num_vae=0, save_interval=50, recon_loss_type="mse",
                         beta_kl=1.0, beta_rec=1.0, beta_neg=1.0, test_iter=1000, seed=-1, pretrained=None,
                         device=torch.device("cpu"), num_row=8, gamma_r=1e-8, with_fid=False):
    """Same signature and behaviour as the reference's entry point (train_soft_intro_vae.py:337-341)."""
    return _train(dataset, z_dim, lr_e, lr_d, batch_size, num_workers, start_epoch, exit_on_negative_diff, num_epochs,
                  num_vae, save_interval, recon_loss_type, beta_kl, beta_rec, beta_neg, test_iter, seed, pretrained,
                  device, num_row, gamma_r, with_fid, bootstrap=False, copy_to_target_freq=1,
                  model_factory=SoftIntroVAE, tag="soft_intro")


def _train(dataset, z_dim, lr_e, lr_d, batch_size, num_workers, start_epoch, exit_on_negative_diff, num_epochs,
           num_vae, save_interval, recon_loss_type, beta_kl, beta_rec, beta_neg, test_iter, seed, pretrained, device,
           num_row, gamma_r, with_fid, bootstrap, copy_to_target_freq, model_factory, tag):
    # ---- data parallelism: under torchrun (WORLD_SIZE > 1) this entry point is one of N processes, one per
    # GPU; `batch_size` stays the GLOBAL batch of the reference's signature and is split per image over the ranks,
    # weights / Adam state are replicated (rank 0's initial values), the two flat gradient buffers are all-reduced once
    # per iteration each (RCCL; SIVAE_DP_BACKEND overrides the backend for tests), BatchNorm statistics stay per rank,
    # rank 0 writes checkpoints / figures / logs (precedent: style_soft_intro_vae/train_style_soft_intro_vae.py:154-161,
    # :231).  World size 1 is the reference's single-process run.
    world, rank, local = _dp.init(backend=os.environ.get("SIVAE_DP_BACKEND") or None)
    main_rank = rank == 0
    if seed != -1:
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        _rng.manual_seed(seed, rank)
        if main_rank:
            print("random seed: ", seed)
        stream_seed = seed
    else:
        # unseeded (the reference's default): every run / every call draws fresh noise, like torch.randn does.  Under DP
        # rank 0's draw is broadcast: DistributedSampler partitions the dataset only if every rank permutes it with the
        # SAME seed (the per-rank offset stays in the noise stream and in the prefetcher's mirror draws)
        stream_seed = int.from_bytes(os.urandom(7), "little")
        if world > 1:
            stream_seed = _dp.broadcast_int(stream_seed, device=torch.device("cuda", local)
                                            if os.environ.get("SIVAE_DP_SAME_DEVICE", "0") != "1" else None)
        _rng.manual_seed(stream_seed, rank)
    arch_key = dataset[len("synthetic-"):] if dataset.startswith("synthetic-") else dataset
    if arch_key not in _ARCH:
        raise NotImplementedError("dataset is not supported")
    image_size, channels, ch = _ARCH[arch_key]
    device = _require_device(device, local if world > 1 else None)
    train_set = _build_dataset(dataset, image_size, ch)
    shard_start, per_rank = _dp.shard_batch(batch_size, world, rank)

    model = model_factory(cdim=ch, zdim=z_dim, channels=channels, image_size=image_size).to(device)
    if pretrained is not None:
        load_model(model, pretrained, device)
    if main_rank:
        print(model)
    fig_dir = "./figures_" + dataset
    if main_rank:
        os.makedirs(fig_dir, exist_ok=True)

    optimizer_e = FlatAdam(model.encoder.parameters(), lr=lr_e)
    optimizer_d = FlatAdam(model.decoder.parameters(), lr=lr_d)
    e_scheduler = MultiStepLR(optimizer_e, milestones=(350,), gamma=0.1)
    d_scheduler = MultiStepLR(optimizer_d, milestones=(350,), gamma=0.1)
    grad_sync = None
    if world > 1:
        extra = [p.data for p in model.target_decoder.parameters()] if bootstrap else []
        _dp.broadcast_([optimizer_e.flat, optimizer_d.flat] + extra + [b for b in model.buffers()])
        grad_sync = _dp.GradSync()
    # SIVAE_DTYPE=bf16 selects the build-defined mixed-precision mode (BASELINE.json config 3); the reference's
    # signature has no such argument, hence the environment switch
    eng = _engine.SoftIntroEngine(model, optimizer_e, optimizer_d, beta_kl=beta_kl, beta_rec=beta_rec,
                                  beta_neg=beta_neg, gamma_r=gamma_r, recon_loss_type=recon_loss_type,
                                  bootstrap=bootstrap, grad_sync=grad_sync,
                                  compute_dtype=os.environ.get("SIVAE_DTYPE") or None)

    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(train_set, num_replicas=world, rank=rank,
                                                                  shuffle=True, seed=stream_seed % (2 ** 31))
    loader = torch.utils.data.DataLoader(train_set, batch_size=per_rank, shuffle=sampler is None, sampler=sampler,
                                         num_workers=num_workers, pin_memory=True)
    # torchvision datasets yield (img, label) (reference :510-511); 3-D batches are unsqueezed (:513-514)
    batches = _data.DevicePrefetcher(loader, device, take_first=arch_key in _TUPLE_DATASETS,
                                     hflip=dataset.startswith("synthetic-"), seed=(stream_seed + rank) % (2 ** 31))
    # how many iterations' statistics may sit on the device before they are read back and checked for NaN (the
    # reference checks every iteration, :625-626, at the price of a device sync per iteration; 1 reproduces that)
    nan_every = max(1, int(os.environ.get("SIVAE_NAN_CHECK_EVERY", "64")))
    start_time = time.time()
    cur_iter = 0
    hist = {k: [] for k in ("kl_real", "kl_fake", "kl_rec", "rec_err", "exp_elbo_f", "exp_elbo_r")}
    best_fid = None
    real_batch = None
    for epoch in range(start_epoch, num_epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        if main_rank and with_fid and ((epoch == 0) or (epoch >= 100 and epoch % 20 == 0) or epoch == num_epochs - 1):
            from metrics.fid_score import calculate_fid_given_dataset  # the reference's metrics package
            with torch.no_grad():
                print("calculating fid...")
                fid = calculate_fid_given_dataset(loader, model, batch_size, cuda=True, dims=2048, device=device,
                                                  num_images=50000)
                print("fid:", fid)
                if best_fid is None:
                    best_fid = fid
                elif best_fid > fid:
                    print("best fid updated: {} -> {}".format(best_fid, fid))
                    best_fid = fid
                    prefix = "{}_{}_betas_{}_{}_{}_fid_{}_".format(dataset, tag, beta_kl, beta_neg, beta_rec, fid)
                    save_checkpoint(model, epoch, cur_iter, prefix)
        # (every checkpoint is written at an epoch boundary, after the previous epoch's last drain(): a NaN iteration
        # raises before its weights can be saved, whatever SIVAE_NAN_CHECK_EVERY is)
        if main_rank and epoch % save_interval == 0 and epoch > 0:
            prefix = "{}_{}_betas_{}_{}_{}_".format(dataset, tag, beta_kl, beta_neg, beta_rec)
            save_checkpoint(model, (epoch // save_interval) * save_interval, cur_iter, prefix)
        if world > 1:
            _dp.barrier()  # the other ranks wait here, not inside the first all-reduce, while rank 0 runs FID / saves
        model.train()
        ep = {k: [] for k in hist}
        diff_kls = []
        pending = []  # stats vectors still on the device; read back in one go per logging interval

        def drain():
            # an abandoned persistent BatchNorm backward is reported here instead of trapping the GPU.  The check runs
            # whether or not statistics are pending (the VAE warm-up epochs queue none) and, like the NaN abort, it is
            # COLLECTIVE under DP: every rank drains at the same iterations, the flags are all-reduced and all ranks raise
            # together (a rank raising alone would leave the others in the next all-reduce)
            poisoned = _ops.bn_fused_poisoned()
            if _dp.any_rank(poisoned is not None, device if world > 1 else None):
                raise RuntimeError(poisoned or "sivae_hip: another rank's one-pass BatchNorm backward gave up at its grid "
                                               "barrier (see that rank's message); this iteration's gradients are invalid")
            if not pending:
                return
            rows = torch.stack(pending).cpu()  # ONE device->host copy for the whole interval
            pending.clear()
            if _dp.any_rank(bool(torch.isnan(rows[:, :2]).any()), device if world > 1 else None):
                raise SystemError
            for r in rows.tolist():
                s = dict(zip(_engine.STAT_NAMES, r))
                diff_kls.append(-s["kl_real"] + s["kl_fake"])
                ep["kl_real"].append(s["kl_real"])
                ep["kl_fake"].append(s["kl_fake"])
                ep["kl_rec"].append(s["kl_rec"])
                ep["rec_err"].append(s["loss_rec"])
                ep["exp_elbo_f"].append(s["expelbo_fake"])
                ep["exp_elbo_r"].append(s["expelbo_rec"])

        for real_batch in batches:  # device-resident fp32 NCHW, one batch prefetched (sivae_hip/data.py)
            if epoch < num_vae:
                res = eng.vae_step(real_batch)
                if main_rank and cur_iter % test_iter == 0:
                    _save_image(torch.cat([real_batch, res["rec"]], dim=0).cpu(),
                                "{}/image_{}.jpg".format(fig_dir, cur_iter), num_row)
            else:
                res = eng.soft_intro_step(real_batch)
                pending.append(res["stats"])
                if cur_iter % test_iter == 0:
                    drain()
                    with torch.no_grad():
                        _, _, _, rec_det = model(real_batch, deterministic=True)
                    k = min(real_batch.size(0), 16)
                    if main_rank:
                        _save_image(torch.cat([real_batch[:k], rec_det[:k], res["fake"][:k]], dim=0).cpu(),
                                    "{}/image_{}.jpg".format(fig_dir, cur_iter), num_row)
                elif len(pending) >= nan_every:
                    drain()
            cur_iter += 1
        drain()
        e_scheduler.step()
        d_scheduler.step()
        if bootstrap and epoch % copy_to_target_freq == 0:
            model.target_decoder.load_state_dict(model.decoder.state_dict())
        # epoch statistics: the mean over the GLOBAL batch stream (every rank logs and decides on the same numbers)
        keys = list(hist)
        local_means = [float(np.mean(ep[k])) if ep[k] else 0.0 for k in keys] + \
            [float(np.mean(diff_kls)) if diff_kls else 0.0]
        glob = _dp.mean_over_ranks(local_means, device) if world > 1 else local_means
        diff_kl_mean = glob[-1]
        if exit_on_negative_diff and epoch > 50 and diff_kl_mean < -1.0:
            if main_rank:
                print(f"the kl difference [{diff_kl_mean:.3f}] between fake and real is negative "
                      f"(no sampling improvement)")
                print("try to lower beta_neg hyperparameter")
                print("exiting...")
            raise SystemError("Negative KL Difference")
        if epoch > num_vae - 1:
            for k, v in zip(keys, glob):
                hist[k].append(v)
        if main_rank and epoch > num_vae - 1:
            print("#" * 50)
            print(f"Epoch {epoch} Summary:")
            print(f"beta_rec: {beta_rec}, beta_kl: {beta_kl}, beta_neg: {beta_neg}")
            print(f"rec: {hist['rec_err'][-1]:.3f}, kl: {hist['kl_real'][-1]:.3f}, "
                  f"kl_fake: {hist['kl_fake'][-1]:.3f}, kl_rec: {hist['kl_rec'][-1]:.3f}")
            print(f"diff_kl: {diff_kl_mean:.3f}, exp_elbo_f: {hist['exp_elbo_f'][-1]:.4e}, "
                  f"exp_elbo_r: {hist['exp_elbo_r'][-1]:.4e}")
            print(f"time: {time.time() - start_time}")
            print("#" * 50)
        if epoch == num_epochs - 1:
            with torch.no_grad():
                _, _, _, rec_det = model(real_batch, deterministic=True)
                fake = model.sample(_rng.randn((real_batch.size(0), z_dim), device))
                k = min(real_batch.size(0), 16)
                if main_rank:
                    _save_image(torch.cat([real_batch[:k], rec_det[:k], fake[:k]], dim=0).cpu(),
                                "{}/image_{}.jpg".format(fig_dir, cur_iter), num_row)
            if not main_rank:
                model.train()
                continue
            try:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                fig = plt.figure()
                ax = fig.add_subplot(1, 1, 1)
                for key, label in (("kl_real", "kl_real"), ("kl_fake", "kl_fake"), ("kl_rec", "kl_rec"),
                                   ("rec_err", "rec_err")):
                    ax.plot(np.arange(len(hist[key])), hist[key], label=label)
                ax.legend()
                plt.savefig("./{}_train_graphs.jpg".format(tag))
                plt.close(fig)
            except Exception:  # plotting is best-effort
                pass
            with open("./{}_train_graphs_data.pickle".format(tag), "wb") as fp:
                pickle.dump({"kl_real": hist["kl_real"], "kl_fake": hist["kl_fake"], "kl_rec": hist["kl_rec"],
                             "rec_err": hist["rec_err"]}, fp)
            prefix = "{}_{}_betas_{}_{}_{}_".format(dataset, tag, beta_kl, beta_neg, beta_rec)
            save_checkpoint(model, epoch, cur_iter, prefix)
            model.train()
    return model


if __name__ == "__main__":
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    try:
        train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=128, batch_size=32, num_workers=0, num_epochs=1,
                             beta_kl=1.0, beta_neg=256, beta_rec=1.0, device=dev, test_iter=1000)
    except SystemError:
        print("Error, probably loss is NaN, try again...")
