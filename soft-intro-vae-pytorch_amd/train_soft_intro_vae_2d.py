"""MI355X-native drop-in for `soft_intro_vae_2d/train_soft_intro_vae_2d.py` (BASELINE config 1).

Same public names and signatures as the reference (ToyDataset, sample_2d_data, EncoderSimple,
DecoderSimple, SoftIntroVAESimple, calc_kl, calc_reconstruction_loss, reparameterize,
train_soft_intro_vae_toy, ...).  The MLP layers run on the ks=1 path of the MFMA conv kernels (Linear+ReLU
fused), the sampler and the KL / reconstruction / exp-ELBO losses on the loss kernels, Adam on the fused
flat-buffer kernel.  The toy distributions are host-side numpy/torch sampling exactly as in the reference
(they are the data source, not the hot path).  Like the image variants there is no CPU compute path:
the reference's CPU semantics are restated in oracle/sivae_oracle_2d.py for tests.
"""
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from sivae_hip import functional as SF
from sivae_hip import rng as _rng
from sivae_hip.engine import calc_reconstruction_loss, reparameterize  # noqa: F401
from sivae_hip.optim import FlatAdam, MultiStepLR

_SQ2 = 1.0 / np.sqrt(2.0)
_RING8 = [(1, 0), (-1, 0), (0, 1), (0, -1), (_SQ2, _SQ2), (_SQ2, -_SQ2), (-_SQ2, _SQ2), (-_SQ2, -_SQ2)]


# ---- toy data (reference :29-177) ---------------------------------------------------------------------------
class ToyDataset:
    """8Gaussians / 25Gaussians / Sequential8Gaussians via numpy's global RNG, the rest via sample_2d_data."""

    def __init__(self, distr="8Gaussians", dim=2, scale=2, iter_per_mode=100):
        self.distr, self.dim, self.scale = distr, dim, scale
        grid = []
        for _ in range(100000 // 25):
            for gx in range(-2, 3):
                for gy in range(-2, 3):
                    p = np.random.randn(2) * 0.05
                    p[0] += 2 * gx
                    p[1] += 2 * gy
                    grid.append(p)
        self.dataset = np.array(grid, dtype="float32")
        np.random.shuffle(self.dataset)
        self.dataset /= 2.828
        self.range = 2 if distr == "25Gaussians" else 1
        self.curr_iter, self.curr_mode, self.iter_per_mode = 0, 0, iter_per_mode

    def _ring(self, batch_size, sig, pick):
        centers = [(self.scale * cx, self.scale * cy) for cx, cy in _RING8]
        pts = []
        for _ in range(batch_size):
            p = np.random.randn(2) * sig
            c = pick(centers)
            p[0] += c[0]
            p[1] += c[1]
            pts.append(p)
        return torch.FloatTensor(np.array(pts, dtype="float32") / 1.414)

    def next_batch(self, batch_size=64, device=None, sig=0.02):
        if self.distr in ("2spirals", "checkerboard", "rings"):
            return sample_2d_data(self.distr, batch_size).to(device)
        if self.distr == "8Gaussians":
            return self._ring(batch_size, sig, random.choice).to(device)
        if self.distr == "25Gaussians":
            i = np.random.randint(100000 // batch_size)
            return torch.FloatTensor(self.dataset[i * batch_size:(i + 1) * batch_size]).to(device) * self.scale
        if self.distr == "Sequential8Gaussians":
            out = self._ring(batch_size, 0.02, lambda c: c[self.curr_mode]).to(device)
            if self.curr_iter % self.iter_per_mode == self.iter_per_mode - 1:
                self.curr_mode = (self.curr_mode + 1) % 8
            self.curr_iter += 1
            return out
        return None


def sample_2d_data(dataset, n_samples):
    z = torch.randn(n_samples, 2)
    if dataset == "8gaussians":
        centers = torch.tensor([(4 * cx, 4 * cy) for cx, cy in
                                [(1, 0), (-1, 0), (0, 1), (0, -1), (_SQ2, _SQ2), (-_SQ2, _SQ2), (_SQ2, -_SQ2),
                                 (-_SQ2, -_SQ2)]])
        return _SQ2 * (0.5 * z + centers[torch.randint(len(centers), size=(n_samples,))])
    if dataset == "2spirals":
        half = n_samples // 2
        n = torch.sqrt(torch.rand(half)) * 540 * (2 * np.pi) / 360
        dx = -torch.cos(n) * n + torch.rand(half) * 0.5
        dy = torch.sin(n) * n + torch.rand(half) * 0.5
        x = torch.cat([torch.stack([dx, dy], dim=1), torch.stack([-dx, -dy], dim=1)], dim=0) / 3
        return x + 0.1 * z
    if dataset == "checkerboard":
        x1 = torch.rand(n_samples) * 4 - 2
        x2 = torch.rand(n_samples) - torch.randint(0, 2, (n_samples,), dtype=torch.float) * 2 + x1.floor() % 2
        return torch.stack([x1, x2], dim=1) * 2
    if dataset == "rings":
        n4 = n3 = n2 = n_samples // 4
        n1 = n_samples - n4 - n3 - n2
        ang = [torch.linspace(0, 2 * np.pi, n + 1)[:-1] for n in (n4, n3, n2, n1)]
        # (the reference pairs cos(linspace4) with sin(linspace3) for the 0.75 ring, :160-161)
        xs = torch.cat([torch.cos(ang[0]), torch.cos(ang[0]) * 0.75, torch.cos(ang[2]) * 0.5, torch.cos(ang[3]) * 0.25])
        ys = torch.cat([torch.sin(ang[0]), torch.sin(ang[1]) * 0.75, torch.sin(ang[2]) * 0.5, torch.sin(ang[3]) * 0.25])
        x = torch.stack([xs, ys], dim=1) * 3.0
        x = x[torch.randint(0, n_samples, size=(n_samples,))]
        return x + torch.normal(mean=torch.zeros_like(x), std=0.08 * torch.ones_like(x))
    raise RuntimeError("Invalid `dataset` to sample from.")


# ---- helpers (reference :185-308) ----------------------------------------------------------------------------
def load_model(model, pretrained):
    model.load_state_dict(torch.load(pretrained)["model"])


def save_checkpoint(model, epoch, iteration, prefix=""):
    path = "./saves/" + prefix + "model_epoch_{}_iter_{}.pth".format(epoch, iteration)
    os.makedirs("./saves/", exist_ok=True)
    torch.save({"epoch": epoch, "model": model.state_dict()}, path)
    print("model checkpoint saved @ {}".format(path))


def setup_grid(range_lim=4, n_pts=1000, device=torch.device("cpu")):
    x = torch.linspace(-range_lim, range_lim, n_pts)
    xx, yy = torch.meshgrid((x, x), indexing="ij")
    return xx, yy, torch.stack((xx.flatten(), yy.flatten()), dim=1).to(device)


def calc_kl(logvar, mu, mu_o=10, is_outlier=False, reduce="sum"):
    """2-D variant of calc_kl (:290-308): KL to N(0, I), or to N(mu_o, I) when is_outlier."""
    red = reduce if reduce in ("sum", "mean") else "none"
    if is_outlier:
        return SF.kl(logvar, mu, float(mu_o), 0.0, red)
    return SF.kl(logvar, mu, 0.0, 0.0, red)


# ---- model (reference :402-483) --------------------------------------------------------------------------------
def _mlp_forward(main, x):
    layers = [m for m in main.children() if isinstance(m, nn.Linear)]
    h = x
    for i, lin in enumerate(layers):
        h = SF.linear(h, lin.weight, lin.bias, relu=(i + 1 < len(layers)))
    return h


def _make_mlp(din, dout, n_layers, num_hidden):
    main = nn.Sequential()
    main.add_module("input", nn.Linear(din, num_hidden))
    main.add_module("act0", nn.ReLU(True))
    for i in range(n_layers):
        main.add_module("hidden_%d" % (i + 1), nn.Linear(num_hidden, num_hidden))
        main.add_module("act_%d" % (i + 1), nn.ReLU(True))
    main.add_module("output", nn.Linear(num_hidden, dout))
    return main


class EncoderSimple(nn.Module):
    def __init__(self, x_dim=2, zdim=2, n_layers=2, num_hidden=64):
        super().__init__()
        self.xdim, self.zdim, self.n_layer, self.num_hidden = x_dim, zdim, n_layers, num_hidden
        self.main = _make_mlp(x_dim, zdim * 2, n_layers, num_hidden)

    def forward(self, x):
        y = _mlp_forward(self.main, x).view(x.size(0), -1)
        return y.chunk(2, dim=1)


class DecoderSimple(nn.Module):
    def __init__(self, x_dim=2, zdim=2, n_layers=2, num_hidden=64):
        super().__init__()
        self.xdim, self.zdim, self.n_layer, self.num_hidden = x_dim, zdim, n_layers, num_hidden
        self.loggamma = nn.Parameter(torch.tensor(0.0))  # present in the reference's state_dict, never used
        self.main = _make_mlp(zdim, x_dim, n_layers, num_hidden)

    def forward(self, z):
        return _mlp_forward(self.main, z.reshape(z.size(0), -1))


class SoftIntroVAESimple(nn.Module):
    def __init__(self, x_dim=2, zdim=2, n_layers=2, num_hidden=64):
        super().__init__()
        self.xdim, self.zdim, self.n_layer, self.num_hidden = x_dim, zdim, n_layers, num_hidden
        self.encoder = EncoderSimple(x_dim, zdim, n_layers, num_hidden)
        self.decoder = DecoderSimple(x_dim, zdim, n_layers, num_hidden)

    def forward(self, x, deterministic=False):
        mu, logvar = self.encode(x)
        z = mu if deterministic else reparameterize(mu, logvar)
        return mu, logvar, z, self.decode(z)

    def sample(self, z):
        return self.decode(z)

    def sample_with_noise(self, num_samples=1, device=torch.device("cpu")):
        return self.decode(_rng.randn((num_samples, self.zdim), device))

    def encode(self, x):
        return self.encoder(x)

    def decode(self, z):
        return self.decoder(z)


# ---- evaluation helpers (reference :345-394) ----------------------------------------------------------------------
def _neg_elbo(model, x, beta_kl, beta_recon):
    mu, logvar, _, rec = model(x, deterministic=True)
    err = calc_reconstruction_loss(x, rec, loss_type="mse", reduction="none")
    return beta_kl * calc_kl(logvar, mu, reduce="none") + beta_recon * err


def calculate_elbo_with_grid(model, evalset, test_grid, beta_kl=1.0, beta_recon=1.0, batch_size=512, num_iter=100,
                             device=torch.device("cpu")):
    model.eval()
    _, _, zz = test_grid
    with torch.no_grad():
        grid = torch.cat([_neg_elbo(model, zi.to(device), beta_kl, beta_recon) for zi in zz.split(batch_size, dim=0)])
        data = torch.cat([_neg_elbo(model, evalset.next_batch(batch_size=batch_size, device=device), beta_kl,
                                    beta_recon) for _ in range(num_iter)])
    return (data / torch.cat([grid, data]).sum()).mean().item()


def calculate_sample_kl(model, evalset, num_samples=5000, device=torch.device("cpu"), hist_bins=100, use_jsd=False,
                        xy_range=(-2, 2)):
    import torch.nn.functional as F
    rng2 = [[xy_range[0], xy_range[1]], [xy_range[0], xy_range[1]]]
    real = evalset.next_batch(batch_size=num_samples, device=device).cpu().numpy()
    fake = model.sample_with_noise(num_samples=num_samples, device=device).detach().cpu().numpy()
    hr = torch.tensor(np.histogram2d(real[:, 0], real[:, 1], bins=hist_bins, density=True, range=rng2)[0])
    hf = torch.tensor(np.histogram2d(fake[:, 0], fake[:, 1], bins=hist_bins, density=True, range=rng2)[0])
    if use_jsd:
        mid = 0.5 * (hf + hr)
        return (0.5 * (F.kl_div(torch.log(hr + 1e-14), mid, reduction="batchmean")
                       + F.kl_div(torch.log(hf + 1e-14), mid, reduction="batchmean"))).item()
    return F.kl_div(torch.log(hf + 1e-14), hr, reduction="batchmean").item()


# ---- training (reference :486-725) ---------------------------------------------------------------------------------
DIM_SCALE = 0.5


def soft_intro_iteration_2d(model, opt_e, opt_d, batch, hp, noise=None, eps=None):
    """One Soft-Intro iteration of the toy loop (:554-642). eps: optional 5 draws in the reference's order
    (real, fake, rec, rec, fake). Returns detached loss scalars."""
    br, bk, bn, gr, lt = hp["beta_rec"], hp["beta_kl"], hp["beta_neg"], hp.get("gamma_r", 1e-8), hp.get(
        "recon_loss_type", "mse")
    e = eps if eps is not None else [None] * 5
    if noise is None:
        noise = _rng.randn((batch.size(0), model.zdim), batch.device)
    for p in model.encoder.parameters():
        p.requires_grad = True
    for p in model.decoder.parameters():
        p.requires_grad = False
    fake = model.sample(noise)
    real_mu, real_logvar = model.encode(batch)
    z = reparameterize(real_mu, real_logvar, e[0])
    rec = model.decoder(z)
    loss_rec = calc_reconstruction_loss(batch, rec, loss_type=lt, reduction="mean")
    kl_real = calc_kl(real_logvar, real_mu, reduce="mean")
    fake_mu, fake_logvar = model.encode(fake.detach())
    rec_fake = model.decode(reparameterize(fake_mu, fake_logvar, e[1]))
    rec_mu, rec_logvar = model.encode(rec.detach())
    rec_rec = model.decode(reparameterize(rec_mu, rec_logvar, e[2]))
    e_fake = SF.expelbo(calc_reconstruction_loss(fake, rec_fake, loss_type=lt, reduction="none"),
                        calc_kl(fake_logvar, fake_mu, reduce="none"), DIM_SCALE, br, bn)
    e_rec = SF.expelbo(calc_reconstruction_loss(rec, rec_rec, loss_type=lt, reduction="none"),
                       calc_kl(rec_logvar, rec_mu, reduce="none"), DIM_SCALE, br, bn)
    lossE = DIM_SCALE * (bk * kl_real + br * loss_rec) + 0.25 * (e_fake + e_rec)
    opt_e.zero_grad()
    lossE.backward()
    opt_e.step()

    for p in model.encoder.parameters():
        p.requires_grad = False
    for p in model.decoder.parameters():
        p.requires_grad = True
    fake = model.sample(noise)
    rec = model.decoder(z.detach())
    loss_rec_d = calc_reconstruction_loss(batch, rec, loss_type=lt, reduction="mean")
    rec_mu, rec_logvar = model.encode(rec)
    z_rec = reparameterize(rec_mu, rec_logvar, e[3])
    fake_mu, fake_logvar = model.encode(fake)
    z_fake = reparameterize(fake_mu, fake_logvar, e[4])
    l_rr = calc_reconstruction_loss(rec.detach(), model.decode(z_rec.detach()), loss_type=lt, reduction="mean")
    l_rf = calc_reconstruction_loss(fake.detach(), model.decode(z_fake.detach()), loss_type=lt, reduction="mean")
    fkl = calc_kl(fake_logvar, fake_mu, reduce="mean")
    rkl = calc_kl(rec_logvar, rec_mu, reduce="mean")
    lossD = DIM_SCALE * (br * loss_rec_d + 0.5 * bk * (fkl + rkl) + gr * 0.5 * br * (l_rr + l_rf))
    opt_d.zero_grad()
    lossD.backward()
    opt_d.step()
    return dict(lossE=lossE.detach(), lossD=lossD.detach(), loss_rec=loss_rec_d.detach(), kl_real=kl_real.detach(),
                expelbo_fake=e_fake.detach(), expelbo_rec=e_rec.detach(), kl_fake=fkl.detach(), kl_rec=rkl.detach())


def vae_iteration_2d(model, opt_e, opt_d, batch, hp, eps=None):
    """vanilla-VAE iteration of the toy loop (:526-543)"""
    for p in model.parameters():
        p.requires_grad = True
    mu, logvar = model.encode(batch)
    rec = model.decode(reparameterize(mu, logvar, eps))
    loss_rec = calc_reconstruction_loss(batch, rec, loss_type=hp.get("recon_loss_type", "mse"), reduction="mean")
    loss_kl = calc_kl(logvar, mu, reduce="mean")
    loss = hp["beta_rec"] * loss_rec + hp["beta_kl"] * loss_kl
    opt_e.zero_grad()
    opt_d.zero_grad()
    loss.backward()
    opt_e.step()
    opt_d.step()
    return dict(loss=loss.detach(), loss_rec=loss_rec.detach(), loss_kl=loss_kl.detach())


def train_soft_intro_vae_toy(z_dim=2, lr_e=2e-4, lr_d=2e-4, batch_size=32, n_iter=30000, num_vae=0,
                             save_interval=1, recon_loss_type="mse", beta_kl=1.0, beta_rec=1.0,
                             beta_neg=1.0, test_iter=5000, seed=-1, pretrained=None, scale=1,
                             device=torch.device("cpu"), dataset="8Gaussians", gamma_r=1e-8):
    if seed != -1:
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        _rng.manual_seed(seed)
        print("random seed: ", seed)
    else:
        _rng.manual_seed(int.from_bytes(os.urandom(7), "little"))  # unseeded runs draw fresh noise, like torch.randn
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("train_soft_intro_vae_toy (MI355X build): device %s is not a ROCm device; this engine has "
                           "no CPU path" % device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    torch.cuda.set_device(device)  # kernels launch on the current stream of the tensors' device
    train_set = ToyDataset(distr=dataset)
    scale *= train_set.range
    model = SoftIntroVAESimple(x_dim=2, zdim=z_dim, n_layers=3, num_hidden=256).to(device)
    if pretrained is not None:
        load_model(model, pretrained)
    print(model)
    # `loggamma` has no gradient path; the reference hands it to Adam, which skips grad-less parameters
    opt_e = FlatAdam(model.encoder.parameters(), lr=lr_e)
    opt_d = FlatAdam([p for n, p in model.decoder.named_parameters() if n != "loggamma"], lr=lr_d)
    e_sched = MultiStepLR(opt_e, milestones=(10000, 15000), gamma=0.1)
    d_sched = MultiStepLR(opt_d, milestones=(10000, 15000), gamma=0.1)
    hp = dict(beta_rec=beta_rec, beta_kl=beta_kl, beta_neg=beta_neg, gamma_r=gamma_r, recon_loss_type=recon_loss_type)
    start = time.time()
    # NaN guard (reference: `if torch.isnan(lossD) or torch.isnan(lossE): raise SystemError` every iteration, :644-645):
    # a device-side flag accumulates isnan(lossE) | isnan(lossD) every iteration and is read back every
    # SIVAE_NAN_CHECK_EVERY iterations (default 64; 1 = the reference's cadence), before EVERY checkpoint and before the
    # final evaluation — a NaN model is never saved or scored.
    nan_flag = torch.zeros((), dtype=torch.float32, device=device)
    nan_every = max(1, int(os.environ.get("SIVAE_NAN_CHECK_EVERY", "64")))
    unchecked = 0

    def check_nan():
        nonlocal unchecked
        if unchecked and float(nan_flag.item()) != 0.0:
            raise SystemError("loss is NaN.")
        unchecked = 0

    for it in range(n_iter):
        batch = train_set.next_batch(batch_size=batch_size, device=device)
        if it % save_interval == 0 and it > 0:
            check_nan()
            save_checkpoint(model, (it // save_interval) * save_interval, it, "")
        model.train()
        if it < num_vae:
            res = vae_iteration_2d(model, opt_e, opt_d, batch, hp)
            if it % test_iter == 0:
                print("\nIter: {}/{} : time: {:4.4f}: Rec: {:.4f}, KL: {:.4f} ".format(
                    it, n_iter, time.time() - start, res["loss_rec"].item(), res["loss_kl"].item()))
        else:
            if batch.dim() == 3:
                batch = batch.unsqueeze(0)
            res = soft_intro_iteration_2d(model, opt_e, opt_d, batch, hp)
            nan_flag += torch.isnan(res["lossE"]).float() + torch.isnan(res["lossD"]).float()
            unchecked += 1
            if unchecked >= nan_every:
                check_nan()
            if it % test_iter == 0:
                check_nan()
                s = {k: v.item() for k, v in res.items()}  # one sync per logging interval
                print("\nIter: {}/{} : time: {:4.4f}: Rec: {:.4f}, Kl_E: {:.4f}, expELBO_R: {:.4f}, expELBO_F: {:.4f}, "
                      "Kl_F: {:.4f}, KL_R: {:.4f}, DIFF_Kl_F: {:.4f}".format(
                          it, n_iter, time.time() - start, s["loss_rec"], s["kl_real"], s["expelbo_rec"],
                          s["expelbo_fake"], s["kl_fake"], s["kl_rec"], -s["kl_real"] + s["kl_fake"]))
        e_sched.step()
        d_sched.step()
    check_nan()
    res = {}
    with torch.no_grad():
        res["sample_kl"] = calculate_sample_kl(model, train_set, num_samples=5000, device=device, hist_bins=100,
                                               use_jsd=False, xy_range=(-2 * scale, 2 * scale))
        res["jsd"] = calculate_sample_kl(model, train_set, num_samples=5000, device=device, hist_bins=100,
                                         use_jsd=True, xy_range=(-2 * scale, 2 * scale))
        grid = setup_grid(range_lim=scale * 2, n_pts=256, device=torch.device("cpu"))
        res["elbo"] = calculate_elbo_with_grid(model, train_set, test_grid=grid, beta_kl=1.0, beta_recon=1.0,
                                               device=device, batch_size=128)
    model.train()
    print("#" * 50)
    print(f"dataset: {dataset}, beta_kl: {beta_kl}, beta_rec: {beta_rec}, beta_neg: {beta_neg}")
    print(f'grid-normalized elbo: {res["elbo"]:.4e}, kl: {res["sample_kl"]:.4f}, jsd: {res["jsd"]:.4f}')
    print("#" * 50)
    with open("./results_log_soft_intro_vae.txt", "a") as fp:
        fp.write("{}_beta_kl_{}_beta_neg_{}_beta_rec_{}_gnelbo_{}_kl_{}_jsd_{}_seed_{}\n".format(
            dataset, beta_kl, beta_neg, beta_rec, res["elbo"], res["sample_kl"], res["jsd"], seed))
    return model
