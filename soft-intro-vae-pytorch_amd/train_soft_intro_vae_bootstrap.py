"""MI355X-native drop-in for `soft_intro_vae_bootstrap/train_soft_intro_vae_bootstrap.py`.

Deltas over train_soft_intro_vae (reference lines in the bootstrap file):
  * a frozen `target_decoder` inside the model and its state_dict (:192-194), `decode_target` (:241-246),
    `forward(..., target=True)` (:196-216);
  * E-step reconstructions of rec/fake come from the target decoder (:593-594 via model(...));
  * D-step uses decode_target(z_rec) / decode_target(z_fake) WITHOUT detach and un-detached targets (:635-641);
  * decoder -> target_decoder weight copy every `copy_to_target_freq` epochs (:680-682); gamma_r defaults to 1.0.
All of it runs on the same HIP engine (sivae_hip.engine.SoftIntroEngine with bootstrap=True).
"""
import torch
import torch.nn as nn

import train_soft_intro_vae as _base
from sivae_hip import rng as _rng
from sivae_hip.engine import calc_kl, calc_reconstruction_loss, reparameterize  # noqa: F401
from sivae_hip.nn import Decoder, Encoder, ResidualBlock  # noqa: F401
from train_soft_intro_vae import (is_image_file, load_model, record_image, record_scalar,  # noqa: F401
                                  save_checkpoint, str_to_list)


class SoftIntroVAE(nn.Module):
    """reference: train_soft_intro_vae_bootstrap.py:178-246"""

    def __init__(self, cdim=3, zdim=512, channels=(64, 128, 256, 512, 512, 512), image_size=256, conditional=False,
                 cond_dim=10):
        super().__init__()
        self.zdim = zdim
        self.conditional = conditional
        self.cond_dim = cond_dim
        self.encoder = Encoder(cdim, zdim, channels, image_size, conditional=conditional, cond_dim=cond_dim)
        self.decoder = Decoder(cdim, zdim, channels, image_size, conditional=conditional,
                               conv_input_size=self.encoder.conv_output_size, cond_dim=cond_dim)
        # the target decoder is never trained; it lags the decoder by `copy_to_target_freq` epochs
        self.target_decoder = Decoder(cdim, zdim, channels, image_size, conditional=conditional,
                                      conv_input_size=self.encoder.conv_output_size, cond_dim=cond_dim)

    def forward(self, x, o_cond=None, deterministic=False, target=True):
        cond = o_cond if (self.conditional and o_cond is not None) else None
        mu, logvar = self.encode(x, o_cond=cond)
        z = mu if deterministic else reparameterize(mu, logvar)
        y = self.decode_target(z, y_cond=cond) if target else self.decode(z, y_cond=cond)
        return mu, logvar, z, y

    def sample(self, z, y_cond=None):
        return self.decode(z, y_cond=y_cond)

    def sample_with_noise(self, num_samples=1, device=torch.device("cpu"), y_cond=None):
        return self.decode(_rng.randn((num_samples, self.zdim), device), y_cond=y_cond)

    def encode(self, x, o_cond=None):
        if self.conditional and o_cond is not None:
            return self.encoder(x, o_cond=o_cond)
        return self.encoder(x)

    def decode(self, z, y_cond=None):
        if self.conditional and y_cond is not None:
            return self.decoder(z, y_cond=y_cond)
        return self.decoder(z)

    def decode_target(self, z, y_cond=None):
        if self.conditional and y_cond is not None:
            return self.target_decoder(z, y_cond=y_cond)
        return self.target_decoder(z)


def train_soft_intro_vae(dataset="cifar10", z_dim=128, lr_e=2e-4, lr_d=2e-4, batch_size=128, num_workers=4,
                         start_epoch=0, exit_on_negative_diff=False, copy_to_target_freq=1,
                         num_epochs=250, num_vae=0, save_interval=50, recon_loss_type="mse",
                         beta_kl=1.0, beta_rec=1.0, beta_neg=1.0, test_iter=1000, seed=-1, pretrained=None,
                         device=torch.device("cpu"), num_row=8, gamma_r=1.0, with_fid=False):
    """Same signature as the reference's bootstrap entry point (train_soft_intro_vae_bootstrap.py:360-364)."""
    return _base._train(dataset, z_dim, lr_e, lr_d, batch_size, num_workers, start_epoch, exit_on_negative_diff,
                        num_epochs, num_vae, save_interval, recon_loss_type, beta_kl, beta_rec, beta_neg, test_iter,
                        seed, pretrained, device, num_row, gamma_r, with_fid, bootstrap=True,
                        copy_to_target_freq=copy_to_target_freq, model_factory=SoftIntroVAE, tag="soft_intro_bootstrap")


if __name__ == "__main__":
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    try:
        train_soft_intro_vae(dataset="synthetic-cifar10", z_dim=128, batch_size=32, num_workers=0, num_epochs=1,
                             beta_kl=1.0, beta_neg=256, beta_rec=1.0, device=dev, test_iter=1000)
    except SystemError:
        print("Error, probably loss is NaN, try again...")
