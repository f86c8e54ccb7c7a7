"""``BasicAE``: the masked-view autoencoder LightningModule (reference src/autoencoder/autoencoder.py) on the HIP
hot path: 6-view gather + blanking, encoder, decoder and the MSE loss all run behind the C ABI.
"""
from argparse import ArgumentParser

import numpy as np
import torch

from . import ops
from .components import Decoder, Encoder
from .lightning import LightningModule, hparam


class BasicAE(LightningModule):
    def __init__(self, hparams=None):
        super().__init__()
        self.hidden_dim = hparam(hparams, "hidden_dim", 128)          # defaults: autoencoder.py:32-43
        self.latent_dim = hparam(hparams, "latent_dim", 128)
        self.input_width = hparam(hparams, "input_width", 306 * 6)
        self.input_height = hparam(hparams, "input_height", 256)
        self.output_width = hparam(hparams, "output_width", 306)
        self.output_height = hparam(hparams, "output_height", 256)
        self.batch_size = hparam(hparams, "batch_size", 16)
        self.in_channels = hparam(hparams, "in_channels", 3)
        self.hparams = hparams
        self.encoder = self.init_encoder(self.hidden_dim, self.latent_dim, self.in_channels, self.input_height,
                                         self.input_width)
        self.decoder = self.init_decoder(self.hidden_dim, self.latent_dim, self.in_channels, self.output_height,
                                         self.output_width)

    def init_decoder(self, hidden_dim, latent_dim, in_channels, output_height, output_width):
        return Decoder(hidden_dim, latent_dim, in_channels, output_height, output_width)

    def init_encoder(self, hidden_dim, latent_dim, in_channels, input_height, input_width):
        return Encoder(hidden_dim, latent_dim, in_channels, input_height, input_width)

    def six_to_one_task(self, x):
        """[B,6,3,H,W] -> (wide image with one of views 0..4 blanked, that view).  autoencoder.py:53-73.

        ``np.random.randint(0, 5)`` (exclusive bound: slot 5 is never chosen) is drawn on the host
        from numpy's global state exactly as the reference does; the gather, the copy of the target
        view and the blanking happen in one HIP kernel.
        """
        target_img_index = int(np.random.randint(0, 5))
        if ops.is_u8_frames(x):         # decoded frames [B,6,H,W,3]: ToTensor's /255 (a true division) inside the gather kernel
            wide4, y = ops.wide_image(x, "fp32", mask_slot=target_img_index, want_target=True)
            return ops.nhwc_to_nchw(wide4, 3), y
        _, wide, y = ops.stitch6(x.contiguous(), mask_slot=target_img_index, want_nhwc4=False, want_nchw=True,
                                 want_target=True)
        assert wide.size(-1) == 6 * x.size(-1)
        assert y.size(-1) == x.size(-1)
        return wide, y

    def forward(self, z, keeps=(None, None)):
        return self.decoder(z, keeps)                                   # autoencoder.py:75-76

    def _run_step(self, batch, batch_idx, step_name, keeps=None):
        """autoencoder.py:78-93: mask one view, encode, decode, ``mse_loss(y, y_hat)``."""
        keeps = keeps or {}
        target_img_index = int(np.random.randint(0, 5))
        wide4, y = ops.wide_image(batch, "fp32", mask_slot=target_img_index, want_target=True)      # fp32 views or uint8 frames
        z = self.encoder.forward_nhwc4(wide4, keeps.get("enc", (None, None)))
        y_hat = self(z, keeps.get("dec", (None, None)))
        if self.logger is not None and batch_idx % self.hparams.output_img_freq == 0:
            self._log_images(y, y_hat, step_name)
        return ops.MseLoss.apply(y_hat, y)

    def _log_images(self, y, y_hat, step_name, limit=1):
        exp = self.logger.experiment
        step = self.trainer.global_step if self.trainer is not None else 0
        exp.add_image(f"{step_name}_predicted_images", y_hat[:limit][0], step)
        exp.add_image(f"{step_name}_target_images", y[:limit][0], step)

    def training_step(self, batch, batch_idx):
        train_loss = self._run_step(batch, batch_idx, step_name="train")
        return {"loss": train_loss, "log": {"train_loss": train_loss}}

    def validation_step(self, batch, batch_idx):
        return {"val_loss": self._run_step(batch, batch_idx, step_name="valid")}

    def validation_epoch_end(self, outputs):
        avg_val_loss = torch.stack([x["val_loss"] for x in outputs]).mean()
        return {"val_loss": avg_val_loss, "log": {"avg_val_loss": avg_val_loss}}

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.hparams.learning_rate)     # autoencoder.py:119-120

    @staticmethod
    def add_model_specific_args(parent_parser):
        """Same flags and defaults as autoencoder.py:161-182 (plain argparse; test_tube is absent)."""
        p = ArgumentParser(parents=[parent_parser], add_help=False)
        p.add_argument("--hidden_dim", type=int, default=256)
        p.add_argument("--latent_dim", type=int, default=128)
        p.add_argument("--learning_rate", type=float, default=0.001)
        p.add_argument("--batch_size", type=int, default=16)
        p.add_argument("--input_width", type=int, default=306 * 6)
        p.add_argument("--input_height", type=int, default=256)
        p.add_argument("--output_width", type=int, default=306)
        p.add_argument("--output_height", type=int, default=256)
        p.add_argument("--in_channels", type=int, default=3)
        p.add_argument("--link", type=str, default="/scratch/ab8690/DLSP20Dataset/data")
        p.add_argument("--output_img_freq", type=int, default=500)
        return p
