"""``BasicAE``: the masked-view autoencoder LightningModule (reference src/autoencoder/autoencoder.py).

Round-1 scope: the encoder half runs on the HIP hot path; the decoder (SURVEY.md section 8f row 1:
DenseBlock 128 -> 1,253,376 and the ConvTranspose2d stack) is not built yet, so ``forward`` /
``training_step`` of the AE pre-training task raise ``NotImplementedError`` instead of falling back to
another backend.  ``six_to_one_task`` and the encoder are complete and used by the roadmap model.
"""
from argparse import ArgumentParser

import numpy as np
import torch

from . import ops
from .components import Encoder
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
        self.decoder = None     # next row of SURVEY.md section 8(f)

    def init_encoder(self, hidden_dim, latent_dim, in_channels, input_height, input_width):
        return Encoder(hidden_dim, latent_dim, in_channels, input_height, input_width)

    def six_to_one_task(self, x):
        """[B,6,3,H,W] -> (wide image with one of views 0..4 blanked, that view).  autoencoder.py:53-73.

        ``np.random.randint(0, 5)`` (exclusive bound: slot 5 is never chosen) is drawn on the host
        from numpy's global state exactly as the reference does; the gather, the copy of the target
        view and the blanking happen in one HIP kernel.
        """
        target_img_index = int(np.random.randint(0, 5))
        _, wide, y = ops.stitch6(x.contiguous(), mask_slot=target_img_index, want_nhwc4=False, want_nchw=True,
                                 want_target=True)
        assert wide.size(-1) == 6 * x.size(-1)
        assert y.size(-1) == x.size(-1)
        return wide, y

    def forward(self, z):
        raise NotImplementedError("BasicAE.forward = decoder(z): the decoder is not built yet (SURVEY.md 8f row 1)")

    def training_step(self, batch, batch_idx):
        raise NotImplementedError("AE pre-training needs the decoder (SURVEY.md 8f row 1)")

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
