"""A self-hosted slice of the PyTorch-Lightning 0.7.5 ``LightningModule`` surface.

The reference's models subclass ``pytorch_lightning.LightningModule`` (autoencoder.py:20,
roadmap_bce_v2.py:25, spatial_w_rm.py:25) and are driven by ``Trainer.fit``.  Neither package is
in this image (and 0.7.5 predates torch 2.x), so the handful of members the hot path touches are
provided here with the same names and meaning: ``hparams``, ``current_epoch``, ``logger``,
``trainer``, ``freeze()``, ``unfreeze()``, ``load_from_checkpoint()`` plus ``save_checkpoint()``.
Checkpoints are ``{'state_dict': ..., 'hparams': vars(hparams)}`` dicts, the layout Lightning
0.7.5 writes, so reference ``.ckpt`` files load.
"""
from argparse import Namespace

import torch
from torch import nn


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.current_epoch = 0
        self.logger = None
        self.trainer = None

    # --- reference call sites: roadmap_bce_v2.py:46,129; spatial_w_rm.py:46,150
    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad = True
        self.train()

    # --- reference call sites: roadmap_bce_v2.py:43; spatial_w_rm.py:43
    @classmethod
    def load_from_checkpoint(cls, path, map_location=None):
        ckpt = torch.load(path, map_location=map_location or "cpu", weights_only=False)
        hp = ckpt.get("hparams", ckpt.get("hyper_parameters", {}))
        model = cls(Namespace(**hp) if isinstance(hp, dict) else hp)
        model.load_state_dict(ckpt["state_dict"])
        return model

    def save_checkpoint(self, path):
        hp = getattr(self, "hparams", None)
        torch.save({"state_dict": self.state_dict(), "hparams": dict(vars(hp)) if hp is not None else {},
                    "epoch": self.current_epoch}, path)


def hparam(hparams, name, default):
    """``hparams.x if hasattr(hparams, 'x') else default`` (autoencoder.py:32-43)."""
    return getattr(hparams, name) if hparams is not None and hasattr(hparams, name) else default
