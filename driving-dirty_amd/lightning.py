"""A self-hosted slice of the PyTorch-Lightning 0.7.5 ``LightningModule`` surface.

The reference's models subclass ``pytorch_lightning.LightningModule`` (autoencoder.py:20,
roadmap_bce_v2.py:25, spatial_w_rm.py:25) and are driven by ``Trainer.fit``.  Neither package is
in this image (and 0.7.5 predates torch 2.x), so the handful of members the hot path touches are
provided here with the same names and meaning: ``hparams``, ``current_epoch``, ``logger``,
``trainer``, ``freeze()``, ``unfreeze()``, ``load_from_checkpoint()`` plus ``save_checkpoint()``.
Checkpoints are ``{'state_dict': ..., 'hparams': vars(hparams)}`` dicts, the layout Lightning
0.7.5 writes, so reference ``.ckpt`` files load.
"""
import weakref
from argparse import Namespace

import torch
from torch import nn

# objects with a ``refresh()`` that must hear about parameters becoming trainable: the data-parallel gradient hooks
# (ddp.GradSync) and the optimizer's early-step hooks (optim.HipAdam) can only be registered on tensors that require gradients
_UNFREEZE_LISTENERS = weakref.WeakSet()


def on_unfreeze(listener):
    _UNFREEZE_LISTENERS.add(listener)


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
        for listener in list(_UNFREEZE_LISTENERS):
            listener.refresh()

    # --- reference call sites: roadmap_bce_v2.py:43; spatial_w_rm.py:43
    @classmethod
    def load_from_checkpoint(cls, path, map_location=None):
        ckpt = torch.load(path, map_location=map_location or "cpu", weights_only=False)
        hp = ckpt.get("hparams", ckpt.get("hyper_parameters", {}))
        model = cls(Namespace(**hp) if isinstance(hp, dict) else hp)
        model.load_state_dict(ckpt["state_dict"])
        return model

    def save_checkpoint(self, path):
        """``{'state_dict', 'hparams', 'epoch'}`` as Lightning 0.7.5's ModelCheckpoint writes it.  ``hparams`` keeps the
        plain argparse values only: an in-memory ``pretrained_ae`` module or a ``rasterizer`` callable handed in through the
        Namespace is not a hyper-parameter (its weights are in the ``state_dict`` under ``ae.*``)."""
        hp = getattr(self, "hparams", None)
        plain = (bool, int, float, str, type(None), list, tuple, dict)
        hp = {k: v for k, v in vars(hp).items() if isinstance(v, plain)} if hp is not None else {}
        ae = getattr(self, "ae", None)
        if ae is not None and not hp.get("pretrained_path"):
            # the feature extractor was handed in as a module (no checkpoint file to name): keep what is needed to rebuild
            # its skeleton, the weights themselves are in the state_dict under 'ae.*'
            hp["ae_hparams"] = {k: getattr(ae, k) for k in ("hidden_dim", "latent_dim", "input_width", "input_height",
                                                           "output_width", "output_height", "in_channels")}
        torch.save({"state_dict": self.state_dict(), "hparams": hp, "epoch": self.current_epoch}, path)


def pretrained_ae(hparams):
    """The pretrained ``BasicAE`` of a fine-tuning module: ``BasicAE.load_from_checkpoint(hparams.pretrained_path)`` as in the
    reference (roadmap_bce_v2.py:43, spatial_w_rm.py:43), an in-memory module handed in as ``hparams.pretrained_ae``, or --
    when a checkpoint of THIS module is being re-loaded -- a skeleton from ``hparams.ae_hparams`` that the state_dict fills."""
    from .autoencoder import BasicAE
    pre = hparam(hparams, "pretrained_ae", None)
    if pre is not None:
        return pre
    path = hparam(hparams, "pretrained_path", "")
    if path:
        return BasicAE.load_from_checkpoint(path)
    skeleton = hparam(hparams, "ae_hparams", None)
    if skeleton:
        return BasicAE(Namespace(**skeleton))
    raise ValueError("no pretrained autoencoder: set hparams.pretrained_path (a BasicAE checkpoint) or hparams.pretrained_ae")


def hparam(hparams, name, default):
    """``hparams.x if hasattr(hparams, 'x') else default`` (autoencoder.py:32-43)."""
    return getattr(hparams, name) if hparams is not None and hasattr(hparams, name) else default
