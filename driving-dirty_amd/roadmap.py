"""``RoadMapBCE`` / ``RoadMap``: the road-map LightningModules on the HIP hot path.

Reference: src/roadmap_model/roadmap_bce_v2.py (registry name ``roadmap_bce``) and
src/roadmap_model/roadmap_pretrain_ae.py (``roadmap_mse``).  Same constructor, ``forward``,
``_run_step``, ``training_step``, ``validation_step``, ``validation_epoch_end``,
``configure_optimizers`` and ``state_dict`` keys (``ae.encoder.*``, ``fc1.*``).
"""
from argparse import ArgumentParser

import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .autoencoder import BasicAE
from .lightning import LightningModule, hparam, pretrained_ae


def compute_ts_road_map(road_map1, road_map2):
    """Threat score, reference src/utils/helper.py:74-77 (one fused pass on the device)."""
    return ops.threat_score(road_map1.contiguous(), road_map2.contiguous())


class RoadMapBCE(LightningModule):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.output_dim = 800 * 800
        # pretrained feature extractor (roadmap_bce_v2.py:43-47); ``pretrained_ae`` lets callers hand in
        # an already-built BasicAE when no checkpoint file exists (the reference's paths are NYU-local)
        self.ae = pretrained_ae(hparams)
        self.frozen = True
        self.ae.freeze()
        self.ae.decoder = None
        self.ae.encoder.precision = str(hparam(hparams, "precision", self.ae.encoder.precision))
        if self.ae.encoder.precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {self.ae.encoder.precision!r}")
        self.fc1 = nn.Linear(self.ae.latent_dim, self.output_dim)

    def wide_stitch_six_images(self, sample):
        """tuple of B [6,3,H,W] -> [B,3,H,6W] (NCHW), views re-ordered.  roadmap_bce_v2.py:53-64."""
        if ops.is_u8_frames(sample):    # decoded frames: ToTensor's /255 (a true division) inside the gather kernel
            return ops.nhwc_to_nchw(ops.wide_image(sample), 3)
        x = torch.stack(tuple(sample), dim=0) if isinstance(sample, (tuple, list)) else sample
        return ops.stitch6(x.contiguous(), want_nhwc4=False, want_nchw=True)[1]

    def _encode(self, sample, keeps=(None, None)):
        """The 6-view gather (+ NHWC, + ToTensor's /255 for uint8 frames, + the bf16 rounding) in one pass over whatever the data
        pipeline handed over -- ops.wide_image -- then the encoder."""
        wide4 = ops.wide_image(sample, self.ae.encoder.precision)
        return self.ae.encoder.forward_nhwc4(wide4, keeps)

    def _logits(self, x, keeps=(None, None)):
        representations = self._encode(x, keeps)
        y = ops.linear(representations, self.fc1.weight, self.fc1.bias)
        return y.reshape(y.size(0), 800, 800)

    def forward(self, x, keeps=(None, None)):
        """-> (logits [B,800,800], sigmoid(logits)).  roadmap_bce_v2.py:66-81."""
        y = self._logits(x, keeps)
        return y, ops.sigmoid(y.detach())

    def _run_step(self, batch, batch_idx, step_name, keeps=(None, None)):
        sample, target, road_image = batch
        logging = self.logger is not None and batch_idx % self.hparams.output_img_freq == 0
        per_sample = tuple(road_image)
        if (step_name == "train" and not logging and 0 < len(per_sample) <= 64 and
                all(t.is_cuda and t.is_contiguous() and t.dtype in (torch.bool, torch.uint8) and t.numel() % 4 == 0 for t in per_sample)):
            # the training step only needs the masks inside the loss: the kernel reads them where the collate left them
            # (a pointer table) instead of torch.stack-ing them first (roadmap_bce_v2.py:87)
            pred_rm = self._logits(sample, keeps)
            b = len(per_sample)
            loss, probs = ops.BceWithLogitsProbs.apply(pred_rm.reshape(b, -1), per_sample)
            return loss, None, pred_rm, probs.reshape(b, 800, 800)
        masks = torch.stack(per_sample, dim=0)            # bool as the dataset hands them over (data_helper.py:137-139)
        # self(sample) of the reference = (logits, probabilities); here the probabilities come out of the loss kernel's pass
        # over the logits (same values as forward()'s), which saves reading the 82 MB of logits a second time
        pred_rm = self._logits(sample, keeps)            # names as in the reference: pred_rm = logits, pred_logit_rm = probabilities
        # the loss reads the bool masks as bytes; the fp32 copy (roadmap_bce_v2.py:87) is only made where it is looked at
        target_rm = masks.float() if (logging or step_name != "train" or masks.dtype.is_floating_point) else masks
        batch_size = masks.size(0)
        loss_target = masks if masks.dtype in (torch.bool, torch.uint8) else target_rm
        loss, probs = ops.BceWithLogitsProbs.apply(pred_rm.reshape(batch_size, -1), loss_target.reshape(batch_size, -1).contiguous())
        pred_logit_rm = probs.reshape(batch_size, 800, 800)
        if logging:
            self._log_rm_images(self.wide_stitch_six_images(sample), target_rm, pred_logit_rm, step_name)
        return loss, target_rm, pred_rm, pred_logit_rm

    def _log_rm_images(self, x, target_rm, pred_rm, step_name, limit=1):
        exp = self.logger.experiment
        step = self.trainer.global_step if self.trainer is not None else 0
        exp.add_image(f"{step_name}_input_images", x[:limit][0], step)
        exp.add_image(f"{step_name}_target_roadmaps", target_rm[:limit], step)
        exp.add_image(f"{step_name}_pred_roadmaps", pred_rm[:limit].round(), step)

    def training_step(self, batch, batch_idx):
        if self.current_epoch >= self.hparams.unfreeze_epoch_no and self.frozen:
            self.frozen = False
            self.ae.unfreeze()
        train_loss, _, _, _ = self._run_step(batch, batch_idx, step_name="train")
        return {"loss": train_loss, "log": {"train_loss": train_loss}}

    def validation_step(self, batch, batch_idx):
        val_loss, target_rm, pred_rm, pred_logit_rm = self._run_step(batch, batch_idx, step_name="valid")
        val_ts = compute_ts_road_map(target_rm, pred_logit_rm)
        val_ts_rounded = ops.threat_score(target_rm.contiguous(), pred_logit_rm.contiguous(), round_b=True)
        return {"val_loss": val_loss, "val_ts_rounded": val_ts_rounded, "val_ts": val_ts}

    def validation_epoch_end(self, outputs):
        avg = {k: torch.stack([x[k] for x in outputs]).mean() for k in ("val_loss", "val_ts", "val_ts_rounded")}
        logs = {"avg_val_loss": avg["val_loss"], "avg_val_ts_rounded": avg["val_ts_rounded"], "avg_val_ts": avg["val_ts"]}
        return {"val_loss": avg["val_loss"], "log": logs}

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=self.hparams.learning_rate)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=10)
        return [optimizer], [scheduler]

    @staticmethod
    def add_model_specific_args(parent_parser):
        p = ArgumentParser(parents=[parent_parser], add_help=False)
        p.add_argument("--learning_rate", type=float, default=1e-3)
        p.add_argument("--unfreeze_epoch_no", type=int, default=0)
        p.add_argument("--batch_size", type=int, default=16)
        p.add_argument("--link", type=str, default="/scratch/ab8690/DLSP20Dataset/data")
        p.add_argument("--pretrained_path", type=str, default="")
        p.add_argument("--output_img_freq", type=int, default=500)
        return p


class RoadMap(RoadMapBCE):
    """MSE twin (roadmap_pretrain_ae.py): sigmoid inside ``forward``, ``mse_loss(target, pred)``, unfreeze at epoch 30."""

    def forward(self, x, keeps=(None, None)):
        """-> sigmoid(Linear(z)) [B,800,800], differentiable (roadmap_pretrain_ae.py:67-82)."""
        y = ops.Sigmoid.apply(ops.linear(self._encode(x, keeps), self.fc1.weight, self.fc1.bias))
        return y.reshape(y.size(0), 800, 800)

    def _run_step(self, batch, batch_idx, step_name, keeps=(None, None)):
        sample, target, road_image = batch
        target_rm = torch.stack(tuple(road_image), dim=0).float()
        pred_rm = self(sample, keeps)
        loss = ops.MseLoss.apply(pred_rm.contiguous(), target_rm)      # F.mse_loss(target_rm, pred_rm), roadmap_pretrain_ae.py:100
        return loss, target_rm, pred_rm

    def training_step(self, batch, batch_idx):
        if self.current_epoch >= 30 and self.frozen:          # roadmap_pretrain_ae.py:131
            self.frozen = False
            self.ae.unfreeze()
        train_loss, _, _ = self._run_step(batch, batch_idx, step_name="train")
        return {"loss": train_loss, "log": {"train_loss": train_loss}}

    def validation_step(self, batch, batch_idx):
        val_loss, target_rm, pred_rm = self._run_step(batch, batch_idx, step_name="valid")
        return {"val_loss": val_loss, "val_ts": compute_ts_road_map(target_rm, pred_rm),
                "val_ts_rounded": ops.threat_score(target_rm.contiguous(), pred_rm.contiguous(), round_b=True)}
