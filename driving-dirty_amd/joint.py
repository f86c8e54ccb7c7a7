"""``JointRoadMapBBox``: BASELINE.json config 4, the joint roadmap + bounding-box multi-task step.

The reference has no joint model: its ``c3_only`` switch (components.py:44-45) makes the two heads mutually
exclusive users of the encoder.  This is a build-side composition of rows a8 + a11 of SURVEY.md 8(a): ONE pass of
the encoder conv stack feeds both the latent path (pool -> dense blocks -> Linear(64, 640000) -> BCE-with-logits,
roadmap_bce_v2.py:66-108) and the box path (SpatialMappingCNN + RoadMapBoxesMergingCNN -> BCE on probabilities,
spatial_w_rm.py:67-131); the losses add, the two gradients of the shared c3 feature add inside the encoder's
backward.  Parity is checked per head against the oracle (tests/test_gpu_heads.py).
"""
import torch
from torch import nn

from . import ops
from .autoencoder import BasicAE
from .lightning import LightningModule, hparam, pretrained_ae
from .spatial import RoadMapBoxesMergingCNN, SpatialMappingCNN, bb_coord_to_map, per_sample_inputs


class JointRoadMapBBox(LightningModule):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.ae = pretrained_ae(hparams)
        self.ae.decoder = None
        self.ae.encoder.c3_only = False
        self.fc1 = nn.Linear(self.ae.latent_dim, 800 * 800)          # roadmap head, roadmap_bce_v2.py:50
        self.space_map_cnn = SpatialMappingCNN()                      # spatial_w_rm.py:50-52
        self.box_merge = RoadMapBoxesMergingCNN()
        precision = hparam(hparams, "precision", None)                # "fp32" (default) | "fp32x3": the box head's up-convs by split products
        if precision is not None:
            if precision not in ("fp32", "fp32x3"):
                raise ValueError(f"precision must be 'fp32' or 'fp32x3', got {precision!r}")
            self.box_merge.precision = precision

    def forward(self, x, rm):
        """x [B,6,3,256,306], rm [B,1,800,800] -> (roadmap logits [B,800,800], box probabilities [B,800,800]).  Both may also be
        the collate's tuples (per-sample views, bool road masks), read through pointer tables."""
        x = tuple(t.contiguous() for t in x) if isinstance(x, (tuple, list)) else x.contiguous()
        wide4 = ops.wide_image(x)                                     # fp32 views or uint8 frames, tensor or the collate's tuple
        feat, pooled = self.ae.encoder.conv_feature_and_pooled(wide4)
        # the box branch FIRST, the encoder's dense tail and the road-map head after it: autograd runs the newest nodes first, so
        # the two tensors that are 99.6 % of the data-parallel message (encoder fc1.fc1.weight 481 MB, head fc1.weight 164 MB) get
        # their gradients at the START of the backward and their reduction has the whole box-head backward (~40 ms) to hide under
        # instead of its last 5 (tools/step_phases.py); same arithmetic, same results
        boxes = self.box_merge(feat, self.space_map_cnn(x), rm).squeeze(1)
        z = self.ae.encoder._tail(pooled, (None, None))
        logits = ops.linear(z, self.fc1.weight, self.fc1.bias).reshape(-1, 800, 800)
        return logits, boxes

    def training_step(self, batch, batch_idx):
        sample, target, road_image = batch
        if per_sample_inputs(sample, road_image) and all(t.numel() % 4 == 0 for t in road_image) and len(road_image) <= 64:
            # roadmap_bce_v2.py:87 / spatial_w_rm.py:100-105 without the stacks: views and masks are read where the collate left them
            dev = sample[0].device
            b = len(sample)
            target_bb = bb_coord_to_map(target, dev).to(dev).float()
            logits, boxes = self(tuple(sample), tuple(road_image))
            loss_rm, _ = ops.BceWithLogitsProbs.apply(logits.reshape(b, -1), tuple(road_image))
        else:
            sample = torch.stack(tuple(sample), dim=0) if isinstance(sample, (tuple, list)) else sample
            target_rm = torch.stack(tuple(road_image), dim=0).float()
            target_bb = bb_coord_to_map(target, sample.device).to(sample.device).float()
            logits, boxes = self(sample, target_rm.unsqueeze(1))
            b = target_rm.size(0)
            loss_rm = ops.BceWithLogits.apply(logits.reshape(b, -1), target_rm.reshape(b, -1))
        loss_bb = ops.BceProbs.apply(boxes.reshape(b, -1), target_bb.reshape(b, -1))
        loss = loss_rm + loss_bb
        return {"loss": loss, "log": {"train_loss": loss, "roadmap_loss": loss_rm, "bbox_loss": loss_bb}}

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.hparams.learning_rate)
