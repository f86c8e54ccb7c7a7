"""Spatial bounding-box heads on the HIP hot path: ``SpatialMappingCNN``, ``RoadMapBoxesMergingCNN``,
``BoxesMergingCNN`` (reference src/bounding_box_model/spatial_bb/components.py) and the ``BBSpatialRoadMap``
LightningModule (reference spatial_bb/spatial_w_rm.py, registry name ``spatial_rm``).  Same constructors,
parameter names, construction order (default init parity) and ``forward`` signatures; the modules only hold
parameters, the arithmetic is ``heads.SpatialMapFn`` / ``heads.MergeFn``.
"""
from argparse import ArgumentParser

import torch
from torch import nn

from . import gconv, ops
from .autoencoder import BasicAE
from .heads import MergeFn, SpatialMapFn, _ORDER, as_nhwc, road_map_taps
from .lightning import LightningModule, hparam, pretrained_ae


def _gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the hot path runs on MI355X only (got a {t.device} tensor)")


class SpatialMappingCNN(nn.Module):
    def __init__(self):
        super().__init__()
        self.f_conv = nn.Conv2d(3, 32, kernel_size=(52, 1), stride=(3, 2), padding=(1))
        self.fl_conv = nn.Conv2d(3, 32, kernel_size=(1, 50), stride=(3, 2))
        self.fr_conv = nn.Conv2d(3, 32, kernel_size=(1, 50), stride=(3, 2))
        self.b_conv = nn.Conv2d(3, 32, kernel_size=(52, 1), stride=(3, 2), padding=(1))
        self.bl_conv = nn.Conv2d(3, 32, kernel_size=(1, 50), stride=(3, 2))
        self.br_conv = nn.Conv2d(3, 32, kernel_size=(1, 50), stride=(3, 2))
        self.out_conv = nn.Conv2d(32, 32, kernel_size=(3, 3))

    def forward(self, x):
        """(b, 6, 3, 256, 306) -- or the collate's tuple of b [6,3,256,306] tensors, read through a pointer table instead of being
        stacked first (spatial_w_rm.py:100-103) -- -> (b, 32, 256, 256), returned as an NCHW-shaped view of the NHWC result."""
        per_sample = isinstance(x, (tuple, list))
        _gpu(x[0] if per_sample else x, "SpatialMappingCNN")
        params = []
        for n in _ORDER:
            m = getattr(self, n)
            params += [m.weight, m.bias]
        # uint8 frames ([b,6,256,306,3] or a tuple of [6,256,306,3]) are read as they are: ToTensor's /255 happens in the re-layout
        views = tuple(t.contiguous() for t in x) if per_sample else x.contiguous()
        return SpatialMapFn.apply(views, *params).permute(0, 3, 1, 2)


class _Merging(nn.Module):
    with_rm = False
    # "fp32": the reference's arithmetic on the exact-fp32 matrix kernels.  "fp32x3": the dilated up-convs take every fp32 product as six
    # bf16 x bf16 products of three-way split operands (csrc/dconv_split.hip; same 2e-5-of-peak bound against fp64 as the exact kernels in
    # the tests, error model in DESIGN.md 3.3d; 59 -> 43 ms per config-3 step).  None: the process-wide default (gconv.SPLIT_BF16).
    precision = None

    def _params(self, names):
        out = []
        for n in names:
            m = getattr(self, n)
            out += [m.weight, m.bias]
        return out

    def _run(self, ssr, spatial_map, rm):
        _gpu(ssr, type(self).__name__)
        names = ["ss_conv", "ss_deconv"] + (["rm_conv_1", "rm_conv_2"] if self.with_rm else []) + self.up_names
        rm4 = road_map_taps(rm) if self.with_rm else None
        if self.precision not in (None, "fp32", "fp32x3"):
            raise ValueError(f"{type(self).__name__}.precision must be 'fp32' or 'fp32x3', got {self.precision!r}")
        with gconv.split_products(None if self.precision is None else self.precision == "fp32x3"):
            probs = MergeFn.apply(as_nhwc(ssr, 32), as_nhwc(spatial_map, 32), rm4, self.with_rm, *self._params(names))
        return probs.unsqueeze(1)                        # [B,1,800,800] like the reference


class BoxesMergingCNN(_Merging):
    up_names = ["up_conv_1", "up_conv_2", "up_conv_3", "up_conv_4"]

    def __init__(self):
        super().__init__()
        self.ss_conv = nn.Conv2d(32, 32, kernel_size=(1, 24), stride=(1, 7))
        self.ss_deconv = nn.ConvTranspose2d(32, 32, kernel_size=2, stride=2)
        self.up_conv_1 = nn.ConvTranspose2d(64, 32, kernel_size=8, stride=1, dilation=8)
        self.up_conv_2 = nn.ConvTranspose2d(32, 16, kernel_size=8, stride=1, dilation=8)
        self.up_conv_3 = nn.ConvTranspose2d(16, 8, kernel_size=6, stride=1, dilation=6, output_padding=2)
        self.up_conv_4 = nn.ConvTranspose2d(8, 1, kernel_size=2, stride=2)

    def forward(self, ssr, spatial_map):
        return self._run(ssr, spatial_map, None)


class RoadMapBoxesMergingCNN(_Merging):
    with_rm = True
    up_names = ["up_conv_1", "up_conv_2", "up_conv_3", "up_conv_4", "up_conv_5"]

    def __init__(self):
        super().__init__()
        self.ss_conv = nn.Conv2d(32, 32, kernel_size=(1, 24), stride=(1, 7))
        self.ss_deconv = nn.ConvTranspose2d(32, 32, kernel_size=2, stride=2)
        self.rm_conv_1 = nn.Conv2d(1, 32, kernel_size=7, stride=3, dilation=3, padding=1)
        self.rm_conv_2 = nn.Conv2d(32, 32, kernel_size=3, stride=1, dilation=3)
        self.up_conv_1 = nn.ConvTranspose2d(96, 64, kernel_size=7, stride=1, dilation=7)
        self.up_conv_2 = nn.ConvTranspose2d(64, 32, kernel_size=7, stride=1, dilation=7)
        self.up_conv_3 = nn.ConvTranspose2d(32, 16, kernel_size=7, stride=1, dilation=7)
        self.up_conv_4 = nn.ConvTranspose2d(16, 8, kernel_size=7, stride=1, dilation=3)
        self.up_conv_5 = nn.ConvTranspose2d(8, 1, kernel_size=2, stride=2)

    def forward(self, ssr, spatial_map, rm):
        return self._run(ssr, spatial_map, rm)


def per_sample_inputs(sample, road_image):
    """True when the collate's tuples (helper.py:22-23) can be read where they lie: fp32 [6,3,H,W] views and bool / uint8 road
    masks, contiguous, on the GPU, at most 64 x k samples of one size."""
    if not (isinstance(sample, (tuple, list)) and isinstance(road_image, (tuple, list)) and 0 < len(sample) == len(road_image)):
        return False
    shape = tuple(sample[0].shape)
    dtype = sample[0].dtype                                   # fp32 [6,3,H,W] views or uint8 [6,H,W,3] decoded frames
    return (dtype in (torch.float32, torch.uint8) and
            all(t.is_cuda and t.is_contiguous() and t.dtype == dtype and tuple(t.shape) == shape and t.dim() == 4 for t in sample)
            and all(t.is_cuda and t.is_contiguous() and t.dtype in (torch.bool, torch.uint8) and t.dim() == 2 for t in road_image))


def bb_coord_to_map(target, device=None, rasterizer=None):
    """Targets -> [b,800,800] maps: pre-rasterised ``'bb_map'`` entries are taken as they are, a caller-supplied
    ``rasterizer`` is honoured, everything else goes through the HIP rasteriser in one launch."""
    if all("bb_map" in t for t in target):
        return torch.stack([t["bb_map"] for t in target], dim=0)
    if rasterizer is not None:
        return torch.stack([torch.as_tensor(rasterizer(t["bounding_box"])) for t in target], dim=0)
    return ops.boxes_to_binary_map([t["bounding_box"] for t in target], device)


class BBSpatialRoadMap(LightningModule):
    """spatial_w_rm.py:25-167.  ``bb_coord_to_map`` (the per-sample PIL polygon loop of src/utils/bb_to_img.py) runs
    as one launch of the HIP rasteriser over the batch's ``'bounding_box'`` tensors; batches may instead carry a
    pre-rasterised ``'bb_map'`` [800,800] tensor in each target dict, or name their own ``hparams.rasterizer``."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.output_dim = 800 * 800
        self.ae = pretrained_ae(hparams)
        self.frozen = True
        self.ae.freeze()
        self.ae.encoder.c3_only = True
        self.ae.decoder = None
        # hparams.precision = "bf16": the (frozen or fine-tuned) encoder conv stack on the bf16 matrix cores, its feature handed to
        # the fp32 heads (no reference counterpart: oracle/bf16_parts.py states the contract).  "fp32x3": everything fp32, the box
        # head's dilated up-convs by split products on the bf16 pipe (_Merging.precision).  Default "fp32": the reference's arithmetic.
        precision = str(hparam(hparams, "precision", self.ae.encoder.precision))
        if precision not in ("fp32", "bf16", "fp32x3"):
            raise ValueError(f"precision must be 'fp32', 'bf16' or 'fp32x3', got {precision!r}")
        self.ae.encoder.precision = "bf16" if precision == "bf16" else "fp32"
        self.space_map_cnn = SpatialMappingCNN()
        self.box_merge = RoadMapBoxesMergingCNN()
        if hparam(hparams, "precision", None) is not None:
            self.box_merge.precision = "fp32x3" if precision == "fp32x3" else "fp32"

    def wide_stitch_six_images(self, x):
        return ops.stitch6(x.contiguous(), want_nhwc4=False, want_nchw=True)[1]

    def forward(self, x, rm):
        """x [b,6,3,256,306], rm [b,1,800,800] -> [b,800,800].  spatial_w_rm.py:67-83.  Both may also be the collate's tuples
        (b x [6,3,256,306] views, b x bool [800,800] masks): the kernels then gather from the per-sample tensors."""
        space_rep = self.space_map_cnn(x)
        wide4 = ops.wide_image(x, self.ae.encoder.precision)      # fp32 views or uint8 frames, tensor or the collate's tuple
        ssr = self.ae.encoder.forward_nhwc4(wide4)
        yhat = self.box_merge(ssr, space_rep, rm)
        return yhat.squeeze(1)

    def bb_coord_to_map(self, target, device=None):
        """tuple of b target dicts -> [b,800,800].  spatial_w_rm.py:85-95."""
        return bb_coord_to_map(target, device, hparam(self.hparams, "rasterizer", None))

    def _run_step(self, batch, batch_idx, step_name):
        sample, target, road_image = batch
        if per_sample_inputs(sample, road_image):
            # the collate's tuples are read where they lie (pointer tables): no torch.stack of the 180 MB of views, no stack +
            # float() of the road masks (spatial_w_rm.py:100-105)
            dev = sample[0].device
            target_bb_img = self.bb_coord_to_map(target, dev).to(dev).float()
            pred_bb_img = self(tuple(sample), tuple(road_image))
        else:
            sample = torch.stack(tuple(sample), dim=0) if isinstance(sample, (tuple, list)) else sample
            target_bb_img = self.bb_coord_to_map(target, sample.device).to(sample.device)
            target_bb_img = target_bb_img.float() if sample.dtype == torch.uint8 else target_bb_img.type_as(sample)
            rm = torch.stack(tuple(road_image), dim=0).float().unsqueeze(1)
            pred_bb_img = self(sample, rm)
        batch_size = target_bb_img.size(0)
        target_bb_img = target_bb_img.reshape(batch_size, -1)
        pred_bb_img = pred_bb_img.reshape(batch_size, -1)
        if hparam(self.hparams, "mse_loss", False):
            loss = ops.MseLoss.apply(pred_bb_img, target_bb_img)
        else:
            loss = ops.BceProbs.apply(pred_bb_img, target_bb_img)
        return loss, target_bb_img, pred_bb_img

    def training_step(self, batch, batch_idx):
        if self.current_epoch >= self.hparams.unfreeze_epoch_no and self.frozen:
            self.frozen = False
            self.ae.unfreeze()
        train_loss, _, _ = self._run_step(batch, batch_idx, step_name="train")
        return {"loss": train_loss, "log": {"train_loss": train_loss}}

    def validation_step(self, batch, batch_idx):
        val_loss, _, _ = self._run_step(batch, batch_idx, step_name="valid")
        return {"val_loss": val_loss}

    def validation_epoch_end(self, outputs):
        avg_val_loss = torch.stack([x["val_loss"] for x in outputs]).mean()
        return {"val_loss": avg_val_loss, "log": {"avg_val_loss": avg_val_loss}}

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.hparams.learning_rate)

    @staticmethod
    def add_model_specific_args(parent_parser):
        p = ArgumentParser(parents=[parent_parser], add_help=False)
        p.add_argument("--learning_rate", type=float, default=1e-3)
        p.add_argument("--unfreeze_epoch_no", type=int, default=0)
        p.add_argument("--batch_size", type=int, default=16)
        p.add_argument("--mse_loss", action="store_true")
        p.add_argument("--link", type=str, default="/scratch/ab8690/DLSP20Dataset/data")
        p.add_argument("--pretrained_path", type=str, default="")
        p.add_argument("--output_img_freq", type=int, default=500)
        p.add_argument("--precision", type=str, default="fp32", choices=("fp32", "bf16", "fp32x3"),
                       help="fp32: the reference's arithmetic; bf16: encoder conv stack on the bf16 matrix cores; fp32x3: the box head's "
                            "dilated up-convs as six bf16 products per fp32 product (MI355X build only)")
        return p
