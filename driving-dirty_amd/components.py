"""Encoder / Decoder / DenseBlock with the reference's constructor signatures, attribute names and
``state_dict`` keys (reference src/autoencoder/components.py), computed by the HIP hot path.

Drop-in contract kept from the reference:
  * ``Encoder(hidden_dim, latent_dim, in_channels, input_height, input_width)``, attribute ``c3_only``
    (components.py:11,31,44-45); ``forward(x[B,3,H,W]) -> z[B,latent]`` or the conv feature
    ``[B,32,H/2,W/2]`` (returned as an NCHW-shaped channels_last tensor: same values, NHWC memory);
  * ``DenseBlock(in_dim, out_dim, drop_p=0.2)``: Linear -> BatchNorm1d -> ReLU -> dropout that is
    ALWAYS active (components.py:108 omits ``training=``) with ``drop_p`` read at call time;
  * parameters are ordinary ``nn.Parameter``s in PyTorch layouts, created in the reference's order with
    the same RNG consumption, so the same seed gives the same initial weights and checkpoints load.
There is no CPU path: the forward raises if its input is not on a GPU or the HIP library is missing.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops

POOL = 4


def _require_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the hot path runs on MI355X only (got a {t.device} tensor); "
                           "the CPU restatement lives in oracle/ and is test infrastructure")


class DenseBlock(nn.Module):
    def __init__(self, in_dim, out_dim, drop_p=0.2):
        super().__init__()
        self.drop_p = drop_p
        self.fc1 = nn.Linear(in_dim, out_dim)
        self.fc_bn = nn.BatchNorm1d(out_dim)
        self.in_dim = in_dim

    def forward(self, x, keep=None):
        """``keep`` (optional 0/1 tensor) injects the dropout mask; otherwise one is drawn on the device."""
        _require_gpu(x, "DenseBlock")
        lin = ops.linear(x, self.fc1.weight, self.fc1.bias)
        bn = self.fc_bn
        p = float(self.drop_p)
        if keep is None and p > 0.0:
            keep = torch.empty_like(lin).bernoulli_(1.0 - p)      # one kernel (no probability tensor to fill first)
        training = bn.training or bn.running_mean is None
        nbt = bn.num_batches_tracked if (bn.training and bn.num_batches_tracked is not None) else None      # incremented by the kernel
        momentum = 0.1 if bn.momentum is None else bn.momentum
        scale = 1.0 / (1.0 - p) if p < 1.0 else 0.0
        return ops.BnReluDrop.apply(lin, bn.weight, bn.bias, bn.running_mean, bn.running_var, keep,
                                    training, bn.eps, momentum, scale, nbt)


class Decoder(nn.Module):
    """latent -> image: 2 DenseBlocks, view [B,64,h,w], 4 ConvTranspose2d (reference components.py:55-93)."""

    def __init__(self, hidden_dim, latent_dim, in_channels, output_height, output_width):
        super().__init__()
        if in_channels != 3:
            raise ValueError("the MI355X decoder is built for 3-channel images")
        # RNG parity with the reference's sizing dry run (components.py:75-83): one rand + four throw-away convs
        torch.rand(1, in_channels, output_height, output_width)
        nn.Conv2d(in_channels, 32, 1)
        nn.Conv2d(32, 32, 2, stride=2)
        nn.Conv2d(32, 32, 3, padding=1)
        nn.Conv2d(32, 64, 3, padding=1)
        self.deconv_dim_h = (output_height - 2) // 2 + 1
        self.deconv_dim_w = (output_width - 2) // 2 + 1
        self.latent_dim = latent_dim
        self.fc1 = DenseBlock(latent_dim, hidden_dim)
        self.fc2 = DenseBlock(hidden_dim, self.deconv_dim_h * self.deconv_dim_w * 64)
        self.dc1 = nn.ConvTranspose2d(64, 32, kernel_size=3, padding=1)
        self.dc2 = nn.ConvTranspose2d(32, 32, kernel_size=3, padding=1)
        self.dc3 = nn.ConvTranspose2d(32, 32, kernel_size=2, stride=2)
        self.dc4 = nn.ConvTranspose2d(32, in_channels, kernel_size=1, stride=1)

    def forward(self, z, keeps=(None, None)):
        from .heads import DecoderConvStack
        _require_gpu(z, "Decoder")
        h = self.fc2(self.fc1(z, keeps[0]), keeps[1])
        return DecoderConvStack.apply(h, self.deconv_dim_h, self.deconv_dim_w, self.dc1.weight, self.dc1.bias, self.dc2.weight,
                                      self.dc2.bias, self.dc3.weight, self.dc3.bias, self.dc4.weight, self.dc4.bias)


class Encoder(nn.Module):
    def __init__(self, hidden_dim, latent_dim, in_channels, input_height, input_width):
        super().__init__()
        if in_channels != 3:
            raise ValueError("the MI355X conv stack is built for 3-channel camera images")
        self.hidden_dim, self.latent_dim = hidden_dim, latent_dim
        self.input_height, self.input_width, self.in_channels = input_height, input_width, in_channels
        # nn.Conv2d modules only hold the parameters (names + default init); the arithmetic is ops.EncoderConvStack
        self.c1 = nn.Conv2d(in_channels, 32, kernel_size=3, padding=1)
        self.c2 = nn.Conv2d(32, 32, kernel_size=3, padding=1)
        self.c3 = nn.Conv2d(32, 32, kernel_size=3, stride=2, padding=1)
        self.pooling_size = POOL
        torch.rand(1, in_channels, input_height, input_width)   # RNG parity with the reference's sizing dry run (:34)
        ho, wo = ops.conv_out(input_height, 2), ops.conv_out(input_width, 2)
        self.fc1 = DenseBlock((32 * ho * wo) // POOL, hidden_dim)
        self.fc2 = DenseBlock(hidden_dim, hidden_dim)
        self.fc_z_out = nn.Linear(hidden_dim, latent_dim)
        self.c3_only = False
        self.rows_per_task = 0     # kernel tuning knob (never changes results)
        # "fp32" (the reference's arithmetic) or "bf16": conv stack on the bf16 matrix cores with fp32 accumulation,
        # activations stored in bf16 (BASELINE config 5); the FC tail and every parameter stay fp32 either way
        self.precision = "fp32"

    def _tail(self, pooled, keeps):
        """pooled -> DenseBlock fc1 -> DenseBlock fc2 -> fc_z_out.  After the big fc1 GEMM the rest is tiny: one fused launch
        each way where its sizes allow (ops.EncoderTail), the separate kernels otherwise."""
        fc1, fc2, fcz = self.fc1, self.fc2, self.fc_z_out
        m, h1, h2, l = pooled.shape[0], fc1.fc1.out_features, fc2.fc1.out_features, fcz.out_features
        bn1, bn2 = fc1.fc_bn, fc2.fc_bn
        same_mode = bn1.training == bn2.training and bn1.running_mean is not None and bn2.running_mean is not None
        if not (same_mode and ops.mlp_tail_supported(m, h1, h2, l)):
            h = fc2(fc1(pooled, keeps[0]), keeps[1])
            return ops.linear(h, fcz.weight, fcz.bias)
        lin1 = ops.linear(pooled, fc1.fc1.weight, fc1.fc1.bias)
        ks, scales = [], []
        for blk, keep, width in ((fc1, keeps[0], h1), (fc2, keeps[1], h2)):
            p = float(blk.drop_p)
            if keep is None and p > 0.0:      # F.dropout(training=True): always on (components.py:108)
                keep = torch.empty((m, width), device=pooled.device, dtype=torch.float32).bernoulli_(1.0 - p)
            ks.append(keep)
            scales.append(1.0 / (1.0 - p) if p < 1.0 else 0.0)
        return ops.EncoderTail.apply(lin1, bn1.weight, bn1.bias, fc2.fc1.weight, fc2.fc1.bias, bn2.weight, bn2.bias, fcz.weight,
                                     fcz.bias, ks[0], ks[1], bn1, bn2, scales[0], scales[1])

    def forward_nhwc4(self, x4, keeps=(None, None)):
        """x4: [B,H,W,4] NHWC image (channel 3 zero), e.g. straight from ``ops.stitch6`` (fp32) or
        ``ops_bf16.stitch6_bf16`` (bf16: selects the mixed-precision conv stack)."""
        if x4.dtype == torch.bfloat16:
            from . import ops_bf16
            if self.c3_only:      # the conv feature for the (fp32) box heads: NCHW-shaped view of an fp32 NHWC buffer
                return ops_bf16.encoder_conv_stack(x4, self.c1, self.c2, self.c3, pool=False).permute(0, 3, 1, 2)
            pooled = ops_bf16.encoder_conv_stack(x4, self.c1, self.c2, self.c3)
            return self._tail(pooled, keeps)
        if self.c3_only:
            feat = ops.encoder_conv_stack(x4, self.c1, self.c2, self.c3, False, self.rows_per_task)
            return feat.permute(0, 3, 1, 2)         # NCHW-shaped view of the NHWC buffer
        pooled = ops.encoder_conv_stack(x4, self.c1, self.c2, self.c3, True, self.rows_per_task)
        return self._tail(pooled, keeps)

    def conv_feature_and_pooled(self, x4):
        """The conv stack once, both exits: (conv feature [B,32,H/2,W/2] as an NCHW-shaped view, pooled vector for ``_tail``).  A
        caller that runs other work on the feature BEFORE ``_tail(pooled)`` gets the tail differentiated first in the backward
        (autograd runs the newest nodes first): see joint.JointRoadMapBBox.forward."""
        feat, pooled = ops.encoder_conv_stack(x4, self.c1, self.c2, self.c3, 2, self.rows_per_task)
        return feat.permute(0, 3, 1, 2), pooled

    def forward_both(self, x4, keeps=(None, None)):
        """One pass of the conv stack feeding BOTH exits the reference's ``c3_only`` switch chooses between
        (components.py:44-45): returns (conv feature [B,32,H/2,W/2], latent z).  Used by the joint roadmap + box model."""
        feat, pooled = ops.encoder_conv_stack(x4, self.c1, self.c2, self.c3, 2, self.rows_per_task)
        return feat.permute(0, 3, 1, 2), self._tail(pooled, keeps)

    def forward(self, x, keeps=(None, None)):
        _require_gpu(x, "Encoder")
        x4 = ops.nchw_to_nhwc(x.contiguous(), 4)
        if self.precision == "bf16":
            from . import ops_bf16
            x4 = ops_bf16.to_bf16(x4)
        return self.forward_nhwc4(x4, keeps)
