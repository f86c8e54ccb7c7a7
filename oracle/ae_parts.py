"""Oracle restatement of the autoencoder building blocks (CPU, plain torch).

Follows reference ``src/autoencoder/components.py``:
  * ``FcBlock``      <- ``DenseBlock``  (components.py:96-109)
  * ``EncoderNet``   <- ``Encoder``     (components.py:6-52)
  * ``DecoderNet``   <- ``Decoder``     (components.py:55-93)
and ``src/autoencoder/components_v2.py`` (``EncoderNetV2`` <- ``Encoder.forward`` :43-57, pinned by fixtures the class's own
forward produced on an instance assembled without the broken constructor; ``DecoderNetV2`` <- ``Decoder`` :59-98).

The classes register their parameters under the SAME attribute names as the
reference so ``state_dict()`` keys are interchangeable, and they draw from the
torch RNG in the same order during construction, so that
``torch.manual_seed(s); EncoderNet(...)`` reproduces the reference's default
initialisation bit for bit (the reference sizes its FC layers with dry-run
forwards on ``torch.rand`` inputs and, in the decoder, throw-away conv layers;
we size analytically but still consume the same random numbers).

Test infrastructure only -- see ``oracle/__init__.py``.
"""
import torch
from torch import nn
from torch.nn import functional as F

from .branch import PLAIN

POOL = 4          # components.py:23  (max_pool1d kernel over the NCHW-flattened vector)
CONV_CH = 32      # components.py:19-21


def conv_out_hw(h, w):
    """Spatial size after c1,c2 (k3 s1 p1: unchanged) and c3 (k3 s2 p1).  components.py:19-21."""
    return (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1


def pooled_len(h, w):
    """Length of the max_pool1d(4) output over the C*H*W flattened c3 feature.  components.py:33-38."""
    ho, wo = conv_out_hw(h, w)
    return (CONV_CH * ho * wo) // POOL


class FcBlock(nn.Module):
    """Linear -> BatchNorm1d -> ReLU -> dropout(p, ALWAYS active).  components.py:96-109.

    ``F.dropout(x, p)`` in the reference is called without ``training=`` so its
    default ``training=True`` applies even in eval mode (components.py:108).
    ``drop_p`` is read at call time, so tests set it to 0.0 for determinism.
    ``mask`` (optional) injects a pre-drawn keep mask (0/1) for bit-parity tests:
    out = x * mask / (1 - p).
    """

    def __init__(self, in_dim, out_dim, drop_p=0.2):
        super().__init__()
        self.drop_p = drop_p
        self.in_dim = in_dim
        self.fc1 = nn.Linear(in_dim, out_dim)
        self.fc_bn = nn.BatchNorm1d(out_dim)

    def forward(self, x, mask=None, branch=PLAIN, name="dense"):
        h = branch.relu(self.fc_bn(self.fc1(x)), name)
        if mask is not None:
            return h * mask / (1.0 - self.drop_p)
        return F.dropout(h, self.drop_p)


class EncoderNet(nn.Module):
    """3 convs (+ReLU) -> NCHW flatten -> max_pool1d(4) -> 2 FcBlocks -> Linear.  components.py:6-52."""

    def __init__(self, hidden_dim, latent_dim, in_channels, input_height, input_width):
        super().__init__()
        self.hidden_dim, self.latent_dim = hidden_dim, latent_dim
        self.in_channels, self.input_height, self.input_width = in_channels, input_height, input_width
        self.c1 = nn.Conv2d(in_channels, CONV_CH, 3, padding=1)
        self.c2 = nn.Conv2d(CONV_CH, CONV_CH, 3, padding=1)
        self.c3 = nn.Conv2d(CONV_CH, CONV_CH, 3, stride=2, padding=1)
        self.pooling_size = POOL
        # RNG parity with the reference's dry run (components.py:34): one rand of the input shape.
        torch.rand(1, in_channels, input_height, input_width)
        feat = pooled_len(input_height, input_width)
        self.fc1 = FcBlock(feat, hidden_dim)
        self.fc2 = FcBlock(hidden_dim, hidden_dim)
        self.fc_z_out = nn.Linear(hidden_dim, latent_dim)
        self.c3_only = False

    def conv_stack(self, x, branch=PLAIN):
        x = branch.relu(F.conv2d(x, self.c1.weight, self.c1.bias, padding=1), "relu1")
        x = branch.relu(F.conv2d(x, self.c2.weight, self.c2.bias, padding=1), "relu2")
        return branch.relu(F.conv2d(x, self.c3.weight, self.c3.bias, stride=2, padding=1), "relu3")

    def pool(self, feat, branch=PLAIN):
        # windows of 4 run over the C,H,W-flattened vector and may straddle image rows (W_out % 4 != 0)
        flat = feat.reshape(feat.size(0), 1, -1)
        return branch.max_pool1d(flat, POOL, "pool").squeeze(1)

    def forward(self, x, masks=(None, None), branch=PLAIN):
        """``branch`` (oracle.branch.Branch) records or replays the ReLU / max-pool decisions; default: plain F.relu."""
        feat = self.conv_stack(x, branch)
        if self.c3_only:                      # components.py:44-45
            return feat
        return self.tail(feat, masks, branch)

    def tail(self, feat, masks=(None, None), branch=PLAIN):
        """conv feature -> pool -> two FcBlocks -> Linear (components.py:46-52)."""
        h = self.fc1(self.pool(feat, branch), masks[0], branch, "fc1")
        h = self.fc2(h, masks[1], branch, "fc2")
        return self.fc_z_out(h)


class DecoderNet(nn.Module):
    """2 FcBlocks -> view [B,64,h,w] -> 4 ConvTranspose2d (ReLU after the first three).  components.py:55-93."""

    def __init__(self, hidden_dim, latent_dim, in_channels, output_height, output_width):
        super().__init__()
        # RNG parity with components.py:75-83: rand input, then four throw-away convs
        torch.rand(1, in_channels, output_height, output_width)
        nn.Conv2d(in_channels, 32, 1)
        nn.Conv2d(32, 32, 2, stride=2)
        nn.Conv2d(32, 32, 3, padding=1)
        nn.Conv2d(32, 64, 3, padding=1)
        # k1 s1 -> k2 s2 -> k3 p1 -> k3 p1 : only the k2 s2 layer changes the size
        self.deconv_dim_h = (output_height - 2) // 2 + 1
        self.deconv_dim_w = (output_width - 2) // 2 + 1
        self.latent_dim = latent_dim
        self.fc1 = FcBlock(latent_dim, hidden_dim)
        self.fc2 = FcBlock(hidden_dim, self.deconv_dim_h * self.deconv_dim_w * 64)
        self.dc1 = nn.ConvTranspose2d(64, 32, 3, padding=1)
        self.dc2 = nn.ConvTranspose2d(32, 32, 3, padding=1)
        self.dc3 = nn.ConvTranspose2d(32, 32, 2, stride=2)
        self.dc4 = nn.ConvTranspose2d(32, in_channels, 1)

    def forward(self, z, masks=(None, None), branch=PLAIN):
        """``branch`` (oracle.branch.Branch) records or replays the five ReLU decisions; default: plain F.relu."""
        h = self.fc2(self.fc1(z, masks[0], branch, "d_fc1"), masks[1], branch, "d_fc2")
        h = h.reshape(h.size(0), 64, self.deconv_dim_h, self.deconv_dim_w)
        h = branch.relu(F.conv_transpose2d(h, self.dc1.weight, self.dc1.bias, padding=1), "dc1")
        h = branch.relu(F.conv_transpose2d(h, self.dc2.weight, self.dc2.bias, padding=1), "dc2")
        h = branch.relu(F.conv_transpose2d(h, self.dc3.weight, self.dc3.bias, stride=2), "dc3")
        return F.conv_transpose2d(h, self.dc4.weight, self.dc4.bias)      # no activation (components.py:92)


class DecoderNetV2(nn.Module):
    """ConvTranspose2d -> BatchNorm2d -> ReLU decoder, reference src/autoencoder/components_v2.py:59-98 (this class of
    the v2 file constructs; pinned by tests/golden/tiny_decoder_v2.npz, generated from it)."""

    def __init__(self, hidden_dim, latent_dim, in_channels, output_height, output_width):
        super().__init__()
        # RNG parity with components_v2.py:80-87: rand input, then four throw-away convs
        torch.rand(1, in_channels, output_height, output_width)
        nn.Conv2d(in_channels, 32, 1)
        nn.Conv2d(32, 32, 2, stride=2)
        nn.Conv2d(32, 32, 3, padding=1)
        nn.Conv2d(32, 64, 3, padding=1)
        self.deconv_dim_h = (output_height - 2) // 2 + 1
        self.deconv_dim_w = (output_width - 2) // 2 + 1
        self.latent_dim = latent_dim
        self.fc1 = FcBlock(latent_dim, hidden_dim)
        self.fc2 = FcBlock(hidden_dim, self.deconv_dim_h * self.deconv_dim_w * 64)
        self.dc1 = nn.ConvTranspose2d(64, 32, 3, padding=1)        # construction order = components_v2.py:71-78
        self.bn1 = nn.BatchNorm2d(32)
        self.dc2 = nn.ConvTranspose2d(32, 32, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(32)
        self.dc3 = nn.ConvTranspose2d(32, 32, 2, stride=2)
        self.bn3 = nn.BatchNorm2d(32)
        self.dc4 = nn.ConvTranspose2d(32, in_channels, 1)

    def forward(self, z, masks=(None, None)):
        h = self.fc2(self.fc1(z, masks[0]), masks[1])
        h = h.reshape(h.size(0), 64, self.deconv_dim_h, self.deconv_dim_w)
        h = F.relu(self.bn1(F.conv_transpose2d(h, self.dc1.weight, self.dc1.bias, padding=1)))
        h = F.relu(self.bn2(F.conv_transpose2d(h, self.dc2.weight, self.dc2.bias, padding=1)))
        h = F.relu(self.bn3(F.conv_transpose2d(h, self.dc3.weight, self.dc3.bias, stride=2)))
        return F.conv_transpose2d(h, self.dc4.weight, self.dc4.bias)      # no activation (components_v2.py:99)


class EncoderNetV2(nn.Module):
    """Conv -> BatchNorm2d -> ReLU variant, reference src/autoencoder/components_v2.py:6-57.

    The class's constructor raises (``self.bn3 = nn.Conv2d(32)``, components_v2.py:24), but only that line is broken: its
    ``forward`` (:43-57) and ``_calculate_output_dim`` (:36-41) run on an instance assembled without ``__init__`` with
    ``bn3 = BatchNorm2d(32)``, the one stated interpretation.  tests/golden/make_golden.py does exactly that and commits the
    outputs (``tiny_encoder_v2.npz``, ``full_encoder_v2.npz``); tests/test_oracle_golden.py holds this restatement to them
    (fp32 2e-6, fp64 1e-12 / 1e-10 at 256 x 1836).  PINNED.
    """

    def __init__(self, hidden_dim, latent_dim, in_channels, input_height, input_width):
        super().__init__()
        self.c1 = nn.Conv2d(in_channels, CONV_CH, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(CONV_CH)
        self.c2 = nn.Conv2d(CONV_CH, CONV_CH, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(CONV_CH)
        self.c3 = nn.Conv2d(CONV_CH, CONV_CH, 3, stride=2, padding=1)
        self.bn3 = nn.BatchNorm2d(CONV_CH)
        torch.rand(1, in_channels, input_height, input_width)
        self.fc1 = FcBlock(pooled_len(input_height, input_width), hidden_dim)
        self.fc2 = FcBlock(hidden_dim, hidden_dim)
        self.fc_z_out = nn.Linear(hidden_dim, latent_dim)
        self.c3_only = False

    def _bn(self, bn, x):
        return bn(x)      # nn.BatchNorm2d: batch statistics + running-stat update in train mode, running stats in eval

    def forward(self, x, masks=(None, None), branch=PLAIN):
        """``branch`` (oracle.branch) records / replays / censuses the ReLU and max-pool decisions; default: plain F.relu."""
        x = branch.relu(self._bn(self.bn1, F.conv2d(x, self.c1.weight, self.c1.bias, padding=1)), "relu1")
        x = branch.relu(self._bn(self.bn2, F.conv2d(x, self.c2.weight, self.c2.bias, padding=1)), "relu2")
        x = branch.relu(self._bn(self.bn3, F.conv2d(x, self.c3.weight, self.c3.bias, stride=2, padding=1)), "relu3")
        if self.c3_only:
            return x
        flat = x.reshape(x.size(0), 1, -1)
        h = self.fc1(branch.max_pool1d(flat, POOL, "pool").squeeze(1), masks[0], branch, "fc1")
        return self.fc_z_out(self.fc2(h, masks[1], branch, "fc2"))
