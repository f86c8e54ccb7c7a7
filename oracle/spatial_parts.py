"""Oracle restatement of the spatial bounding-box heads (CPU, plain torch).

Follows reference ``src/bounding_box_model/spatial_bb/components.py``:
  * ``SpatialMapNet``        <- ``SpatialMappingCNN``        (components.py:6-77)
  * ``BoxMergeNet``          <- ``BoxesMergingCNN``          (components.py:80-119)
  * ``RoadBoxMergeNet``      <- ``RoadMapBoxesMergingCNN``   (components.py:122-170)

Same attribute names (hence ``state_dict`` keys) and the same construction order
(hence the same default init under a fixed seed).  Test infrastructure only.
"""
import torch
from torch import nn
from torch.nn import functional as F

from .branch import PLAIN


class SpatialMapNet(nn.Module):
    """Six per-view strip convs laid out as the car sees the road, then a 3x3 conv.

    Layout of the 258x258 mosaic (components.py:9-13,70-73):
        BL FL      views 3, 0   (raw)
        B  F       views 4, 1   (rot90)
        BR FR      views 5, 2   (flipped in H and W)
    """

    def __init__(self):
        super().__init__()
        # construction order = components.py:18-26
        self.f_conv = nn.Conv2d(3, 32, (52, 1), stride=(3, 2), padding=1)
        self.fl_conv = nn.Conv2d(3, 32, (1, 50), stride=(3, 2))
        self.fr_conv = nn.Conv2d(3, 32, (1, 50), stride=(3, 2))
        self.b_conv = nn.Conv2d(3, 32, (52, 1), stride=(3, 2), padding=1)
        self.bl_conv = nn.Conv2d(3, 32, (1, 50), stride=(3, 2))
        self.br_conv = nn.Conv2d(3, 32, (1, 50), stride=(3, 2))
        self.out_conv = nn.Conv2d(32, 32, 3)

    def forward(self, x, branch=PLAIN):
        """``branch`` (oracle.branch.Branch) records or replays the ReLU decisions; default: plain F.relu."""
        R = branch.relu
        v = [x[:, i] for i in range(6)]
        bl = R(self.bl_conv(v[3]), "bl")                                   # :34-35
        fl = R(self.fl_conv(v[0]), "fl")                                   # :37-38
        b = R(self.b_conv(torch.rot90(v[4], 1, [2, 3])), "b")              # :43-47
        f = R(self.f_conv(torch.rot90(v[1], 1, [3, 2])), "f")              # :49-52
        br = R(self.br_conv(torch.flip(v[5], [2, 3])), "br")               # :57-60
        fr = R(self.fr_conv(torch.flip(v[2], [2, 3])), "fr")               # :62-65
        rows = [torch.cat(p, dim=3) for p in ((bl, fl), (b, f), (br, fr))]
        return R(self.out_conv(torch.cat(rows, dim=2)), "out")             # :70-76


class BoxMergeNet(nn.Module):
    """Encoder feature + spatial map -> 800x800 box mask (no road-map input).  components.py:80-119."""

    def __init__(self):
        super().__init__()
        self.ss_conv = nn.Conv2d(32, 32, (1, 24), stride=(1, 7))
        self.ss_deconv = nn.ConvTranspose2d(32, 32, 2, stride=2)
        self.up_conv_1 = nn.ConvTranspose2d(64, 32, 8, dilation=8)
        self.up_conv_2 = nn.ConvTranspose2d(32, 16, 8, dilation=8)
        self.up_conv_3 = nn.ConvTranspose2d(16, 8, 6, dilation=6, output_padding=2)
        self.up_conv_4 = nn.ConvTranspose2d(8, 1, 2, stride=2)

    def forward(self, ssr, spatial_map, branch=PLAIN):
        R = branch.relu
        s = R(self.ss_deconv(R(self.ss_conv(ssr), "ss_conv")), "ss_deconv")
        h = torch.cat([s, spatial_map], dim=1)
        for i, layer in enumerate((self.up_conv_1, self.up_conv_2, self.up_conv_3)):
            h = R(layer(h), f"up{i + 1}")
        return torch.sigmoid(self.up_conv_4(h))


class RoadBoxMergeNet(nn.Module):
    """Encoder feature + spatial map + road map -> 800x800 box mask.  components.py:122-170."""

    def __init__(self):
        super().__init__()
        self.ss_conv = nn.Conv2d(32, 32, (1, 24), stride=(1, 7))
        self.ss_deconv = nn.ConvTranspose2d(32, 32, 2, stride=2)
        self.rm_conv_1 = nn.Conv2d(1, 32, 7, stride=3, dilation=3, padding=1)
        self.rm_conv_2 = nn.Conv2d(32, 32, 3, dilation=3)
        self.up_conv_1 = nn.ConvTranspose2d(96, 64, 7, dilation=7)
        self.up_conv_2 = nn.ConvTranspose2d(64, 32, 7, dilation=7)
        self.up_conv_3 = nn.ConvTranspose2d(32, 16, 7, dilation=7)
        self.up_conv_4 = nn.ConvTranspose2d(16, 8, 7, dilation=3)
        self.up_conv_5 = nn.ConvTranspose2d(8, 1, 2, stride=2)

    def forward(self, ssr, spatial_map, rm, branch=PLAIN):
        R = branch.relu
        s = R(self.ss_deconv(R(self.ss_conv(ssr), "ss_conv")), "ss_deconv")
        r = R(self.rm_conv_2(R(self.rm_conv_1(rm), "rm1")), "rm2")
        h = torch.cat([s, spatial_map, r], dim=1)                          # :159, 96 channels
        for i, layer in enumerate((self.up_conv_1, self.up_conv_2, self.up_conv_3, self.up_conv_4)):
            h = R(layer(h), f"up{i + 1}")
        return torch.sigmoid(self.up_conv_5(h))
