#!/usr/bin/env python3
"""Which weight gradient is closer to fp64 at a realistic size: the exact fp32 MFMA kernel or the split-product one?
up_conv_1 (96 -> 64, k7 d7) at 256 x 256, batch 4: fp64 torch on the device as the truth."""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import gconv  # noqa: E402

dev = torch.device("cuda:0")
for cin, cout, hw in ((96, 64, 256), (64, 32, 298)):
    b = 4
    g0 = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand(b, hw, hw, cin, device=dev, generator=g0)
    layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
    oh, ow = layer.out_hw(hw, hw)
    g = torch.randn(b, oh, ow, cout, device=dev, generator=g0)
    mod = nn.ConvTranspose2d(cin, cout, 7, dilation=7).to(dev).double()
    xd = x.permute(0, 3, 1, 2).double().requires_grad_(False)
    y = mod(xd)
    y.backward(g.permute(0, 3, 1, 2).double())
    ref = mod.weight.grad
    out = {}
    for split in (False, True):
        gconv.SPLIT_BF16 = split
        dw, _ = layer.backward_weight(gconv.View(x), gconv.View(g))
        out[split] = float((dw.double() - ref).abs().max() / ref.abs().max())
    gconv.SPLIT_BF16 = False
    print(f"{cin}->{cout} at {hw}x{hw}, batch {b}: |dW - fp64| / peak: exact kernel {out[False]:.3e}, split kernel {out[True]:.3e}")
