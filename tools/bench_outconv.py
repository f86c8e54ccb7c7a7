#!/usr/bin/env python3
"""out_conv of SpatialMappingCNN (32 -> 32, k3, 258 x 258 -> 256 x 256, bs 32): the dilated-conv engine's three passes against the c2 layer's
Winograd F(2x2,3x3) kernels run as a padding-1 convolution on the 258 x 258 mosaic (the reference's outputs are its interior)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import gconv, ops  # noqa: E402
from driving_dirty_amd.gconv import Layer, View  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
b = int(os.environ.get("BATCH", "32"))
x = torch.relu(torch.randn(b, 258, 258, 32, device=dev))
w = torch.randn(32, 32, 3, 3, device=dev) * 0.06
bias = torch.randn(32, device=dev) * 0.1
L = Layer(32, 32, 3)
y = torch.empty(b, 256, 256, 32, device=dev)
g = torch.randn(b, 256, 256, 32, device=dev)
dx = torch.empty_like(x)
print("engine fwd   %.3f ms" % timeit(lambda: L.forward(w, bias, View(x), View(y), gconv.EPI_BIAS_RELU), 10))
print("engine dgrad %.3f ms" % timeit(lambda: L.backward_data(w, View(g), View(dx), relu_src=x), 10))
print("engine wgrad %.3f ms" % timeit(lambda: L.backward_weight(View(x), View(g)), 10))
d = ops.conv_desc(b, 258, 258, 32, 1)
pf, pd = ops.conv_wino2_pack(w, d, 0), ops.conv_wino2_pack(w, d, 1)
y1, bits = ops.conv_wino2_fwd_bits(x, pf, bias, d)
print("interior equals the engine's output: max |diff| %.2e" % float((y1[:, 1:257, 1:257] - y).abs().max()))
g258 = torch.zeros(b, 258, 258, 32, device=dev)
g258[:, 1:257, 1:257] = g
xb = torch.zeros(b, 258, 258, device=dev, dtype=torch.int32)
print("wino2 fwd    %.3f ms" % timeit(lambda: ops.conv_wino2_fwd_bits(x, pf, bias, d), 10))
print("wino2 dgrad  %.3f ms" % timeit(lambda: ops.conv_wino2_dgrad_bits(g258, pd, xb, d), 10))
print("wino2 wgrad  %.3f ms" % timeit(lambda: ops.conv_wino2_wgrad(x, g258, d), 10))
