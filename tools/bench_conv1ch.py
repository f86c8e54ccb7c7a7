#!/usr/bin/env python3
"""rm_conv_1 (Conv2d 1 -> 32, k7 s3 d3 p1 on the 800 x 800 road map, bs 32): forward and weight gradient of csrc/conv1ch.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
b = 32
rm4 = torch.rand(b, 268, 268, 4, device=dev)
w = torch.randn(32, 1, 7, 7, device=dev) * 0.1
bias = torch.zeros(32, device=dev)
y = ops.conv1ch_fwd(rm4, w, bias, relu=True)
g = torch.randn_like(y)
print("conv1ch fwd   %.4f ms" % timeit(lambda: ops.conv1ch_fwd(rm4, w, bias, relu=True), 10))
print("conv1ch wgrad %.4f ms" % timeit(lambda: ops.conv1ch_wgrad(rm4, g), 10))
