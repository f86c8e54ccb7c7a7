#!/usr/bin/env python3
"""Diagnostic (needs an experiment build that writes per-stage cycle sums into the output: see DESIGN §3.1c)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import _lib, ops
_lib.LIB = os.environ["DD_AB_LIB"]
dev = torch.device("cuda:0")
b, h, w = 32, 256, 1836
a1 = torch.rand(b, h, w, 32, device=dev); g = torch.randn(b, h, w, 32, device=dev)
w2 = torch.randn(32, 32, 3, 3, device=dev) * 0.06; bias = torch.randn(32, device=dev) * 0.1
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (b, h, w), device=dev, dtype=torch.int32)
d2 = ops.conv_desc(b, h, w, 32, 1)
for name, fn in (("fwd", lambda: ops.conv_wino2_fwd_bits(a1, ops.conv_wino2_pack(w2, d2, 0), bias, d2)[0]),
                 ("dgrad", lambda: ops.conv_wino2_dgrad_bits(g, ops.conv_wino2_pack(w2, d2, 1), bits, d2))):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    v = y.reshape(-1)[:18].view(torch.int64).cpu().tolist()
    n = v[8]
    print(name, "steps", n, "cycles per stage:", [round(x / n) for x in v[:8]], "sum", round(sum(v[:8]) / n))
