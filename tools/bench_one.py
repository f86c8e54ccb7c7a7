#!/usr/bin/env python3
"""Repeatable timing of the c2 kernels (warm clocks: each case is timed after a long warm-up of itself)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import _lib, ops
if os.environ.get("DD_AB_LIB"):      # A/B of two builds on one box (tools only): load another build of the library
    _lib.LIB = os.environ["DD_AB_LIB"]
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
b, h, w = 32, 256, 1836
a1 = torch.rand(b, h, w, 32, device=dev); g = torch.randn(b, h, w, 32, device=dev)
w2 = torch.randn(32, 32, 3, 3, device=dev) * 0.06; bias = torch.randn(32, device=dev) * 0.1
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (b, h, w), device=dev, dtype=torch.int32)
d2 = ops.conv_desc(b, h, w, 32, 1)
x4 = torch.rand(b, h, w, 4, device=dev)
cases = {"wino2_wgrad": lambda: ops.conv_wino2_wgrad(a1, g, d2), "wino_wgrad": lambda: ops.conv_wino_wgrad(a1, g, d2), "wino_fwd": lambda: ops.conv_wino_fwd_bits(a1, ops.conv_wino_pack(w2, d2, 0), bias, d2),
         "wino2_fwd": lambda: ops.conv_wino2_fwd_bits(a1, ops.conv_wino2_pack(w2, d2, 0), bias, d2),
         "wino2_dgrad": lambda: ops.conv_wino2_dgrad_bits(g, ops.conv_wino2_pack(w2, d2, 1), bits, d2),
         "wino2_dgrad_w1": lambda: ops.conv_wino2_dgrad_w1(g, ops.conv_wino2_pack(w2, d2, 1), bits, x4, d2),
         "wino_dgrad": lambda: ops.conv_wino_dgrad_bits(g, ops.conv_wino_pack(w2, d2, 1), bits, d2), "wgrad": lambda: ops.conv_wgrad(a1, g, d2)}
only = sys.argv[1].split(",") if len(sys.argv) > 1 else list(cases)
for name in only:
    fn = cases[name]
    for _ in range(10):
        fn()
    print(name, round(timeit(fn, 15), 4), "ms", flush=True)
