#!/usr/bin/env python3
"""dd_adam_step alone on the encoder fc1 weight (481 MB: 3.37 GB of traffic per pass).  DD_AB_LIB: another build of the library."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import _lib, ops
if os.environ.get("DD_AB_LIB"):
    _lib.LIB = os.environ["DD_AB_LIB"]
from tools.bench_kernels import timeit
n = 940032 * 128
dev = torch.device("cuda:0")
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 0.01; m = torch.zeros_like(p); v = torch.zeros_like(p)
fn = lambda: ops.adam_step_flat(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 3, 1.0)
for _ in range(5):
    fn()
ms = timeit(fn, 20)
print(f"adam {n} elements: {ms:.4f} ms = {7 * 4 * n / ms / 1e9:.2f} TB/s")
