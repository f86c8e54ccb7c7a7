#!/usr/bin/env python3
"""FC kernel timings at the roadmap model's shapes (fc1: 32 x 940032 -> 128, head: 32 x 64 -> 640000)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
b = 32
K1 = 32 * 128 * 918 // 4
pooled = torch.randn(b, K1, device=dev); wfc = torch.randn(128, K1, device=dev) * 0.01
z = torch.randn(b, 64, device=dev); wh = torch.randn(640000, 64, device=dev) * 0.1
dy = torch.randn(b, 128, device=dev); dl = torch.randn(b, 640000, device=dev)
wfc_g, wh_g, zg, xw = torch.empty_like(wfc), torch.empty_like(wh), torch.empty_like(z), torch.empty(b, K1, device=dev)
ws = torch.empty(ops._lib.lib().dd_linear_workspace_bytes(b, 640000, 64), device=dev, dtype=torch.uint8)
L = ops._lib.lib(); P = ops._p; S = ops._stream; chk = ops._lib.check
cases = [
    ("fc1_fwd", lambda: ops.Linear.apply(pooled, wfc, None), wfc.numel() * 4 + pooled.numel() * 4),
    ("fc1_dgrad", lambda: chk(L.dd_linear_dgrad(P(dy), P(wfc), P(xw), b, 128, K1, None, 0, S()), "d"), wfc.numel() * 4 + xw.numel() * 4),
    ("fc1_wgrad", lambda: chk(L.dd_linear_wgrad(P(dy), P(pooled), P(wfc_g), None, b, 128, K1, S()), "w"), wfc.numel() * 4 + pooled.numel() * 4),
    ("head_fwd", lambda: ops.Linear.apply(z, wh, None), wh.numel() * 4 + dl.numel() * 4),
    ("head_dgrad", lambda: chk(L.dd_linear_dgrad(P(dl), P(wh), P(zg), b, 640000, 64, P(ws), ws.numel(), S()), "d"), wh.numel() * 4 + dl.numel() * 4),
    ("head_wgrad", lambda: chk(L.dd_linear_wgrad(P(dl), P(z), P(wh_g), None, b, 640000, 64, S()), "w"), wh.numel() * 4 + dl.numel() * 4),
]
for _ in range(2):
    for name, fn, nbytes in cases:
        ms = timeit(fn, 7)
        print(name, {"ms": round(ms, 4), "GBs": round(nbytes / ms / 1e6, 1)}, flush=True)
