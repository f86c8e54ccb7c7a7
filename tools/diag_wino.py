#!/usr/bin/env python3
"""Diagnostic: Winograd vs direct c2 kernels on one shape.  python tools/diag_wino.py B H W"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops, synth
b, h, w = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = synth.hash_uniform((b, h, w, 32), 1, 0.0, 1.0).to(dev)
x[:, :, w // 3: w // 2] = 0
wt = synth.hash_uniform((32, 32, 3, 3), 2, -0.3, 0.3).to(dev)
bias = synth.hash_uniform((32,), 3, -0.2, 0.2).to(dev)
d = ops.conv_desc(b, h, w, 32, 1)
y0, s0 = ops.conv_fwd_bits(x, ops.conv_pack(wt, d, 0), bias, d)
y1, s1 = ops.conv_wino_fwd_bits(x, ops.conv_wino_pack(wt, d, 0), bias, d)
print("fwd max abs diff", float((y0 - y1).abs().max()), "bits equal", bool(torch.equal(s0, s1)), "mismatch px", int((s0 != s1).sum()))
g = synth.hash_uniform((b, h, w, 32), 4).to(dev)
dx0 = ops.conv_dgrad_bits(g, ops.conv_pack(wt, d, 1), s0, d)
dx1 = ops.conv_wino_dgrad_bits(g, ops.conv_wino_pack(wt, d, 1), s0, d)
diff = (dx0 - dx1).abs()
print("dgrad max abs diff", float(diff.max()), "ref max", float(dx0.abs().max()))
if float(diff.max()) > 1e-3:
    idx = (diff > 1e-3).nonzero()
    print("bad count", idx.shape[0], "first", idx[:8].tolist(), "cols", sorted(set(idx[:, 2].tolist()))[:40])
