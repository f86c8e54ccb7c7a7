#!/usr/bin/env python3
"""Per-kernel timing of the hot path on one MI355X (HIP events on the launch stream).

    python tools/bench_kernels.py [--batch 32] [--rows 16] [--iters 5]
Prints ms, achieved TFLOP/s against the fp32 MFMA peak (157.3 TF) for the matrix kernels and GB/s for the
byte movers.  Diagnostic tool; the contract benchmark is bench.py.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops, synth  # noqa: E402

PEAK_TF = 157.3


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--w", type=int, default=1836)
    ap.add_argument("--only", default="", help="comma-separated substrings of case names to run")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    b, h, w = a.batch, a.h, a.w
    ho, wo = ops.conv_out(h, 2), ops.conv_out(w, 2)
    res = {}
    x4 = torch.rand(b, h, w, 4, device=dev)
    x4[..., 3] = 0
    a1 = torch.rand(b, h, w, 32, device=dev)
    g = torch.randn(b, h, w, 32, device=dev)
    g3 = torch.randn(b, ho, wo, 32, device=dev)
    w1 = torch.randn(32, 3, 3, 3, device=dev) * 0.2
    w2 = torch.randn(32, 32, 3, 3, device=dev) * 0.06
    bias = torch.randn(32, device=dev) * 0.1
    d1, d2, d3 = (ops.conv_desc(b, h, w, 3, 1, a.rows), ops.conv_desc(b, h, w, 32, 1, a.rows),
                  ops.conv_desc(b, h, w, 32, 2, a.rows))
    px, pxo = b * h * w, b * ho * wo
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (b, h, w), device=dev, dtype=torch.int32)
    cases = [
        ("c1_fwd", lambda: ops.conv_fwd(x4, ops.conv_pack(w1, d1, 0), bias, d1), 2 * px * 32 * 27, px * (16 + 128)),
        ("c2_fwd", lambda: ops.conv_fwd_bits(a1, ops.conv_pack(w2, d2, 0), bias, d2), 2 * px * 32 * 288, px * 256),
        ("c3_fwd", lambda: ops.conv_fwd(a1, ops.conv_pack(w2, d3, 0), bias, d3), 2 * pxo * 32 * 288, px * 128 + pxo * 128),
        ("c2_dgrad", lambda: ops.conv_dgrad_bits(g, ops.conv_pack(w2, d2, 1), bits, d2), 2 * px * 32 * 288, px * 260),
        ("c3_dgrad", lambda: ops.conv_dgrad_bits(g3, ops.conv_pack(w2, d3, 2), bits, d3), 2 * pxo * 32 * 288, pxo * 128 + px * 132),
        ("c1_wgrad", lambda: ops.conv_wgrad(x4, g, d1), 2 * px * 32 * 27, px * (16 + 128)),
        ("c2_wgrad", lambda: ops.conv_wgrad(a1, g, d2), 2 * px * 32 * 288, px * 256),
        ("c2_wino2_fwd", lambda: ops.conv_wino2_fwd_bits(a1, ops.conv_wino2_pack(w2, d2, 0), bias, d2), 2 * px * 32 * 288, px * 260),
        ("c2_wino2_dgrad", lambda: ops.conv_wino2_dgrad_bits(g, ops.conv_wino2_pack(w2, d2, 1), bits, d2), 2 * px * 32 * 288, px * 260),
        ("c2_wino2_wgrad", lambda: ops.conv_wino2_wgrad(a1, g, d2), 2 * px * 32 * 288, px * 256),
        ("c2_wino_fwd", lambda: ops.conv_wino_fwd_bits(a1, ops.conv_wino_pack(w2, d2, 0), bias, d2), 2 * px * 32 * 288, px * 260),
        ("c2_wino_wgrad", lambda: ops.conv_wino_wgrad(a1, g, d2), 2 * px * 32 * 288, px * 256),
        ("c2_wino_dgrad", lambda: ops.conv_wino_dgrad_bits(g, ops.conv_wino_pack(w2, d2, 1), bits, d2), 2 * px * 32 * 288, px * 260),
        ("c3_wgrad", lambda: ops.conv_wgrad(a1, g3, d3), 2 * pxo * 32 * 288, px * 128 + pxo * 128),
    ]
    only = [t for t in a.only.split(",") if t]
    for name, fn, flops, nbytes in cases:
        if only and not any(t in name for t in only):
            continue
        ms = timeit(fn, a.iters)
        res[name] = {"ms": round(ms, 4), "TF": round(flops / ms / 1e9, 2), "frac_mfma": round(flops / ms / 1e9 / PEAK_TF, 3),
                     "GBs": round(nbytes / ms / 1e6, 1)}
        print(name, res[name], flush=True)
    if only:
        print(json.dumps(res))
        return
    feat = torch.relu(torch.randn(b, ho, wo, 32, device=dev))
    ms = timeit(lambda: ops.pool4_fwd(feat), a.iters)
    res["pool_fwd"] = {"ms": round(ms, 4), "GBs": round(feat.numel() * 5 / ms / 1e6, 1)}
    gp = torch.randn(b, 32 * ho * wo // 4, device=dev)
    ms = timeit(lambda: ops.pool4_relu_bwd(gp, feat), a.iters)
    res["pool_bwd"] = {"ms": round(ms, 4), "GBs": round(feat.numel() * 9 / ms / 1e6, 1)}
    views = torch.rand(b, 6, 3, h, w // 6, device=dev)
    ms = timeit(lambda: ops.stitch6(views), a.iters)
    res["stitch6"] = {"ms": round(ms, 4), "GBs": round(views.numel() * (4 + 16 / 3) / ms / 1e6, 1)}
    print(json.dumps(res))
    # torch (rocBLAS) skinny GEMMs at the model's shapes, for the decision whether to replace them
    pooled = torch.randn(b, 32 * ho * wo // 4, device=dev)
    wfc = torch.randn(128, pooled.shape[1], device=dev) * 0.01
    z = torch.randn(b, 64, device=dev)
    wh = torch.randn(640000, 64, device=dev) * 0.1
    dy = torch.randn(b, 128, device=dev)
    dl = torch.randn(b, 640000, device=dev)
    wfc_g, wh_g, zg = torch.empty_like(wfc), torch.empty_like(wh), torch.empty_like(z)
    ws = torch.empty(ops._lib.lib().dd_linear_workspace_bytes(b, 640000, 64), device=dev, dtype=torch.uint8)
    xw = torch.randn(b, pooled.shape[1], device=dev)
    for name, fn, nbytes in [
        ("fc1_fwd_hip", lambda: ops.Linear.apply(pooled, wfc, None), wfc.numel() * 4),
        ("fc1_dgrad_hip", lambda: ops._lib.check(ops._lib.lib().dd_linear_dgrad(ops._p(dy), ops._p(wfc), ops._p(xw), b, 128, wfc.shape[1], None, 0, ops._stream()), "dgrad"), wfc.numel() * 4),
        ("fc1_wgrad_hip", lambda: ops._lib.check(ops._lib.lib().dd_linear_wgrad(ops._p(dy), ops._p(pooled), ops._p(wfc_g), None, b, 128, wfc.shape[1], ops._stream()), "wgrad"), wfc.numel() * 4),
        ("head_fwd_hip", lambda: ops.Linear.apply(z, wh, None), wh.numel() * 4 + dl.numel() * 4),
        ("head_dgrad_hip", lambda: ops._lib.check(ops._lib.lib().dd_linear_dgrad(ops._p(dl), ops._p(wh), ops._p(zg), b, 640000, 64, ops._p(ws), ws.numel(), ops._stream()), "dgrad"), wh.numel() * 4 + dl.numel() * 4),
        ("head_wgrad_hip", lambda: ops._lib.check(ops._lib.lib().dd_linear_wgrad(ops._p(dl), ops._p(z), ops._p(wh_g), None, b, 640000, 64, ops._stream()), "wgrad"), wh.numel() * 4 + dl.numel() * 4),
        ("fc1_fwd_torch", lambda: torch.nn.functional.linear(pooled, wfc), wfc.numel() * 4),
        ("fc1_dgrad_torch", lambda: dy @ wfc, wfc.numel() * 4),
        ("fc1_wgrad_torch", lambda: dy.t() @ pooled, wfc.numel() * 4),
        ("head_fwd_torch", lambda: torch.nn.functional.linear(z, wh), wh.numel() * 4 + dl.numel() * 4),
        ("head_dgrad_torch", lambda: dl @ wh, wh.numel() * 4 + dl.numel() * 4),
        ("head_wgrad_torch", lambda: dl.t() @ z, wh.numel() * 4 + dl.numel() * 4),
    ]:
        ms = timeit(fn, a.iters)
        print(name, {"ms": round(ms, 4), "GBs": round(nbytes / ms / 1e6, 1)}, flush=True)


if __name__ == "__main__":
    main()
