#!/usr/bin/env python3
"""Per-layer timing of the generic conv engine at the box head's real sizes (diagnostic)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import gconv  # noqa: E402
from driving_dirty_amd.gconv import Layer, View  # noqa: E402


def timeit(fn, iters=3):
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    b = a.batch
    layers = [
        ("up1 96>64 k7d7", Layer(96, 64, 7, dil=7, transposed=True), (256, 256)),
        ("up2 64>32 k7d7", Layer(64, 32, 7, dil=7, transposed=True), (298, 298)),
        ("up3 32>16 k7d7", Layer(32, 16, 7, dil=7, transposed=True), (340, 340)),
        ("up4 16>8 k7d3", Layer(16, 8, 7, dil=3, transposed=True), (382, 382)),
        ("rm2 32>32 k3d3", Layer(32, 32, 3, dil=3), (262, 262)),
        ("rm1 1>32 k7s3d3", Layer(1, 32, 7, stride=3, dil=3, pad=1), (800, 800)),
        ("ss_conv k1x24 s7", Layer(32, 32, (1, 24), stride=(1, 7)), (128, 918)),
        ("out_conv k3", Layer(32, 32, 3), (258, 258)),
        ("strip 1x50", Layer(3, 32, (1, 50), stride=(3, 2)), (256, 306)),
        ("dc1 64>32 k3p1", Layer(64, 32, 3, pad=1, transposed=True), (128, 153)),
    ]
    for name, L, (h, w) in layers:
        if a.only and a.only not in name:
            continue
        oh, ow = L.out_hw(h, w)
        cis, cos = (L.cin + 3) // 4 * 4, (L.cout + 3) // 4 * 4
        x = torch.rand(b, h, w, cis, device=dev)
        y = torch.empty(b, oh, ow, cos, device=dev)
        g = torch.randn(b, oh, ow, cos, device=dev)
        dx = torch.empty(b, h, w, cis, device=dev)
        wt = torch.randn((L.cin, L.cout) + L.k if L.transposed else (L.cout, L.cin) + L.k, device=dev) * 0.05
        bias = torch.zeros(L.cout, device=dev)
        macs = b * oh * ow * L.cout * L.cin * L.T if not L.transposed else b * h * w * L.cout * L.cin * L.T
        gf = 2 * macs / 1e9
        t_f = timeit(lambda: L.forward(wt, bias, View(x, 0, cis), View(y, 0, L.cout), gconv.EPI_BIAS_RELU))
        t_w = timeit(lambda: L.backward_weight(View(x, 0, cis), View(g, 0, L.cout)))
        t_d = timeit(lambda: L.backward_data(wt, View(g, 0, cos), View(dx, 0, L.cin), relu_src=x)) if L.cin >= 4 else float("nan")
        print(f"{name:18s} {gf:8.1f} GF | fwd {t_f:7.3f} ms {gf / t_f:6.1f} TF | wgrad {t_w:7.3f} ms {gf / t_w:6.1f} TF | "
              f"dgrad {t_d:7.3f} ms {gf / t_d:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
