#!/usr/bin/env python3
"""ss_conv's data gradient (csrc/ssconv.hip) against torch's at the box head's full size, with timings (diagnostic).
DD_SSCONV_DGRAD=0 times the seven phase launches of the generic engine instead."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd.gconv import Layer, View  # noqa: E402
from tools.bench_gconv import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L = Layer(32, 32, (1, 24), stride=(1, 7))
    wt = torch.randn(32, 32, 1, 24, device=dev) * 0.05
    for b, h, xw in ((2, 5, 918), (3, 128, 311), (32, 128, 918)):
        gw = (xw - 24) // 7 + 1
        g = torch.randn(b, h, gw, 32, device=dev)
        dx = torch.full((b, h, xw, 32), float("nan"), device=dev)
        L.backward_data(wt, View(g), View(dx))
        ref = F.conv_transpose2d(g.permute(0, 3, 1, 2).double(), wt.double(), stride=(1, 7))       # [b, 32, h, 7(gw-1)+24]
        ref = F.pad(ref, (0, xw - ref.shape[3])).permute(0, 2, 3, 1)
        err = (dx.double() - ref).abs().max().item() / ref.abs().max().item()
        t = timeit(lambda: L.backward_data(wt, View(g), View(dx)), 10)
        print(f"b {b} h {h} xw {xw}: rel err {err:.2e}   {t:.3f} ms", flush=True)
        assert err < 2e-6


if __name__ == "__main__":
    main()
