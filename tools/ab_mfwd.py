#!/usr/bin/env python3
"""The multi-row gather kernel (csrc/dconv_m.hip) against torch's own convolutions at the box head's full sizes, with timings
(diagnostic).  Run once as it is and once with DD_DCONV_MFWD_OFF=1 for the A/B."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import gconv  # noqa: E402
from driving_dirty_amd.gconv import Layer, View  # noqa: E402
from tools.bench_gconv import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--check-batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-check", action="store_true", help="timings only (ablation builds: their results are wrong)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, cin, cout, hw in (("up2 64>32", 64, 32, 298), ("up3 32>16", 32, 16, 340)):
        L = Layer(cin, cout, 7, dil=7, transposed=True)
        oh, ow = L.out_hw(hw, hw)
        wt = torch.randn(cin, cout, 7, 7, device=dev) * 0.05
        bias = torch.randn(cout, device=dev) * 0.1
        for b, timed in ((a.check_batch, False), (a.batch, True)):
            if a.no_check and not timed:
                continue
            x = torch.randn(b, hw, hw, cin, device=dev)
            g = torch.randn(b, oh, ow, cout, device=dev)
            y = torch.full((b, oh, ow, cout), float("nan"), device=dev)
            dx = torch.full((b, hw, hw, cin), float("nan"), device=dev)
            fwd = lambda: L.forward(wt, bias, View(x, 0, cin), View(y, 0, cout), gconv.EPI_BIAS_RELU)      # noqa: E731
            dgr = lambda: L.backward_data(wt, View(g, 0, cout), View(dx, 0, cin), relu_src=x)               # noqa: E731
            if not timed:
                fwd()
                dgr()
                xr = x.permute(0, 3, 1, 2).double()
                ref = F.relu(F.conv_transpose2d(xr, wt.double(), bias.double(), dilation=7)).permute(0, 2, 3, 1)
                ey = (y.double() - ref).abs().max().item() / ref.abs().max().item()
                gr = g.permute(0, 3, 1, 2).double()
                dref = F.conv2d(gr, wt.double(), dilation=7).permute(0, 2, 3, 1) * (x > 0)
                ed = (dx.double() - dref).abs().max().item() / dref.abs().max().item()
                print(f"{name}: forward rel err {ey:.2e}   data gradient rel err {ed:.2e}", flush=True)
                assert ey < 2e-5 and ed < 2e-5
            else:
                gf = 2 * b * hw * hw * cin * cout * 49 / 1e9
                tf, td = timeit(fwd, a.iters), timeit(dgr, a.iters)
                print(f"{name} bs {b}: fwd {tf:7.3f} ms {gf / tf:6.1f} TF | dgrad {td:7.3f} ms {gf / td:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
