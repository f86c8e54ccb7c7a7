#!/usr/bin/env python3
"""A/B of the box heads' up-conv forwards at bs 32: exact fp32 MFMA kernel vs the bf16 x 3 split-product kernel (csrc/dconv_split.hip).

    python tools/bench_split.py [--batch 32] [--iters 10]
Prints per layer: exact ms, split ms (kernel alone, and with the input split + weight pack passes), max |difference| / peak.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import gconv  # noqa: E402


def timed(fn, iters):
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
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import ctypes as C
    from driving_dirty_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    res = {}
    for name, cin, cout, hw in (("up_conv_1", 96, 64, 256), ("up_conv_2", 64, 32, 298)):
        g = torch.Generator(device=dev).manual_seed(1)
        layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
        oh, ow = layer.out_hw(hw, hw)
        x = torch.rand(a.batch, hw, hw, cin, device=dev, generator=g)
        w = (torch.rand(cin, cout, 7, 7, device=dev, generator=g) - 0.5) * 0.05
        bias = torch.rand(cout, device=dev, generator=g) - 0.5
        y0 = torch.empty(a.batch, oh, ow, cout, device=dev)
        y1 = torch.empty_like(y0)

        def run(split, y):
            gconv.SPLIT_BF16 = split
            layer.forward(w, bias, gconv.View(x), gconv.View(y), gconv.EPI_BIAS_RELU)
        t_exact = timed(lambda: run(False, y0), a.iters)
        t_split_all = timed(lambda: run(True, y1), a.iters)
        # the kernel alone: operands split once
        d = gconv._desc(a.batch, gconv.View(x), gconv.View(y1), cin, cout, (7, 7), (1, 1), (7, 7), (42, 42))
        xs = torch.empty(lib.dd_dconv_split_input_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
        pk = torch.empty(lib.dd_dconv_split_packed_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
        t_in = timed(lambda: _lib.check(lib.dd_dconv_split_input(P(x), P(xs), C.byref(d), st), "in"), a.iters)
        t_pk = timed(lambda: _lib.check(lib.dd_dconv_split_pack(P(w), P(pk), C.byref(d), 0, 49, cout * 49, 1, cout, cin, st), "pk"), a.iters)
        t_k = timed(lambda: _lib.check(lib.dd_dconv_fwd_split(P(xs), P(pk), P(bias), None, P(y1), None, C.byref(d), gconv.EPI_BIAS_RELU, st), "k"), a.iters)
        # the data gradient: dx = dilated conv of dy (ReLU mask of the producer fused)
        g = torch.randn(a.batch, oh, ow, cout, device=dev, generator=g0) if False else torch.rand(a.batch, oh, ow, cout, device=dev) - 0.5
        dx0 = torch.empty(a.batch, hw, hw, cin, device=dev)
        dx1 = torch.empty_like(dx0)

        def rund(split, dx):
            gconv.SPLIT_BF16 = split
            layer.backward_data(w, gconv.View(g), gconv.View(dx), relu_src=x)
        td_exact = timed(lambda: rund(False, dx0), a.iters)
        td_split_all = timed(lambda: rund(True, dx1), a.iters)
        dd = gconv._desc(a.batch, gconv.View(g), gconv.View(dx1), cout, cin, (7, 7), (1, 1), (7, 7), (0, 0))
        gs = torch.empty(lib.dd_dconv_split_input_bytes(C.byref(dd)), device=dev, dtype=torch.uint8)
        pkd = torch.empty(lib.dd_dconv_split_packed_bytes(C.byref(dd)), device=dev, dtype=torch.uint8)
        _lib.check(lib.dd_dconv_split_input(P(g), P(gs), C.byref(dd), st), "in")
        _lib.check(lib.dd_dconv_split_pack(P(w), P(pkd), C.byref(dd), 0, cout * 49, 49, 0, cin, cout, st), "pk")
        td_in = timed(lambda: _lib.check(lib.dd_dconv_split_input(P(g), P(gs), C.byref(dd), st), "in"), a.iters)
        td_k = timed(lambda: _lib.check(lib.dd_dconv_fwd_split(P(gs), P(pkd), None, P(x), P(dx1), None, C.byref(dd), gconv.EPI_RELU_MASK, st), "k"), a.iters)
        ddiff = float((dx1 - dx0).abs().max() / dx0.abs().max())
        # the weight gradient from the two split images (none of the split passes is repeated: xs comes from the forward, gs from the data gradient)
        def runw(split):
            gconv.SPLIT_BF16 = split
            return layer.backward_weight(gconv.View(x), gconv.View(g))
        tw_exact = timed(lambda: runw(False), a.iters)
        dw0 = runw(False)[0]
        xs_c, gs_c = gconv.split_rows(gconv.View(x)), gconv.split_rows(gconv.View(g))

        def runw_split():
            gconv.SPLIT_BF16 = True
            return layer.backward_weight(gconv.View(x), gconv.View(g), xs=xs_c, gs=gs_c)
        tw_split = timed(runw_split, a.iters)
        dw1 = runw_split()[0]
        wdiff = float((dw1 - dw0).abs().max() / dw0.abs().max())
        flop = 2.0 * a.batch * hw * hw * 49 * cin * cout
        diff = float((y1 - y0).abs().max() / y0.abs().max())
        res[name] = {"exact_ms": round(t_exact, 3), "split_total_ms": round(t_split_all, 3), "split_kernel_ms": round(t_k, 3),
                     "split_input_ms": round(t_in, 3), "split_pack_ms": round(t_pk, 3), "exact_TF": round(flop / t_exact / 1e9, 1),
                     "split_kernel_TF_equiv": round(flop / t_k / 1e9, 1), "max_diff_of_peak": diff,
                     "dgrad_exact_ms": round(td_exact, 3), "dgrad_split_total_ms": round(td_split_all, 3), "dgrad_split_kernel_ms": round(td_k, 3),
                     "dgrad_split_input_ms": round(td_in, 3), "dgrad_split_kernel_TF_equiv": round(flop / td_k / 1e9, 1), "dgrad_max_diff_of_peak": ddiff,
                     "wgrad_exact_ms": round(tw_exact, 3), "wgrad_split_ms_planes_given": round(tw_split, 3), "wgrad_split_TF_equiv": round(flop / tw_split / 1e9, 1),
                     "wgrad_max_diff_of_peak": wdiff}
    gconv.SPLIT_BF16 = False
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
