#!/usr/bin/env python3
"""Shader clock delivered while the headline step's longest kernel runs alone, while the optimizer's streaming pass runs alone,
and while both run side by side on two streams (as in the step: optim.HipAdam.overlap_with_backward).

A one-wave probe kernel (dd_clock_probe) on a third stream samples the shader-clock counter against the 100 MHz reference
counter; clock = d(shader ticks) / d(reference ticks) x 100 MHz.  Kernel times are HIP events on their own streams.

    python tools/clock_probe.py  ->  one JSON object (profiles/r03_clock_vs_overlap.json)
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
b, h, w = 32, 256, 1836
g = torch.randn(b, h, w, 32, device=dev)
w2 = torch.randn(32, 32, 3, 3, device=dev) * 0.06
bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (b, h, w), device=dev, dtype=torch.int32)
x4 = torch.rand(b, h, w, 4, device=dev)
d2 = ops.conv_desc(b, h, w, 32, 1)
p2d = ops.conv_wino2_pack(w2, d2, 1)
n = 940032 * 128                                  # fc1.fc1.weight
P, G, M, V = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
V.abs_()
main, side, probe = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def conv():
    ops.conv_wino2_dgrad_w1(g, p2d, bits, x4, d2)


def adam():
    ops.adam_step_flat(P, G, M, V, 1e-3, 0.9, 0.999, 1e-8, 5, 1.0)


def run(case, reps=12):
    nsamp = 4000
    samples = torch.zeros(2 * nsamp, device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    ev = {}
    with torch.cuda.stream(probe):
        _lib.check(_lib.lib().dd_clock_probe(C.c_void_p(samples.data_ptr()), nsamp, 1, C.c_void_p(probe.cuda_stream)), "dd_clock_probe")
    for name, stream, fn in (("conv", main, conv), ("adam", side, adam)):
        if name not in case:
            continue
        with torch.cuda.stream(stream):
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                fn()
            e.record()
            ev[name] = (s, e)
    torch.cuda.synchronize()
    t = samples.view(-1, 2).cpu().double()
    t = t[t[:, 1] > 0]
    # the probe outlives the kernels: keep the samples of the first `busy` milliseconds
    busy_ms = max(s.elapsed_time(e) for s, e in ev.values()) if ev else 5.0
    ref = (t[:, 1] - t[0, 1]) / 100e6 * 1e3         # ms since the first sample (100 MHz reference)
    keep = (ref > 0.3 * busy_ms) & (ref < 0.9 * busy_ms)
    tt = t[keep]
    clock_ghz = float((tt[-1, 0] - tt[0, 0]) / (tt[-1, 1] - tt[0, 1]) * 100e6 / 1e9)
    return {"case": "+".join(case) or "idle", "shader_clock_GHz": round(clock_ghz, 3),
            **{f"{k}_ms": round(s.elapsed_time(e) / reps, 4) for k, (s, e) in ev.items()}}


out = [run(()), run(("conv",)), run(("adam",)), run(("conv", "adam")), run(("conv",)), run(("conv", "adam"))]
print(json.dumps({"what": "shader clock (s_memtime vs the 100 MHz s_memrealtime, one probe wave) and kernel times: conv_wino2_fwd<RELU_BITS_W1> "
                          "(c2 data gradient + fused c1 weight gradient, bs 32) and adam_kernel over fc1.fc1.weight (120 M elements, 3.4 GB "
                          "of traffic), alone and side by side on two streams", "runs": out}, indent=1))
