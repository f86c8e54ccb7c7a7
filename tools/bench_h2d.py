#!/usr/bin/env python3
"""PCIe-inclusive rate of the roadmap step (bs = 32): what the step costs when the batch starts in (pinned) HOST memory,
as a DataLoader hands it over, instead of resident in HBM as bench.py's `value` has it.

    python tools/bench_h2d.py [--steps 10]
Three deliveries of the same synthetic batch:
  resident   inputs already in HBM (= bench.py)
  serial     fp32 views + bool road maps copied host->device on the compute stream before every step
  prefetch   the copy of batch i+1 runs on a copy stream while step i computes (double buffer)
and the same with uint8 camera frames ([B,6,H,W,3], what the JPEG decoder produces: 4x fewer bytes, /255 fused into the
stitch kernel dd_stitch6_u8) for the copy alone.  Diagnostic tool; numbers go to DESIGN.md section 5.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from driving_dirty_amd.optim import HipAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    model.training_step(bench.synthetic_batch(dev, 2, 0), 0)["loss"].backward()
    model.zero_grad(set_to_none=True)
    opt = HipAdam(model.parameters(), lr=1e-3)
    opt.overlap_with_backward()
    B = bench.BATCH
    g = torch.Generator().manual_seed(bench.SEED)
    host_views = torch.rand(B, 6, 3, bench.H, bench.W, generator=g).pin_memory()
    host_road = (torch.rand(B, 800, 800, generator=g) < 0.3).pin_memory()
    host_u8 = torch.randint(0, 256, (B, 6, bench.H, bench.W, 3), dtype=torch.uint8, generator=g).pin_memory()

    def step(views, road, i):
        model.zero_grad(set_to_none=True)
        out = model.training_step((tuple(views), tuple({} for _ in range(B)), tuple(road)), i)
        out["loss"].backward()
        opt.step()

    def timed(fn, n):
        fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i + 1)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    res = {}
    dv, dr = host_views.to(dev), host_road.to(dev)
    res["resident_ms"] = timed(lambda i: step(dv, dr, i), a.steps)
    res["serial_fp32_ms"] = timed(lambda i: step(host_views.to(dev, non_blocking=True), host_road.to(dev, non_blocking=True), i), a.steps)

    copy = torch.cuda.Stream()
    bufs = [(torch.empty_like(dv), torch.empty_like(dr)) for _ in range(2)]
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    freed = [torch.cuda.Event(), torch.cuda.Event()]

    def fetch(slot):
        with torch.cuda.stream(copy):
            copy.wait_event(freed[slot])
            bufs[slot][0].copy_(host_views, non_blocking=True)
            bufs[slot][1].copy_(host_road, non_blocking=True)
            ready[slot].record(copy)

    for e in freed:
        e.record()
    fetch(0)

    def prefetch_step(i):
        slot = i & 1
        fetch(slot ^ 1)
        torch.cuda.current_stream().wait_event(ready[slot])
        step(bufs[slot][0], bufs[slot][1], i)
        freed[slot].record()
    res["prefetch_fp32_ms"] = timed(prefetch_step, a.steps)

    def copy_only(src):
        dst = torch.empty(src.shape, dtype=src.dtype, device=dev)
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            dst.copy_(src, non_blocking=True)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        return ms, src.numel() * src.element_size() / ms / 1e6
    res["h2d_fp32_views_ms"], res["h2d_fp32_GBs"] = copy_only(host_views)
    res["h2d_u8_frames_ms"], res["h2d_u8_GBs"] = copy_only(host_u8)
    for k in list(res):
        res[k] = round(res[k], 3)
    for k in ("resident_ms", "serial_fp32_ms", "prefetch_fp32_ms"):
        res[k.replace("_ms", "_scenes_s")] = round(B / res[k] * 1e3, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
