#!/usr/bin/env python3
"""Where in the step the big gradients appear and the big weights are first read -- the time a gradient reduce-scatter / all-reduce
has to hide under (rest of the backward) and the time an all-gather has (start of the forward up to the first read of the weight).

    python tools/step_phases.py --config {2,4,5} [--steps 10]

HIP events on the compute stream: step start, first read of each big weight in the forward (ops.linear), end of forward, the moment
autograd has accumulated each big gradient, end of backward, end of the optimizer.  Prints one JSON record (medians, ms from step
start).  One GPU, replicated optimizer, no collectives: the phases of the per-GPU step DESIGN.md section 6 budgets against.
"""
import argparse
import json
import os
import statistics
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bench
    from driving_dirty_amd import ops
    from driving_dirty_amd.train import TrainStep
    dev = torch.device("cuda:0")
    cfg = bench.setup_config(Namespace(config=a.config, rows_per_task=0), dev, 0)
    model, batch = cfg["model"], cfg["batch"]
    ts = TrainStep(model, lr=1e-3, scheduler=False)
    big = {n: p for n, p in model.named_parameters() if p.numel() >= (1 << 20)}
    marks = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.setdefault(name, e)

    inner = ops.linear

    def linear(x, w, b):
        for n, p in big.items():
            if w is p:
                mark("first_read:" + n)
        return inner(x, w, b)
    ops.linear = linear
    import driving_dirty_amd.components as comp
    import driving_dirty_amd.roadmap as rm
    comp.ops.linear = linear
    rm.ops.linear = linear
    ts(batch, 0)                                              # the first training_step unfreezes the extractor where the config says so
    for n, p in big.items():
        if p.requires_grad:
            p.register_post_accumulate_grad_hook(lambda q, n=n: mark("grad_ready:" + n))
    rows = []
    for i in range(a.warmup + a.steps):
        marks.clear()
        mark("start")
        model.zero_grad(set_to_none=True)
        out = model.training_step(batch, i)
        mark("forward_end")
        out["loss"].backward()
        mark("backward_end")
        ts.sync.finish()
        ts.optimizer.step(grad_scale=1.0)
        mark("step_end")
        torch.cuda.synchronize()
        if i >= a.warmup:
            rows.append({k: marks["start"].elapsed_time(e) for k, e in marks.items() if k != "start"})
    keys = rows[0].keys()
    rec = {"config": a.config, "per_gpu_batch": cfg["per_gpu_batch"], "big_tensors_MB": {n: round(p.numel() * 4 / 1e6, 1) for n, p in big.items()},
           "ms_from_step_start": {k: round(statistics.median(r[k] for r in rows), 3) for k in keys}}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
