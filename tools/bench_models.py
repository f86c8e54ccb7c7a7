#!/usr/bin/env python3
"""Step timings of the other §8 configurations (diagnostic; the contract benchmark is bench.py):
  ae    BasicAE masked-view pre-training step (config 1's GPU twin), fwd+bwd+Adam
  bbox  BBSpatialRoadMap step, frozen AE encoder (config 3), fwd+bwd+Adam on the heads
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd.autoencoder import BasicAE  # noqa: E402
from driving_dirty_amd.optim import HipAdam  # noqa: E402
from driving_dirty_amd.spatial import BBSpatialRoadMap  # noqa: E402


def run(step, steps, warmup):
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="ae,bbox")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(20200505)
    b = a.batch
    res = {}
    if "ae" in a.which:
        ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64, learning_rate=1e-3, output_img_freq=500)).to(dev)
        opt = HipAdam(ae.parameters(), lr=1e-3)
        opt.overlap_with_backward()          # big tensors' Adam pass rides under the MFMA-bound conv backward (as bench.py)
        views = torch.rand(b, 6, 3, 256, 306, device=dev)

        def step(i):
            ae.zero_grad(set_to_none=True)
            ae.training_step(views, i)["loss"].backward()
            opt.step()
        dt = run(step, a.steps, a.warmup)
        # decoder: fc2 160.4 M MACs + convT 629.2 M MACs per scene; encoder 34.112 GF + FC
        res["ae"] = {"ms_per_step": round(dt * 1e3, 2), "scenes_per_s": round(b / dt, 1), "batch": b}
        del ae, opt
        torch.cuda.empty_cache()
    if "bbox" in a.which:
        ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64))
        m = BBSpatialRoadMap(Namespace(pretrained_ae=ae, unfreeze_epoch_no=10**9, learning_rate=1e-3, output_img_freq=500,
                                       mse_loss=False)).to(dev)
        opt = HipAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        views = torch.rand(b, 6, 3, 256, 306, device=dev)
        road = torch.rand(b, 800, 800, device=dev) < 0.3
        tgt = tuple({"bb_map": (torch.rand(800, 800, device=dev) < 0.02).float()} for _ in range(b))
        batch = (tuple(views), tgt, tuple(road))

        def step(i):
            m.zero_grad(set_to_none=True)
            m.training_step(batch, i)["loss"].backward()
            opt.step()
        dt = run(step, a.steps, a.warmup)
        gf = (69.14 + 138.28 + 11.64) * b       # head fwd + head bwd + frozen encoder fwd, GFLOP
        res["bbox"] = {"ms_per_step": round(dt * 1e3, 2), "scenes_per_s": round(b / dt, 1), "batch": b,
                       "TF": round(gf / dt / 1e3, 1), "frac_fp32_mfma": round(gf / dt / 1e3 / 157.3, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
