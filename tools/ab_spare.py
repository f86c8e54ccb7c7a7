#!/usr/bin/env python3
"""A/B of HipAdam.EARLY_SPARE_CUS (compute units the early rank-B pass leaves free) on the autoencoder step (bs 32) or the roadmap step
(MODEL=roadmap): ONE process, one model, the value alternates between blocks of 20 steps."""
import os
import sys
import time
from argparse import Namespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd.autoencoder import BasicAE  # noqa: E402
from driving_dirty_amd.optim import HipAdam  # noqa: E402
from driving_dirty_amd.train import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
b = int(os.environ.get("BATCH", "32"))
values = [int(v) for v in os.environ.get("SPARES", "0,8").split(",")]
torch.manual_seed(20200505)
if os.environ.get("MODEL", "ae") == "roadmap":      # config 2, as bench.py builds it
    import bench  # noqa: E402
    model = bench.build_model(dev)
    views = bench.synthetic_batch(dev, b, 0)
else:
    model = BasicAE(Namespace(hidden_dim=128, latent_dim=64, learning_rate=1e-3, output_img_freq=500)).to(dev)
    views = torch.rand(b, 6, 3, 256, 306, device=dev)
ts = TrainStep(model, lr=1e-3, scheduler=False)
step = 0
for _ in range(5):
    ts(views, step); step += 1
for rep in range(int(os.environ.get("REPS", "4"))):
    for v in values:
        HipAdam.EARLY_SPARE_CUS = v
        ts(views, step); step += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ts(views, step); step += 1
        torch.cuda.synchronize()
        print(f"spare {v:3d}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
