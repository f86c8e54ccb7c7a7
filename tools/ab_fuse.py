#!/usr/bin/env python3
"""AE pre-training step (config 1's GPU twin, bs 32) through TrainStep with the rank-B optimizer pass on / off (one process, same box).
PASSES_LAST=on|off|auto: HipAdam.passes_last (c2's data gradient first, the optimizer passes beside its weight gradient)."""
import os
import sys
import time
from argparse import Namespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd.autoencoder import BasicAE  # noqa: E402
from driving_dirty_amd.train import TrainStep  # noqa: E402

from driving_dirty_amd.optim import HipAdam  # noqa: E402

dev = torch.device("cuda:0")
b = int(os.environ.get("BATCH", "32"))
if os.environ.get("SPARE"):      # CUs the early rank-B pass leaves free (A/B of HipAdam.EARLY_SPARE_CUS)
    HipAdam.EARLY_SPARE_CUS = int(os.environ["SPARE"])
for rep in range(2):
    for fuse in (False, True):
        torch.manual_seed(20200505)
        ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64, learning_rate=1e-3, output_img_freq=500)).to(dev)
        pl = {"on": True, "off": False, "auto": "auto"}[os.environ.get("PASSES_LAST", "auto")]
        ts = TrainStep(ae, lr=1e-3, scheduler=False, fuse_linear_wgrad=fuse, passes_last=pl)
        views = torch.rand(b, 6, 3, 256, 306, device=dev)
        for i in range(3):
            ts(views, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            out = ts(views, 3 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"AE bs {b} fuse={fuse}: {dt * 1e3:.3f} ms/step  loss {float(out['loss'].detach()):.6f}", flush=True)
        ts.close()
        del ae, ts, views
        torch.cuda.empty_cache()
