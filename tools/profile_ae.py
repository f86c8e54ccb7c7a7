#!/usr/bin/env python3
"""The autoencoder pre-training step (config 1's GPU twin, bs 32) alone, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd.autoencoder import BasicAE
from driving_dirty_amd.train import TrainStep
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fuse = (sys.argv[2] if len(sys.argv) > 2 else "on") == "on"      # rank-B optimizer pass for the encoder fc1 / decoder fc2 weights
dev = torch.device("cuda:0")
if os.environ.get("SPARE"):      # CUs the early rank-B pass leaves free (A/B of HipAdam.EARLY_SPARE_CUS)
    from driving_dirty_amd.optim import HipAdam
    HipAdam.EARLY_SPARE_CUS = int(os.environ["SPARE"])
torch.manual_seed(20200505)
ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64, learning_rate=1e-3, output_img_freq=500)).to(dev)
ts = TrainStep(ae, lr=1e-3, scheduler=False, fuse_linear_wgrad=fuse)
views = torch.rand(b, 6, 3, 256, 306, device=dev)
def step(i):
    ts(views, i)
for i in range(3):
    step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(6):
    step(3 + i)
torch.cuda.synchronize()
print(f"AE bs {b}: {(time.perf_counter() - t0) / 6 * 1e3:.3f} ms/step")
