#!/usr/bin/env python3
import os, sys
import numpy as np, torch
from argparse import Namespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops, synth
from driving_dirty_amd.autoencoder import BasicAE
dev = torch.device("cuda:0")
hp = Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132, output_height=16, output_width=22, learning_rate=1e-3, output_img_freq=500)
res = {}
for wino in (False, True):
    ops.WINOGRAD = wino
    ae = BasicAE(hp); synth.fill_module(ae, seed=13); ae = ae.to(dev)
    for m in (ae.encoder.fc1, ae.encoder.fc2, ae.decoder.fc1, ae.decoder.fc2): m.drop_p = 0.0
    views = synth.camera_batch(3, 16, 22, seed=13)
    np.random.seed(20200505)
    out = ae.training_step(views.to(dev), 0); out["loss"].backward()
    res[wino] = {k: p.grad.clone() for k, p in ae.named_parameters()}
    print("wino", wino, "loss", float(out["loss"]))
for k in res[False]:
    a, b = res[False][k], res[True][k]
    print(f"{k:28s} rel diff {float((a-b).abs().max()/a.abs().max().clamp_min(1e-30)):.3e}")
# isolate: conv stack only
ops.WINOGRAD = False
