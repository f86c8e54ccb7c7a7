#!/usr/bin/env python3
"""Merging heads with signed inputs: product (dilated kernel on / off) vs the fp64 oracle run with torch on the GPU (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import gconv, synth  # noqa: E402
from driving_dirty_amd.spatial import BoxesMergingCNN, RoadMapBoxesMergingCNN  # noqa: E402
from oracle import spatial_parts  # noqa: E402

dev = torch.device("cuda:0")


def stats(a, b):
    d = (a.detach().double() - b.detach().double()).abs() / b.detach().double().abs().max()
    return f"max {float(d.max()):.2e} frac>1e-3 {float((d > 1e-3).double().mean()):.4f} frac>1e-4 {float((d > 1e-4).double().mean()):.4f}"


for variant in ("rboxm", "boxm"):
    ssr0 = synth.hash_uniform((1, 32, 128, 918), synth.key_salt("ssr_signed"), -1.0, 1.0).to(dev)
    space0 = synth.hash_uniform((1, 32, 256, 256), synth.key_salt("space_signed"), -1.0, 1.0).to(dev)
    rm = synth.road_maps(1, seed=11).float().unsqueeze(1).to(dev)
    if variant == "rboxm":
        ref = synth.fill_module(spatial_parts.RoadBoxMergeNet(), seed=12).double().to(dev)
        mine = synth.fill_module(RoadMapBoxesMergingCNN(), seed=12).to(dev)
        extra_ref, extra = (rm.double(),), (rm,)
    else:
        ref = synth.fill_module(spatial_parts.BoxMergeNet(), seed=13).double().to(dev)
        mine = synth.fill_module(BoxesMergingCNN(), seed=13).to(dev)
        extra_ref, extra = (), ()
    a, s = ssr0.double().requires_grad_(True), space0.double().requires_grad_(True)
    pred = ref(a, s, *extra_ref)
    wy = synth.hash_uniform(tuple(pred.shape), synth.key_salt("ms_wy")).to(dev)
    (pred * wy.double()).sum().backward()
    for on in (True, False):
        gconv.DCONV = on
        mine.zero_grad(set_to_none=True)
        a1, s1 = ssr0.clone().requires_grad_(True), space0.clone().requires_grad_(True)
        p1 = mine(a1, s1, *extra)
        (p1 * wy).sum().backward()
        print(f"== {variant} dconv={on}: pred {stats(p1, pred)}")
        print(f"   dssr   {stats(a1.grad, a.grad)}")
        print(f"   dspace {stats(s1.grad, s.grad)}")
        refp = dict(ref.named_parameters())
        for k, p in mine.named_parameters():
            print(f"   {k:18s} {stats(p.grad, refp[k].grad)}")
    gconv.DCONV = True
