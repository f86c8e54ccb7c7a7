#!/usr/bin/env python3
"""Where does the error of the conv weight gradients at the headline batch come from?  (diagnostic, GPU)

Runs the config-2 step at B = 32 on the fixture's inputs (tests/golden/full_roadmap_b32.npz holds the reference's fp64
gradients), captures the operands of the three weight-gradient kernels, and recomputes each gradient (a) as the product
does, in one launch over the batch, (b) image by image with the per-image results summed in fp64 on the host -- the same
kernels with 32x shorter accumulation chains and an exact cross-image sum.  If (b) is much closer to fp64 than (a), the
error is the summation inside / across the waves, not the operands.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops, synth  # noqa: E402
from driving_dirty_amd.autoencoder import BasicAE  # noqa: E402
from driving_dirty_amd.roadmap import RoadMapBCE  # noqa: E402


def err(got, ref):
    got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else got
    return float(np.abs(got - ref).max() / np.abs(ref).max())


def main():
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "full_roadmap_b32.npz"))
    b = 32
    ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64))
    synth.fill_module(ae.encoder, seed=3)
    model = RoadMapBCE(Namespace(pretrained_ae=ae, unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=500))
    synth.fill_module(model.fc1, seed=4)
    model = model.to(dev)
    model.ae.encoder.fc1.drop_p = model.ae.encoder.fc2.drop_p = 0.0
    views = synth.camera_batch(b, seed=3).to(dev)
    road = synth.road_maps(b, seed=3).to(dev)

    cap = {}
    orig_wgrad, orig_w2, orig_w1 = ops.conv_wgrad, ops.conv_wino2_wgrad, ops.conv_wino2_dgrad_w1

    def cap_wgrad(x, dy, desc):
        cap[f"direct_s{desc.stride}_c{desc.cin_real}"] = (x, dy)
        return orig_wgrad(x, dy, desc)

    def cap_w2(x, dy, desc, finish_stream=None):
        cap["c2"] = (x, dy)
        return orig_w2(x, dy, desc, finish_stream=finish_stream)

    def cap_w1(dy, packed, bits, x4, desc):
        cap["c1"] = (dy, packed, bits, x4)
        return orig_w1(dy, packed, bits, x4, desc)

    ops.conv_wgrad, ops.conv_wino2_wgrad, ops.conv_wino2_dgrad_w1 = cap_wgrad, cap_w2, cap_w1
    out = model.training_step((tuple(views), tuple({} for _ in range(b)), tuple(road)), 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    ops.conv_wgrad, ops.conv_wino2_wgrad, ops.conv_wino2_dgrad_w1 = orig_wgrad, orig_w2, orig_w1
    enc = model.ae.encoder
    h, w = 256, 1836

    def per_image(fn, nb=1):
        acc_w, acc_b = None, None
        for i in range(0, b, nb):
            dw, db = fn(i, i + nb)
            acc_w = dw.double() if acc_w is None else acc_w + dw.double()
            acc_b = db.double() if acc_b is None else acc_b + db.double()
        return acc_w, acc_b

    # ---- c3: direct stride-2 weight gradient
    x, dy = cap["direct_s2_c32"]
    for name, nb in (("one launch", b), ("8 images per launch, fp64 sum", 8), ("1 image per launch, fp64 sum", 1)):
        dw, db = per_image(lambda i, j: orig_wgrad(x[i:j].contiguous(), dy[i:j].contiguous(), ops.conv_desc(j - i, h, w, 32, 2)), nb)
        print(f"c3.weight  {name:34s} err {err(dw, g['grad.c3.weight_f64']):.2e}   bias {err(db, g['grad.c3.bias_f64']):.2e}")
    print(f"c3.weight  {'model':34s} err {err(enc.c3.weight.grad, g['grad.c3.weight_f64']):.2e}   bias {err(enc.c3.bias.grad, g['grad.c3.bias_f64']):.2e}")

    # ---- c2: Winograd F(3x3,2x2) weight gradient vs the direct kernel on the same operands
    x, dy = cap["c2"]
    for kind, fn in (("wino2", lambda xx, gg, d: orig_w2(xx, gg, d)), ("direct", lambda xx, gg, d: orig_wgrad(xx, gg, d))):
        for name, nb in (("one launch", b), ("1 image per launch, fp64 sum", 1)):
            dw, db = per_image(lambda i, j: fn(x[i:j].contiguous(), dy[i:j].contiguous(), ops.conv_desc(j - i, h, w, 32, 1)), nb)
            print(f"c2.weight  {kind:7s}{name:27s} err {err(dw, g['grad.c2.weight_f64']):.2e}   bias {err(db, g['grad.c2.bias_f64']):.2e}")
    print(f"c2.weight  {'model':34s} err {err(enc.c2.weight.grad, g['grad.c2.weight_f64']):.2e}   bias {err(enc.c2.bias.grad, g['grad.c2.bias_f64']):.2e}")

    # ---- c1: fused into c2's data gradient vs data gradient + direct weight gradient
    dy2, packed, bits, x4 = cap["c1"]
    for name, nb in (("fused, one launch", b), ("fused, 1 image per launch, fp64", 1)):
        dw, db = per_image(lambda i, j: orig_w1(dy2[i:j].contiguous(), packed, bits[i:j].contiguous(), x4[i:j].contiguous(),
                                                ops.conv_desc(j - i, h, w, 32, 1)), nb)
        print(f"c1.weight  {name:34s} err {err(dw, g['grad.c1.weight_f64']):.2e}   bias {err(db, g['grad.c1.bias_f64']):.2e}")
    g1 = ops.conv_wino2_dgrad_bits(dy2, packed, bits, ops.conv_desc(b, h, w, 32, 1))
    for name, nb in (("dgrad + direct wgrad, one launch", b), ("dgrad + direct, 1 image, fp64", 1)):
        dw, db = per_image(lambda i, j: orig_wgrad(x4[i:j].contiguous(), g1[i:j].contiguous(), ops.conv_desc(j - i, h, w, 3, 1)), nb)
        print(f"c1.weight  {name:34s} err {err(dw, g['grad.c1.weight_f64']):.2e}   bias {err(db, g['grad.c1.bias_f64']):.2e}")
    print(f"c1.weight  {'model':34s} err {err(enc.c1.weight.grad, g['grad.c1.weight_f64']):.2e}   bias {err(enc.c1.bias.grad, g['grad.c1.bias_f64']):.2e}")
    print("reference fp32 vs fp64:", {k: f"{np.abs(g[f'grad.{k}_f64'] - g[f'grad.{k}_f32']).max() / np.abs(g[f'grad.{k}_f64']).max():.2e}"
                                      for k in ("c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias")})


if __name__ == "__main__":
    main()
