#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE modules.

Runs only in the build container (needs /root/reference on disk; the GPU box never sees it):

    python tests/golden/make_golden.py [--only NAME]

It imports the two torch-only reference files
    src/autoencoder/components.py                       (Encoder, Decoder, DenseBlock)
    src/bounding_box_model/spatial_bb/components.py     (SpatialMappingCNN, BoxesMergingCNN, RoadMapBoxesMergingCNN)
and, for the box rasteriser, src/utils/bb_to_img.py (numpy + Pillow),
fills their parameters and inputs from the closed-form generator in
``driving_dirty_amd.synth`` (so every consumer can rebuild the same tensors without the
reference), runs forward + backward in fp32 and fp64 and stores outputs / gradients.
Small cases store full tensors; full-size cases store values at fixed strided positions plus
fp64 sums.  Fixtures are DATA (inputs are implied by seeds, outputs are arrays); no reference
source text is stored.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from driving_dirty_amd import synth  # noqa: E402
from src.autoencoder.components import Decoder, DenseBlock, Encoder  # noqa: E402  (reference)
from src.bounding_box_model.spatial_bb.components import (  # noqa: E402  (reference)
    BoxesMergingCNN, RoadMapBoxesMergingCNN, SpatialMappingCNN)

torch.set_num_threads(8)


def _np(t):
    return t.detach().cpu().numpy()


def _set_drop(module, p):
    for m in module.modules():
        if isinstance(m, DenseBlock):
            m.drop_p = p


def _sample(t, n=64):
    """n values of t at a fixed stride through its flattened form (+ the last element)."""
    flat = t.detach().reshape(-1)
    idx = (torch.arange(n, dtype=torch.int64) * (flat.numel() - 1)) // (n - 1)
    return _np(flat[idx]), _np(idx)


def _grads(module):
    return {k: p.grad for k, p in module.named_parameters()}


def tiny_encoder(out):
    """Encoder(16, 8, 3, 16, 22): W_out = 11 so max_pool1d windows straddle rows; B = 3 for BN."""
    res = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        enc = synth.fill_module(Encoder(16, 8, 3, 16, 22), seed=1).to(dt)
        _set_drop(enc, 0.0)
        x = synth.hash_uniform((3, 3, 16, 22), synth.key_salt("tiny_x"), 0.0, 1.0).to(dt)
        wz = synth.hash_uniform((3, 8), synth.key_salt("tiny_wz")).to(dt)
        enc.train()
        z = enc(x)
        (z * wz).sum().backward()
        res[f"z_{tag}"] = _np(z)
        for k, g in _grads(enc).items():
            res[f"grad.{k}_{tag}"] = _np(g)
        for k, b in enc.named_buffers():
            res[f"buf.{k}_{tag}"] = _np(b)
        # feature map exit (c3_only) with an explicit upstream gradient
        enc.zero_grad()
        enc.c3_only = True
        feat = enc(x)
        wf = synth.hash_uniform(tuple(feat.shape), synth.key_salt("tiny_wf")).to(dt)
        (feat * wf).sum().backward()
        res[f"feat_{tag}"] = _np(feat)
        for k in ("c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias"):
            res[f"featgrad.{k}_{tag}"] = _np(dict(enc.named_parameters())[k].grad)
        enc.c3_only = False
        # eval mode: BN uses running statistics (dropout p = 0 here)
        enc.eval()
        res[f"z_eval_{tag}"] = _np(enc(x))
    np.savez_compressed(out, **res)


def tiny_decoder(out):
    res = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        dec = synth.fill_module(Decoder(16, 8, 3, 16, 22), seed=2).to(dt)
        _set_drop(dec, 0.0)
        z = synth.hash_uniform((3, 8), synth.key_salt("tiny_z"), -1.0, 1.0).to(dt).requires_grad_(True)
        dec.train()
        y = dec(z)
        wy = synth.hash_uniform(tuple(y.shape), synth.key_salt("tiny_wy")).to(dt)
        (y * wy).sum().backward()
        res[f"y_{tag}"] = _np(y)
        res[f"grad.z_{tag}"] = _np(z.grad)
        for k, g in _grads(dec).items():
            res[f"grad.{k}_{tag}"] = _np(g)
    np.savez_compressed(out, **res)


def default_init(out):
    """Default PyTorch init under the reference's seed: pins RNG-order parity of the constructors."""
    res = {}
    torch.manual_seed(20200505)
    enc = Encoder(16, 8, 3, 16, 22)
    dec = Decoder(16, 8, 3, 16, 22)
    for name, m in (("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            v = v.double().reshape(-1)
            res[f"{name}.{k}"] = np.array([v.sum().item(), v.abs().sum().item(), v[0].item(), v[-1].item()])
    torch.manual_seed(20200505)
    sm, bm, rb = SpatialMappingCNN(), BoxesMergingCNN(), RoadMapBoxesMergingCNN()
    for name, m in (("space", sm), ("boxm", bm), ("rboxm", rb)):
        for k, v in m.state_dict().items():
            v = v.double().reshape(-1)
            res[f"{name}.{k}"] = np.array([v.sum().item(), v.abs().sum().item(), v[0].item(), v[-1].item()])
    np.savez_compressed(out, **res)


def full_roadmap(out):
    """Config-2 shapes at B = 2: Encoder(128, 64, 3, 256, 1836) + Linear(64, 640000) + BCE-with-logits.

    Only the component modules come from the reference; the stitch / head / loss lines are the
    obvious torch calls of roadmap_bce_v2.py:58-62,75-81,106 written inline.
    """
    res = {}
    b = 2
    views = synth.camera_batch(b, seed=3)
    road = synth.road_maps(b, seed=3)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        enc = synth.fill_module(Encoder(128, 64, 3, 256, 1836), seed=3).to(dt)
        _set_drop(enc, 0.0)
        head = synth.fill_module(torch.nn.Linear(64, 640000), seed=4).to(dt)
        enc.train()
        x = views.to(dt)[:, [0, 1, 2, 5, 4, 3]]
        x = x.permute(0, 2, 3, 1, 4).reshape(b, 3, 256, -1)
        z = enc(x)
        logits = head(z)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, road.reshape(b, -1).to(dt))
        loss.backward()
        res[f"z_{tag}"] = _np(z)
        res[f"loss_{tag}"] = np.array(loss.item())
        res[f"logits_samp_{tag}"], res["logits_idx"] = _sample(logits, 256)
        res[f"logits_sum_{tag}"] = np.array(logits.double().sum().item())
        for k, g in list(_grads(enc).items()) + [("head." + k, p.grad) for k, p in head.named_parameters()]:
            if g.numel() <= 40000:
                res[f"grad.{k}_{tag}"] = _np(g)
            else:
                res[f"gradsamp.{k}_{tag}"], res[f"gradidx.{k}"] = _sample(g, 512)
            res[f"gradsum.{k}_{tag}"] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
        for k, v in enc.named_buffers():
            if v.numel() > 1:
                res[f"buf.{k}_{tag}"] = _np(v)
        # conv feature (c3_only) statistics at full size
        enc.c3_only = True
        with torch.no_grad():
            feat = enc(x)
        res[f"feat_samp_{tag}"], res["feat_idx"] = _sample(feat, 512)
        res[f"feat_sum_{tag}"] = np.array([feat.double().sum().item(), feat.double().abs().sum().item()])
        del enc, head
    np.savez_compressed(out, **res)


def spatial_heads(out):
    """SpatialMappingCNN / RoadMapBoxesMergingCNN / BoxesMergingCNN at the reference's sizes, B = 1."""
    res = {}
    b = 1
    views = synth.camera_batch(b, seed=5)
    rm = synth.road_maps(b, seed=5).float().unsqueeze(1)
    ssr0 = synth.hash_uniform((b, 32, 128, 918), synth.key_salt("ssr"), 0.0, 1.0)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        sm = synth.fill_module(SpatialMappingCNN(), seed=5).to(dt)
        rb = synth.fill_module(RoadMapBoxesMergingCNN(), seed=6).to(dt)
        bm = synth.fill_module(BoxesMergingCNN(), seed=7).to(dt)
        v = views.to(dt)
        ssr = ssr0.detach().clone().to(dt).requires_grad_(True)
        space = sm(v)
        pred = rb(ssr, space, rm.to(dt))
        wy = synth.hash_uniform(tuple(pred.shape), synth.key_salt("sp_wy")).to(dt)
        (pred * wy).sum().backward()
        res[f"space_samp_{tag}"], res["space_idx"] = _sample(space, 512)
        res[f"space_sum_{tag}"] = np.array([space.double().sum().item(), space.double().abs().sum().item()])
        res[f"pred_samp_{tag}"], res["pred_idx"] = _sample(pred, 512)
        res[f"pred_sum_{tag}"] = np.array([pred.double().sum().item(), pred.double().abs().sum().item()])
        res[f"ssrgrad_samp_{tag}"], res["ssrgrad_idx"] = _sample(ssr.grad, 512)
        for name, m in (("space", sm), ("rboxm", rb)):
            for k, g in _grads(m).items():
                if g.numel() <= 40000:
                    res[f"grad.{name}.{k}_{tag}"] = _np(g)
                else:
                    res[f"gradsamp.{name}.{k}_{tag}"], res[f"gradidx.{name}.{k}"] = _sample(g, 512)
        with torch.no_grad():
            pred2 = bm(ssr0.to(dt), space.detach())
        res[f"pred_nomap_samp_{tag}"], res["pred_nomap_idx"] = _sample(pred2, 512)
        res[f"pred_nomap_sum_{tag}"] = np.array([pred2.double().sum().item(), pred2.double().abs().sum().item()])
    np.savez_compressed(out, **res)


def box_raster(out):
    """Reference boxes_to_binary_map (src/utils/bb_to_img.py:5-20, Pillow underneath) on synthetic car boxes and on
    arbitrary quadrilaterals.  Inputs are stored too (their construction goes through libm's sin/cos)."""
    import PIL
    import importlib.util
    # the file itself needs only numpy + Pillow; its package __init__ pulls in torchvision, which this image lacks,
    # so the module is loaded by path
    spec = importlib.util.spec_from_file_location("ref_bb_to_img", "/root/reference/src/utils/bb_to_img.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    boxes_to_binary_map = mod.boxes_to_binary_map  # reference
    res = {"pillow_version": np.array(PIL.__version__)}
    sets = {"cars_a": synth.car_boxes(24, 1), "cars_b": synth.car_boxes(40, 2), "cars_f32": synth.car_boxes(16, 3).float(),
            "quads_a": synth.wild_quads(36, 1), "quads_b": synth.wild_quads(60, 2), "empty": torch.zeros(0, 2, 4, dtype=torch.float64)}
    for name, boxes in sets.items():
        m = np.ascontiguousarray(boxes_to_binary_map(boxes))
        assert m.shape == (800, 800) and set(np.unique(m)) <= {0.0, 1.0}
        res[f"{name}_boxes"] = _np(boxes)
        res[f"{name}_map_bits"] = np.packbits(m.astype(np.uint8), axis=1)
        res[f"{name}_ones"] = np.array(int(m.sum()))
    np.savez_compressed(out, **res)


CASES = {
    "tiny_encoder": tiny_encoder,
    "tiny_decoder": tiny_decoder,
    "default_init": default_init,
    "full_roadmap": full_roadmap,
    "spatial_heads": spatial_heads,
    "box_raster": box_raster,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    for name, fn in CASES.items():
        if a.only and a.only != name:
            continue
        path = os.path.join(HERE, name + ".npz")
        fn(path)
        print(f"{name}: {os.path.getsize(path)} bytes")
