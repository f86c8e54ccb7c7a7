#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE modules.

Runs only in the build container (needs /root/reference on disk; the GPU box never sees it):

    python tests/golden/make_golden.py [--only NAME]

It imports the two torch-only reference files
    src/autoencoder/components.py                       (Encoder, Decoder, DenseBlock)
    src/bounding_box_model/spatial_bb/components.py     (SpatialMappingCNN, BoxesMergingCNN, RoadMapBoxesMergingCNN)
and src/autoencoder/components_v2.py (Decoder; Encoder through its own forward, see _reference_encoder_v2) and, for the
box rasteriser, src/utils/bb_to_img.py (numpy + Pillow),
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


def full_roadmap(out, b=2, hidden=128, latent=64):
    """Config-2 shapes at B = 2 (``full_roadmap``: the ill-conditioned edge case, train-mode BatchNorm1d over two rows) and
    at B = 32 (``full_roadmap_b32``: the headline batch): Encoder(128, 64, 3, 256, 1836) + Linear(64, 640000) +
    BCE-with-logits.

    Only the component modules come from the reference; the stitch / head / loss lines are the
    obvious torch calls of roadmap_bce_v2.py:58-62,75-81,106 written inline.
    """
    res = {}
    views = synth.camera_batch(b, seed=3)
    road = synth.road_maps(b, seed=3)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        enc = synth.fill_module(Encoder(hidden, latent, 3, 256, 1836), seed=3).to(dt)
        _set_drop(enc, 0.0)
        head = synth.fill_module(torch.nn.Linear(latent, 640000), seed=4).to(dt)
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
        del loss, logits, z
        # conv feature (c3_only) statistics at full size (the first two samples: per-sample independent)
        enc.c3_only = True
        with torch.no_grad():
            feat = enc(x[:2])
        res[f"feat_samp_{tag}"], res["feat_idx"] = _sample(feat, 512)
        res[f"feat_sum_{tag}"] = np.array([feat.double().sum().item(), feat.double().abs().sum().item()])
        del enc, head, feat
    np.savez_compressed(out, **res)


def full_roadmap_b32(out):
    full_roadmap(out, b=32)


def full_roadmap_w256(out):
    """The reference's DEFAULT width (autoencoder.py:33-34,164-166: hidden 256 / latent 128), config-2 shapes, B = 8."""
    full_roadmap(out, b=8, hidden=256, latent=128)


def full_decoder(out):
    """Decoder(128, 64, 3, 256, 306) at B = 8 (components.py:55-93; B = 2 is ill-conditioned: the reference fp32 run is 1e-2 off its fp64 run): the 128 -> 1,253,376 DenseBlock with its BatchNorm1d,
    the [B,64,128,153] view and the four ConvTranspose2d at 128x153 / 256x306.  Output and every gradient."""
    res = {}
    b = 8
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        dec = synth.fill_module(Decoder(128, 64, 3, 256, 306), seed=8).to(dt)
        _set_drop(dec, 0.0)
        z = synth.hash_uniform((b, 64), synth.key_salt("full_z"), -1.0, 1.0).to(dt).requires_grad_(True)
        dec.train()
        y = dec(z)
        wy = synth.hash_uniform(tuple(y.shape), synth.key_salt("full_wy")).to(dt)
        (y * wy).sum().backward()
        res[f"y_samp_{tag}"], res["y_idx"] = _sample(y, 1024)
        res[f"y_sum_{tag}"] = np.array([y.double().sum().item(), y.double().abs().sum().item()])
        res[f"grad.z_{tag}"] = _np(z.grad)
        for k, g in _grads(dec).items():
            if g.numel() <= 40000:
                res[f"grad.{k}_{tag}"] = _np(g)
            else:
                res[f"gradsamp.{k}_{tag}"], res[f"gradidx.{k}"] = _sample(g, 512)
            res[f"gradsum.{k}_{tag}"] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
        for k, v in dec.named_buffers():
            if v.numel() > 1:
                res[f"bufsamp.{k}_{tag}"], res[f"bufidx.{k}"] = _sample(v, 512)
        del dec
    np.savez_compressed(out, **res)


def full_ae_step(out):
    """Config 1's step at full size, B = 4 (the config's batch): masked-view task -> Encoder(128,64,3,256,1836) ->
    Decoder(128,64,3,256,306) -> mse_loss(y, y_hat).  The component modules are the reference's; the task lines are
    autoencoder.py:55-67,91 written inline with the masked view fixed to slot 2."""
    res = {}
    b, t = 4, 2
    views = synth.camera_batch(b, seed=9)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        enc = synth.fill_module(Encoder(128, 64, 3, 256, 1836), seed=9).to(dt)
        dec = synth.fill_module(Decoder(128, 64, 3, 256, 306), seed=10).to(dt)
        _set_drop(enc, 0.0)
        _set_drop(dec, 0.0)
        enc.train()
        dec.train()
        x = views.to(dt)[:, [0, 1, 2, 5, 4, 3]]
        x = x.permute(0, 2, 3, 1, 4).reshape(b, 3, 256, -1).clone()
        y = x[:, :, :, t * 306:(t + 1) * 306].clone()
        x[:, :, :, t * 306:(t + 1) * 306] = 0.0
        y_hat = dec(enc(x))
        loss = torch.nn.functional.mse_loss(y, y_hat)
        loss.backward()
        res[f"loss_{tag}"] = np.array(loss.item())
        res[f"yhat_samp_{tag}"], res["yhat_idx"] = _sample(y_hat, 1024)
        for name, m in (("encoder", enc), ("decoder", dec)):
            for k, g in _grads(m).items():
                if g.numel() <= 40000:
                    res[f"grad.{name}.{k}_{tag}"] = _np(g)
                else:
                    res[f"gradsamp.{name}.{k}_{tag}"], res[f"gradidx.{name}.{k}"] = _sample(g, 512)
                res[f"gradsum.{name}.{k}_{tag}"] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
        del enc, dec, loss, y_hat
    res["mask_slot"] = np.array(t)
    np.savez_compressed(out, **res)


def merge_signed(out):
    """Both merging heads as stand-alone modules with SIGNED inputs (the reference applies no ReLU to its inputs,
    spatial_bb/components.py:95-119,141-170): outputs, input gradients and every parameter gradient, B = 1."""
    res = {}
    b = 1
    rm = synth.road_maps(b, seed=11).float().unsqueeze(1)
    ssr0 = synth.hash_uniform((b, 32, 128, 918), synth.key_salt("ssr_signed"), -1.0, 1.0)
    space0 = synth.hash_uniform((b, 32, 256, 256), synth.key_salt("space_signed"), -1.0, 1.0)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        rb = synth.fill_module(RoadMapBoxesMergingCNN(), seed=12).to(dt)
        bm = synth.fill_module(BoxesMergingCNN(), seed=13).to(dt)
        for name, m, args in (("rboxm", rb, (rm.to(dt),)), ("boxm", bm, ())):
            ssr = ssr0.clone().to(dt).requires_grad_(True)
            space = space0.clone().to(dt).requires_grad_(True)
            pred = m(ssr, space, *args)
            wy = synth.hash_uniform(tuple(pred.shape), synth.key_salt("ms_wy")).to(dt)
            (pred * wy).sum().backward()
            res[f"{name}.pred_samp_{tag}"], res[f"{name}.pred_idx"] = _sample(pred, 512)
            res[f"{name}.pred_sum_{tag}"] = np.array([pred.double().sum().item(), pred.double().abs().sum().item()])
            for iname, t in (("ssr", ssr), ("space", space)):
                res[f"{name}.{iname}grad_samp_{tag}"], res[f"{name}.{iname}grad_idx"] = _sample(t.grad, 2048)
                res[f"{name}.{iname}grad_sum_{tag}"] = np.array([t.grad.double().sum().item(), t.grad.double().abs().sum().item()])
            for k, g in _grads(m).items():
                if g.numel() <= 40000:
                    res[f"grad.{name}.{k}_{tag}"] = _np(g)
                else:
                    res[f"gradsamp.{name}.{k}_{tag}"], res[f"gradidx.{name}.{k}"] = _sample(g, 512)
    np.savez_compressed(out, **res)


def tiny_decoder_v2(out):
    """components_v2.Decoder(16, 8, 3, 16, 22) (ConvTranspose2d -> BatchNorm2d -> ReLU x3 + ConvTranspose2d,
    components_v2.py:59-98; this class of the v2 file DOES construct): train mode (batch statistics, running-stat update,
    all gradients) and eval mode."""
    from src.autoencoder.components_v2 import Decoder as DecoderV2, DenseBlock as DenseBlockV2  # reference
    res = {}
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        dec = synth.fill_module(DecoderV2(16, 8, 3, 16, 22), seed=14).to(dt)
        for m in dec.modules():
            if isinstance(m, DenseBlockV2):
                m.drop_p = 0.0
        z = synth.hash_uniform((3, 8), synth.key_salt("v2_z"), -1.0, 1.0).to(dt).requires_grad_(True)
        dec.train()
        y = dec(z)
        wy = synth.hash_uniform(tuple(y.shape), synth.key_salt("v2_wy")).to(dt)
        (y * wy).sum().backward()
        res[f"y_{tag}"] = _np(y)
        res[f"grad.z_{tag}"] = _np(z.grad)
        for k, g in _grads(dec).items():
            res[f"grad.{k}_{tag}"] = _np(g)
        for k, v in dec.named_buffers():
            res[f"buf.{k}_{tag}"] = _np(v)
        dec.eval()
        with torch.no_grad():
            res[f"y_eval_{tag}"] = _np(dec(z))
    np.savez_compressed(out, **res)


def full_decoder_v2(out):
    """components_v2.Decoder(128, 64, 3, 256, 306) at B = 8: the BatchNorm2d reductions at 128x153 / 256x306."""
    from src.autoencoder.components_v2 import Decoder as DecoderV2, DenseBlock as DenseBlockV2  # reference
    res = {}
    b = 8
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        dec = synth.fill_module(DecoderV2(128, 64, 3, 256, 306), seed=15).to(dt)
        for m in dec.modules():
            if isinstance(m, DenseBlockV2):
                m.drop_p = 0.0
        z = synth.hash_uniform((b, 64), synth.key_salt("v2_full_z"), -1.0, 1.0).to(dt).requires_grad_(True)
        dec.train()
        y = dec(z)
        wy = synth.hash_uniform(tuple(y.shape), synth.key_salt("v2_full_wy")).to(dt)
        (y * wy).sum().backward()
        res[f"y_samp_{tag}"], res["y_idx"] = _sample(y, 1024)
        res[f"y_sum_{tag}"] = np.array([y.double().sum().item(), y.double().abs().sum().item()])
        res[f"grad.z_{tag}"] = _np(z.grad)
        for k, g in _grads(dec).items():
            if g.numel() <= 40000:
                res[f"grad.{k}_{tag}"] = _np(g)
            else:
                res[f"gradsamp.{k}_{tag}"], res[f"gradidx.{k}"] = _sample(g, 512)
        for k, v in dec.named_buffers():
            if 1 < v.numel() <= 64:
                res[f"buf.{k}_{tag}"] = _np(v)
        del dec
    np.savez_compressed(out, **res)


def _reference_encoder_v2(hidden_dim, latent_dim, in_channels, input_height, input_width):
    """An instance of the reference's components_v2.Encoder assembled WITHOUT its ``__init__``: that constructor raises at
    ``self.bn3 = nn.Conv2d(32)`` (components_v2.py:24), the one broken line of the class.  Everything the constructor would have
    set is set here in its order (components_v2.py:13-33) with ``bn3 = BatchNorm2d(32)`` -- the one stated interpretation --,
    the FC width comes from the class's own ``_calculate_output_dim`` and the arithmetic is the class's own ``forward``
    (components_v2.py:43-57), so the fixtures below are outputs of reference code, not of a restatement."""
    from torch import nn
    from src.autoencoder.components_v2 import Encoder as EncoderV2, DenseBlock as DenseBlockV2  # reference
    enc = EncoderV2.__new__(EncoderV2)
    nn.Module.__init__(enc)
    enc.hidden_dim, enc.latent_dim = hidden_dim, latent_dim
    enc.input_height, enc.input_width, enc.in_channels = input_height, input_width, in_channels
    enc.c1 = nn.Conv2d(in_channels, 32, kernel_size=3, padding=1)
    enc.bn1 = nn.BatchNorm2d(32)
    enc.c2 = nn.Conv2d(32, 32, kernel_size=3, padding=1)
    enc.bn2 = nn.BatchNorm2d(32)
    enc.c3 = nn.Conv2d(32, 32, kernel_size=3, stride=2, padding=1)
    enc.bn3 = nn.BatchNorm2d(32)
    enc.pooling_size = 4
    conv_out_dim = enc._calculate_output_dim(in_channels, input_height, input_width, enc.pooling_size)
    enc.fc1 = DenseBlockV2(conv_out_dim, hidden_dim)
    enc.fc2 = DenseBlockV2(hidden_dim, hidden_dim)
    enc.fc_z_out = nn.Linear(hidden_dim, latent_dim)
    enc.c3_only = False
    for m in enc.modules():
        if isinstance(m, DenseBlockV2):
            m.drop_p = 0.0
    return enc


def _encoder_v2_case(res, prefix, h, w, b, seed, xsalt, wzsalt, wfsalt, full):
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        enc = synth.fill_module(_reference_encoder_v2(16, 8, 3, h, w), seed=seed).to(dt)
        x = synth.hash_uniform((b, 3, h, w), synth.key_salt(xsalt), 0.0, 1.0).to(dt)
        wz = synth.hash_uniform((b, 8), synth.key_salt(wzsalt)).to(dt)
        enc.train()
        z = enc(x)                                               # the reference class's own forward
        (z * wz).sum().backward()
        res[f"{prefix}z_{tag}"] = _np(z)
        for k, g in _grads(enc).items():
            if g.numel() <= 40000:
                res[f"{prefix}grad.{k}_{tag}"] = _np(g)
            else:
                res[f"{prefix}gradsamp.{k}_{tag}"], res[f"{prefix}gradidx.{k}"] = _sample(g, 512)
        for k, v in enc.named_buffers():
            res[f"{prefix}buf.{k}_{tag}"] = _np(v)
        # feature exit (c3_only) from the same starting state, with an explicit upstream gradient
        enc = synth.fill_module(_reference_encoder_v2(16, 8, 3, h, w), seed=seed).to(dt)
        enc.train()
        enc.c3_only = True
        feat = enc(x)
        wf = synth.hash_uniform(tuple(feat.shape), synth.key_salt(wfsalt)).to(dt)
        (feat * wf).sum().backward()
        if full:
            res[f"{prefix}feat_samp_{tag}"], res[f"{prefix}feat_idx"] = _sample(feat, 2048)
            res[f"{prefix}feat_sum_{tag}"] = np.array([feat.double().sum().item(), feat.double().abs().sum().item()])
        else:
            res[f"{prefix}feat_{tag}"] = _np(feat)
        for k in ("c1.weight", "c1.bias", "bn1.weight", "bn1.bias", "c2.weight", "c2.bias", "bn2.weight", "bn2.bias",
                  "c3.weight", "c3.bias", "bn3.weight", "bn3.bias"):
            res[f"{prefix}featgrad.{k}_{tag}"] = _np(dict(enc.named_parameters())[k].grad)
        for k, v in enc.named_buffers():
            if k.startswith("bn"):
                res[f"{prefix}featbuf.{k}_{tag}"] = _np(v)
        if not full:
            enc.c3_only = False
            enc.eval()                                           # running statistics after the one train-mode pass above
            with torch.no_grad():
                res[f"{prefix}z_eval_{tag}"] = _np(enc(x))
        del enc


def tiny_encoder_v2(out):
    """components_v2.Encoder(16, 8, 3, 16, {22, 70}) at B = 4 through the reference class's own forward (Conv2d -> BatchNorm2d
    -> ReLU x3, ``c3_only`` exit, NCHW-order pool, FC tail): latent exit, feature exit, every gradient, running statistics,
    eval mode."""
    res = {}
    for h, w in ((16, 22), (16, 70)):
        _encoder_v2_case(res, f"h{h}w{w}.", h, w, 4, 41, "v2x", "v2w", "v2f", full=False)
    np.savez_compressed(out, **res)


def full_encoder_v2(out):
    """components_v2.Encoder(16, 8, 3, 256, 1836) at B = 4: BatchNorm2d statistics over 1.9 M / 470 k pixels per channel."""
    res = {}
    _encoder_v2_case(res, "", 256, 1836, 4, 43, "v2x_full", "v2w_full", "v2f_full", full=True)
    np.savez_compressed(out, **res)


def tiny_ae_ckpt(out):
    """A checkpoint in the layout Lightning 0.7.5 writes for ``BasicAE`` -- ``{'state_dict': {'encoder.*', 'decoder.*'},
    'hparams': {...}}`` -- holding the REFERENCE modules' default init under the reference's seed
    (autoencoder.py:16-18: Encoder first, then Decoder, as BasicAE.__init__ builds them, :26-30), plus the outputs the
    reference modules compute from it.  ``BasicAE.load_from_checkpoint`` must load the file and reproduce them."""
    hp = dict(hidden_dim=16, latent_dim=8, input_height=16, input_width=6 * 22, output_height=16, output_width=22,
              in_channels=3, batch_size=3, learning_rate=1e-3, output_img_freq=500)
    torch.manual_seed(20200505)
    enc = Encoder(hp["hidden_dim"], hp["latent_dim"], 3, hp["input_height"], hp["input_width"])
    dec = Decoder(hp["hidden_dim"], hp["latent_dim"], 3, hp["output_height"], hp["output_width"])
    sd = {f"encoder.{k}": v.clone() for k, v in enc.state_dict().items()}
    sd.update({f"decoder.{k}": v.clone() for k, v in dec.state_dict().items()})
    torch.save({"state_dict": sd, "hparams": hp, "epoch": 0}, os.path.join(HERE, "tiny_ae.ckpt"))
    res = {"keys": np.array(sorted(sd))}
    x = synth.hash_uniform((3, 3, 16, 6 * 22), synth.key_salt("ckpt_x"), 0.0, 1.0)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        e, d = Encoder(16, 8, 3, 16, 6 * 22).to(dt), Decoder(16, 8, 3, 16, 22).to(dt)
        e.load_state_dict({k: v.to(dt) for k, v in enc.state_dict().items()})
        d.load_state_dict({k: v.to(dt) for k, v in dec.state_dict().items()})
        _set_drop(e, 0.0)
        _set_drop(d, 0.0)
        e.train()
        d.train()
        z = e(x.to(dt))
        res[f"z_{tag}"] = _np(z)
        res[f"y_{tag}"] = _np(d(z))
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
    "full_roadmap_b32": full_roadmap_b32,
    "full_roadmap_w256": full_roadmap_w256,
    "full_decoder": full_decoder,
    "full_ae_step": full_ae_step,
    "merge_signed": merge_signed,
    "tiny_decoder_v2": tiny_decoder_v2,
    "full_decoder_v2": full_decoder_v2,
    "tiny_encoder_v2": tiny_encoder_v2,
    "full_encoder_v2": full_encoder_v2,
    "tiny_ae_ckpt": tiny_ae_ckpt,
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
