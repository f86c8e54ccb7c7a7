"""GPU parity of the decoder / autoencoder step and the spatial bounding-box heads against the fixtures captured
from the reference's own modules (tests/golden/tiny_decoder.npz, spatial_heads.npz) and against the CPU oracle."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from driving_dirty_amd import synth

pytestmark = pytest.mark.gpu
CHAIN_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from driving_dirty_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel_err(got, ref, floor=1e-30):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(floor))


def _budget(g, key):
    """1e-3 of peak, or 3x the deviation of the reference's own fp32 run from its fp64 run (whichever is larger):
    the 65k-pixel sums behind the conv weight/bias gradients are ill-conditioned in ANY fp32 summation order."""
    a, b = g[key + "_f64"], g[key + "_f32"]
    ref_dev = float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-30))
    return max(CHAIN_TOL, 3.0 * ref_dev) if ref_dev < 1.0 else CHAIN_TOL


def _floor(g, key):
    if key.endswith(".fc1.bias"):
        wk = "grad." + key[:-len("bias")] + "weight_f64"
        if wk in g.files:
            return float(np.abs(g[wk]).max())
    return 1e-30


def _samp(t, idx):
    return t.detach().reshape(-1)[torch.from_numpy(idx).to(t.device)]


def test_tiny_decoder_against_reference_golden(dev, golden):
    from driving_dirty_amd.components import Decoder
    g = golden("tiny_decoder")
    dec = synth.fill_module(Decoder(16, 8, 3, 16, 22), seed=2).to(dev)
    dec.fc1.drop_p = dec.fc2.drop_p = 0.0
    z = synth.hash_uniform((3, 8), synth.key_salt("tiny_z"), -1.0, 1.0).to(dev).requires_grad_(True)
    dec.train()
    y = dec(z)
    assert y.shape == (3, 3, 16, 22)
    wy = synth.hash_uniform(tuple(y.shape), synth.key_salt("tiny_wy")).to(dev)
    (y * wy).sum().backward()
    assert rel_err(y, torch.from_numpy(g["y_f64"])) < _budget(g, "y")
    assert rel_err(z.grad, torch.from_numpy(g["grad.z_f64"])) < _budget(g, "grad.z")
    for k, p in dec.named_parameters():
        assert rel_err(p.grad, torch.from_numpy(g[f"grad.{k}_f64"]), floor=_floor(g, k)) < _budget(g, f"grad.{k}"), k


def test_autoencoder_step_against_oracle(dev):
    """BasicAE.training_step (mask one view, encode, decode, MSE) vs the CPU oracle on the same seeded inputs."""
    from driving_dirty_amd.autoencoder import BasicAE
    from oracle import ae_parts, steps
    hp = Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132, output_height=16, output_width=22,
                   learning_rate=1e-3, output_img_freq=500)
    ae = BasicAE(hp)
    synth.fill_module(ae, seed=13)
    enc = ae_parts.EncoderNet(16, 8, 3, 16, 132).double()
    dec = ae_parts.DecoderNet(16, 8, 3, 16, 22).double()
    enc.load_state_dict(ae.encoder.state_dict())
    dec.load_state_dict(ae.decoder.state_dict())
    ae = ae.to(dev)
    for m in (ae.encoder.fc1, ae.encoder.fc2, ae.decoder.fc1, ae.decoder.fc2, enc.fc1, enc.fc2, dec.fc1, dec.fc2):
        m.drop_p = 0.0
    views = synth.camera_batch(3, 16, 22, seed=13)
    np.random.seed(20200505)
    out = ae.training_step(views.to(dev), 0)
    out["loss"].backward()
    ref_loss, _ = steps.ae_loss(enc, dec, views.double(), np.random.RandomState(20200505))
    ref_loss.backward()
    assert abs(float(out["loss"].detach()) - float(ref_loss.detach())) / float(ref_loss.detach()) < 1e-4
    ref = dict(("encoder." + k, p) for k, p in enc.named_parameters())
    ref.update(("decoder." + k, p) for k, p in dec.named_parameters())
    # Train-mode BatchNorm1d over a batch of 3 makes these gradients ill-conditioned: the oracle's OWN fp32 run is
    # several % away from its fp64 run on the encoder side.  Budget per tensor: 1e-3 of peak, or three times the deviation
    # the reference arithmetic shows between fp32 and fp64, whichever is larger (3x as in the full-size golden test).
    enc32, dec32 = ae_parts.EncoderNet(16, 8, 3, 16, 132), ae_parts.DecoderNet(16, 8, 3, 16, 22)
    enc32.load_state_dict({k: v.float() for k, v in enc.state_dict().items()})
    dec32.load_state_dict({k: v.float() for k, v in dec.state_dict().items()})
    for m in (enc32.fc1, enc32.fc2, dec32.fc1, dec32.fc2):
        m.drop_p = 0.0
    steps.ae_loss(enc32, dec32, views, np.random.RandomState(20200505))[0].backward()
    ref32 = dict(("encoder." + k, p) for k, p in enc32.named_parameters())
    ref32.update(("decoder." + k, p) for k, p in dec32.named_parameters())
    for k, p in ae.named_parameters():
        scale = max(float(ref[k].grad.abs().max()), 1e-2 * float(ref[k[:-4] + "weight"].grad.abs().max()) if k.endswith("bias") else 0.0)
        budget = max(CHAIN_TOL, 3.0 * rel_err(ref32[k].grad, ref[k].grad, floor=scale))
        assert rel_err(p.grad, ref[k].grad, floor=scale) < budget, (k, budget)
    # the reference's API: six_to_one_task returns the NCHW wide image and the blanked view
    np.random.seed(20200505)
    x, y = ae.six_to_one_task(views.to(dev))
    xr, yr, t = steps.six_to_one_task(views, np.random.RandomState(20200505))
    assert torch.equal(x.cpu(), xr) and torch.equal(y.cpu(), yr)


def test_spatial_heads_three_way(dev, golden):
    """SpatialMappingCNN + RoadMapBoxesMergingCNN (and BoxesMergingCNN forward) at the reference's sizes, B = 1, three ways
    (tests/_branch_check.py): fp64 oracle on the product's branch (2e-4), flip census over the chain's 8 ReLU layers, and the
    reference-generated fixture spatial_heads.npz with per-tensor budgets (each layer alone is held to 2e-5 in test_gpu_gconv.py)."""
    from _branch_check import fixture_entry, grads_of, merge_masks, spatial_masks, three_way
    from driving_dirty_amd import heads
    from driving_dirty_amd.spatial import BoxesMergingCNN, RoadMapBoxesMergingCNN, SpatialMappingCNN
    from oracle import spatial_parts
    g = golden("spatial_heads")
    sm = synth.fill_module(SpatialMappingCNN(), seed=5).to(dev)
    rb = synth.fill_module(RoadMapBoxesMergingCNN(), seed=6).to(dev)
    bm = synth.fill_module(BoxesMergingCNN(), seed=7).to(dev)
    smr = synth.fill_module(spatial_parts.SpatialMapNet(), seed=5).double().to(dev)
    rbr = synth.fill_module(spatial_parts.RoadBoxMergeNet(), seed=6).double().to(dev)
    views = synth.camera_batch(1, seed=5).to(dev)
    rm = synth.road_maps(1, seed=5).float().unsqueeze(1).to(dev)
    ssr = synth.hash_uniform((1, 32, 128, 918), synth.key_salt("ssr"), 0.0, 1.0).to(dev).requires_grad_(True)
    heads.TRACE = {}
    try:
        space = sm(views)
        pred = rb(ssr, space, rm)
        tr = heads.TRACE
    finally:
        heads.TRACE = None
    assert space.shape == (1, 32, 256, 256) and pred.shape == (1, 1, 800, 800)
    wy = synth.hash_uniform(tuple(pred.shape), synth.key_salt("sp_wy")).to(dev)
    (pred * wy).sum().backward()
    product = dict(grads_of(sm, "space."), **grads_of(rb, "rboxm."), space=space.detach().contiguous(), pred=pred.detach(), dssr=ssr.grad)

    def run_oracle(branch):
        smr.zero_grad(set_to_none=True)
        rbr.zero_grad(set_to_none=True)
        a64 = ssr.detach().double().requires_grad_(True)
        s64 = smr(views.double(), branch=branch)
        p64 = rbr(a64, s64, rm.double(), branch=branch)
        (p64 * wy.double()).sum().backward()
        return dict(grads_of(smr, "space."), **grads_of(rbr, "rboxm."), space=s64.detach(), pred=p64.detach(), dssr=a64.grad)

    fixture = {k: fixture_entry(g, k) for k in product if fixture_entry(g, k) is not None}
    for name, key in (("space", "space"), ("pred", "pred"), ("dssr", "ssrgrad")):
        fixture[name] = (g[f"{key}_samp_f64"], g[f"{key}_samp_f32"], g[f"{key}_idx"])
    assert sorted(fixture) == sorted(product)
    three_way("spatial_heads_b1", product, run_oracle, dict(spatial_masks(tr), **merge_masks(tr, True)), fixture)
    for key, t in (("space", space), ("pred", pred)):
        s = g[f"{key}_sum_f64"]
        assert abs(float(t.double().abs().sum()) - s[1]) / s[1] < CHAIN_TOL
    with torch.no_grad():
        pred2 = bm(ssr.detach(), space.detach())
    assert rel_err(_samp(pred2, g["pred_nomap_idx"]), torch.from_numpy(g["pred_nomap_samp_f64"])) < _budget(g, "pred_nomap_samp")


@pytest.mark.parametrize("mse", [False, True])
def test_bbox_training_step_three_way(dev, mse):
    """BBSpatialRoadMap.training_step (spatial_w_rm.py:97-133,146-154; frozen encoder = config 3; BCE on probabilities or the
    ``mse_loss=True`` branch :128-129), reference sizes, B = 2, against the fp64 oracle on the product's branch (loss 1e-6,
    every head gradient 2e-4 of peak) + flip census (encoder conv stack, SpatialMappingCNN, RoadMapBoxesMergingCNN)."""
    from _branch_check import encoder_masks, grads_of, merge_masks, spatial_masks, three_way
    from driving_dirty_amd import heads, ops
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.spatial import BBSpatialRoadMap
    from oracle import ae_parts, spatial_parts, steps
    b = 2
    ae = BasicAE(Namespace(hidden_dim=16, latent_dim=8))
    model = BBSpatialRoadMap(Namespace(pretrained_ae=ae, unfreeze_epoch_no=5, learning_rate=1e-3, output_img_freq=500, mse_loss=mse))
    synth.fill_module(model, seed=19 if mse else 17)
    enc = ae_parts.EncoderNet(16, 8, 3, 256, 1836).double()
    enc.load_state_dict(model.ae.encoder.state_dict())
    enc.c3_only = True
    for p in enc.parameters():
        p.requires_grad_(False)
    enc.eval()                                   # freeze() = eval mode + no grads (lightning.py)
    smr, rbr = spatial_parts.SpatialMapNet().double(), spatial_parts.RoadBoxMergeNet().double()
    smr.load_state_dict(model.space_map_cnn.state_dict())
    rbr.load_state_dict(model.box_merge.state_dict())
    model, enc, smr, rbr = model.to(dev), enc.to(dev), smr.to(dev), rbr.to(dev)
    views, road = synth.camera_batch(b, seed=17).to(dev), synth.road_maps(b, seed=17).to(dev)
    tgt = (synth.hash_uniform((b, 800, 800), synth.key_salt("bbt"), 0.0, 1.0) < 0.02).float().to(dev)
    batch = (tuple(views), tuple({"bb_map": tgt[i]} for i in range(b)), tuple(road))
    ops.TRACE, heads.TRACE = {}, {}
    try:
        out = model.training_step(batch, 0)          # epoch 0 < unfreeze_epoch_no: the encoder stays frozen
        tr = dict(ops.TRACE, **heads.TRACE)
    finally:
        ops.TRACE = heads.TRACE = None
    out["loss"].backward()
    assert model.frozen and all(p.grad is None for p in model.ae.parameters())
    product = dict(grads_of(model.space_map_cnn, "space."), **grads_of(model.box_merge, "rboxm."), loss=out["loss"].detach())

    def run_oracle(branch):
        smr.zero_grad(set_to_none=True)
        rbr.zero_grad(set_to_none=True)
        space_rep = smr(views.double(), branch=branch)
        ssr = enc(steps.wide_stitch(views).double(), branch=branch)
        pred = rbr(ssr, space_rep, road.double().unsqueeze(1), branch=branch).squeeze(1)
        p, t = pred.reshape(b, -1), tgt.double().reshape(b, -1)
        loss = torch.nn.functional.mse_loss(p, t) if mse else torch.nn.functional.binary_cross_entropy(p, t)      # spatial_w_rm.py:128-131
        loss.backward()
        return dict(grads_of(smr, "space."), **grads_of(rbr, "rboxm."), loss=loss.detach())

    masks = dict(encoder_masks(tr, pool=False), **spatial_masks(tr), **merge_masks(tr, True))
    three_way(f"bbox_step_b2[{'mse' if mse else 'bce'}]", product, run_oracle, masks)


def test_joint_model_equals_sum_of_heads(dev):
    """Config 4 composition: the joint step's loss is the sum of the two single-head losses, and every gradient is
    the sum of the two single-head gradients (shared encoder), on the same seeded inputs."""
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.joint import JointRoadMapBBox
    from driving_dirty_amd import ops

    hp = dict(unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=500)
    ae = BasicAE(Namespace(hidden_dim=16, latent_dim=8))
    joint = JointRoadMapBBox(Namespace(pretrained_ae=ae, **hp))
    synth.fill_module(joint, seed=29)
    joint = joint.to(dev)
    enc = joint.ae.encoder
    enc.fc1.drop_p = enc.fc2.drop_p = 0.0
    views = synth.camera_batch(2, seed=29).to(dev)
    road = synth.road_maps(2, seed=29).to(dev)
    tgt = tuple({"bb_map": (synth.hash_uniform((800, 800), 100 + i, 0.0, 1.0) < 0.02).float().to(dev)} for i in range(2))
    out = joint.training_step((tuple(views), tgt, tuple(road)), 0)
    out["loss"].backward()
    gj = {k: p.grad.clone() for k, p in joint.named_parameters()}
    joint.zero_grad(set_to_none=True)
    # head 1: roadmap only
    z = enc.forward_nhwc4(ops.stitch6(views)[0])
    l1 = ops.BceWithLogits.apply(ops.linear(z, joint.fc1.weight, joint.fc1.bias), road.float().reshape(2, -1))
    l1.backward()
    # head 2: boxes only
    enc.c3_only = True
    feat = enc.forward_nhwc4(ops.stitch6(views)[0])
    enc.c3_only = False
    boxes = joint.box_merge(feat, joint.space_map_cnn(views), road.float().unsqueeze(1)).squeeze(1)
    l2 = ops.BceProbs.apply(boxes.reshape(2, -1), torch.stack([t["bb_map"] for t in tgt]).reshape(2, -1))
    l2.backward()
    assert abs(float(out["loss"].detach()) - float((l1 + l2).detach())) / float((l1 + l2).detach()) < 1e-6
    for k, p in joint.named_parameters():
        assert rel_err(gj[k], p.grad, floor=1e-12) < 2e-4, k


@pytest.mark.parametrize("hw", [(16, 22), (16, 70)])
def test_encoder_v2_conv_bn_relu_against_oracle(dev, hw):
    """components_v2 variant (Conv -> BN2d -> ReLU, stats in the conv epilogue, normalise-on-read) vs the hand-composed
    fp64 oracle: latent exit, c3_only exit, all gradients, running statistics, eval mode."""
    from driving_dirty_amd.components_v2 import Encoder
    from oracle import ae_parts
    h, w = hw
    enc = synth.fill_module(Encoder(16, 8, 3, h, w), seed=41)
    ref = ae_parts.EncoderNetV2(16, 8, 3, h, w).double()
    ref.load_state_dict(enc.state_dict())
    enc = enc.to(dev)
    for m in (enc.fc1, enc.fc2, ref.fc1, ref.fc2):
        m.drop_p = 0.0
    x = synth.hash_uniform((4, 3, h, w), synth.key_salt("v2x"), 0.0, 1.0)
    wz = synth.hash_uniform((4, 8), synth.key_salt("v2w"))
    enc.train(); ref.train()
    z = enc(x.to(dev))
    (z * wz.to(dev)).sum().backward()
    zr = ref(x.double())
    (zr * wz.double()).sum().backward()
    assert rel_err(z, zr) < CHAIN_TOL
    refp = dict(ref.named_parameters())
    for k, p in enc.named_parameters():
        floor = 1e-30
        if k.endswith("bias") and (k.startswith("c") or k.endswith(".fc1.bias")):      # a bias in front of a BatchNorm: zero gradient
            floor = float(refp[k[:-4] + "weight"].grad.abs().max())
        assert rel_err(p.grad, refp[k].grad, floor=floor) < CHAIN_TOL, k
    refb = dict(ref.named_buffers())
    for k, b in enc.named_buffers():
        assert rel_err(b.double(), refb[k].double()) < CHAIN_TOL, k
    # feature exit
    enc.zero_grad(); ref.zero_grad()
    enc.c3_only = ref.c3_only = True
    f = enc(x.to(dev)); fr = ref(x.double())
    wf = synth.hash_uniform(tuple(fr.shape), synth.key_salt("v2f"))
    (f * wf.to(dev)).sum().backward()
    (fr * wf.double()).sum().backward()
    assert rel_err(f, fr) < CHAIN_TOL
    for k in ("c1.weight", "bn1.weight", "bn1.bias", "c2.weight", "bn2.weight", "bn2.bias", "c3.weight", "bn3.weight", "bn3.bias"):
        assert rel_err(dict(enc.named_parameters())[k].grad, refp[k].grad) < CHAIN_TOL, k
    enc.c3_only = ref.c3_only = False
    enc.eval(); ref.eval()
    assert rel_err(enc(x.to(dev)), ref(x.double())) < CHAIN_TOL


def test_dropout_statistics_and_masked_view_draws(dev):
    """Glue the golden plan names: DenseBlock's dropout is always on (components.py:108), keeps with probability 0.8
    and rescales by 1.25; six_to_one_task draws its slot from numpy's GLOBAL state with the exclusive bound 5
    (autoencoder.py:59-60), reproduced here for the first ten draws under the reference's seed."""
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.components import DenseBlock
    blk = DenseBlock(8, 4096).to(dev).eval()                 # eval mode: BN uses running stats, dropout still drops
    with torch.no_grad():
        blk.fc1.weight.zero_()
        blk.fc1.bias.fill_(1.0)                              # every pre-dropout activation is exactly relu(bn(1)) = c > 0
    x = torch.zeros(64, 8, device=dev)
    y = blk(x)
    c = float(y.detach().max()) / 1.25
    assert c > 0
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01                            # 262144 draws: sigma = 0.0008
    assert torch.allclose(y[y > 0], torch.full_like(y[y > 0], 1.25 * c))
    assert not torch.equal(blk(x), y)                        # a fresh mask every call
    blk.drop_p = 0.0
    assert torch.equal(blk(x), blk(x))

    ae = BasicAE(Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132, output_height=16, output_width=22)).to(dev)
    views = synth.camera_batch(2, 16, 22, seed=3).to(dev)
    np.random.seed(20200505)                                 # autoencoder.py:16-18
    want = np.random.RandomState(20200505).randint(0, 5, size=10)
    for t in want:
        wide, target = ae.six_to_one_task(views)
        assert 0 <= t < 5
        assert float(wide[..., t * 22:(t + 1) * 22].abs().sum()) == 0.0
        assert float(wide.abs().sum()) > 0
        order = (0, 1, 2, 5, 4, 3)
        assert torch.equal(target, views[:, order[t]])
