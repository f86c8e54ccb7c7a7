"""GPU parity of the bf16 mixed-precision conv stack (BASELINE config 5) against oracle/bf16_parts.py.

bf16 outputs are compared EXACTLY except where the two fp32/fp64 accumulation orders land on different sides of a
rounding boundary: such an element may differ by one bf16 ulp, and only a small fraction of elements may do so."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F
from torch.nn import grad as nngrad

from driving_dirty_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from driving_dirty_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def hu(shape, name, lo=-1.0, hi=1.0, seed=0):
    return synth.hash_uniform(shape, synth.key_salt(name, seed), lo, hi)


def bf16r(t):
    return t.to(torch.bfloat16).to(t.dtype)


def assert_bf16_close(got, ref, what, max_flip_frac=5e-3):
    """got: bf16 tensor from the GPU; ref: fp64 tensor already rounded to bf16 by the oracle."""
    got, ref = got.detach().float().cpu().double(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    diff = (got - ref).abs()
    spacing = torch.pow(2.0, torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)     # bf16 ulp at ref
    spacing = torch.maximum(spacing, torch.full_like(spacing, 2.0 ** -133))
    assert bool((diff <= 1.001 * spacing + 1e-30).all()), (what, "more than one ulp", float((diff / spacing).max()))
    frac = float((diff > 0).double().mean())
    assert frac <= max_flip_frac, (what, "fraction of one-ulp flips", frac)


def nhwc(t):      # NCHW -> NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def pad4(x_nhwc3):
    b, h, w, _ = x_nhwc3.shape
    return torch.cat([x_nhwc3, torch.zeros(b, h, w, 1)], dim=3).contiguous()


def unpack_bits(bits, c=32):
    b = bits.cpu().numpy().astype(np.uint32)
    return torch.from_numpy(((b[..., None] >> np.arange(c, dtype=np.uint32)) & 1).astype(np.float64))      # [B,H,W,32]


CONV_CASES = [(1, 5, 40, 32, 1), (2, 9, 70, 32, 1), (1, 8, 33, 32, 2), (2, 7, 130, 32, 2), (2, 6, 50, 3, 1), (1, 3, 31, 3, 1),
              (1, 1, 64, 32, 1), (1, 2, 5, 32, 2)]


@pytest.mark.parametrize("b,h,w,cin,stride", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, b, h, w, cin, stride):
    from driving_dirty_amd import ops, ops_bf16 as ob
    x = bf16r(hu((b, cin, h, w), "x", 0.0, 1.0))
    wt = hu((32, cin, 3, 3), "w", -0.2, 0.2)
    bias = hu((32,), "b", -0.1, 0.1)
    d = ops.conv_desc(b, h, w, cin, stride)
    xd = (pad4(nhwc(x)) if cin == 3 else nhwc(x)).to(dev).to(torch.bfloat16)
    y, bits = ob.conv_fwd(xd, ob.conv_pack(wt.to(dev), d, ops.PACK_FWD), bias.to(dev), d)
    wr = bf16r(wt).double()
    ref = bf16r(F.relu(F.conv2d(x.double(), wr, bias.double(), stride=stride, padding=1)).float()).double()
    assert_bf16_close(nchw(y.float().cpu()), ref, "forward")
    assert torch.equal(unpack_bits(bits), (y.float().cpu() > 0).double())

    ho, wo = ref.shape[2:]
    g = bf16r(hu((b, 32, ho, wo), "g"))
    dw, db = ob.conv_wgrad(xd, nhwc(g).to(dev).to(torch.bfloat16), d)
    dw_ref = nngrad.conv2d_weight(x.double(), wr.shape, g.double(), stride=stride, padding=1)
    assert float((dw.cpu().double() - dw_ref).abs().max() / dw_ref.abs().max()) < 2e-5
    db_ref = g.double().sum(dim=(0, 2, 3))
    assert float((db.cpu().double() - db_ref).abs().max() / db_ref.abs().max().clamp_min(1e-3)) < 2e-5

    if cin == 32:
        mask = (hu((b, h, w, 32), "m") > 0).numpy()
        words = (mask.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=3).astype(np.uint32).view(np.int32)
        kind = ops.PACK_DGRAD_S1 if stride == 1 else ops.PACK_DGRAD_S2
        dx = ob.conv_dgrad(nhwc(g).to(dev).to(torch.bfloat16), ob.conv_pack(wt.to(dev), d, kind), torch.from_numpy(words).to(dev), d)
        dx_ref = nngrad.conv2d_input((b, 32, h, w), wr, g.double(), stride=stride, padding=1) * torch.from_numpy(mask).permute(0, 3, 1, 2)
        assert_bf16_close(nchw(dx.float().cpu()), bf16r(dx_ref.float()).double(), "dgrad")


def test_conv_refuses_unsupported(dev):
    from driving_dirty_amd import _lib, ops, ops_bf16 as ob
    with pytest.raises(_lib.HotpathError):
        ob.conv_pack(torch.zeros(32, 3, 3, 3, device=dev), ops.conv_desc(1, 8, 8, 3, 1), ops.PACK_DGRAD_S1)
    with pytest.raises(_lib.HotpathError):
        ob.conv_fwd(torch.zeros(1, 8, 8, 32, device=dev), torch.zeros(9216, device=dev, dtype=torch.bfloat16),
                    torch.zeros(32, device=dev), ops.conv_desc(1, 8, 8, 32, 1))      # fp32 input to the bf16 entry point


@pytest.mark.parametrize("b,h,w", [(2, 4, 6), (1, 8, 66)])
def test_pool_and_stitch(dev, b, h, w):
    from driving_dirty_amd import ops_bf16 as ob
    from oracle import steps
    feat = bf16r(hu((b, 32, h, w), "feat", -1.0, 1.0))
    fd = nhwc(feat).to(dev).to(torch.bfloat16)
    pooled = ob.pool4_fwd(fd)
    ref = F.max_pool1d(feat.reshape(b, 1, -1), 4).squeeze(1)
    assert torch.equal(pooled.cpu(), ref)
    gp = hu(tuple(ref.shape), "gp")
    featr = feat.clone().requires_grad_(True)
    F.max_pool1d(F.relu(featr).reshape(b, 1, -1), 4).squeeze(1).backward(gp)
    got = ob.pool4_relu_bwd(gp.to(dev), fd)
    assert torch.equal(nchw(got.float().cpu()), bf16r(featr.grad))
    v = hu((b, 6, 3, h, w), "views", 0.0, 1.0)
    wide = ob.stitch6_bf16(v.to(dev))
    assert torch.equal(wide[..., :3].float().cpu().permute(0, 3, 1, 2), bf16r(steps.wide_stitch(v)))
    assert float(wide[..., 3].float().abs().max()) == 0.0
    # the collate's tuple (one allocation per sample) through the pointer table: the same image, no torch.stack
    vd = v.to(dev)
    assert torch.equal(ob.stitch6_bf16_samples([vd[i].clone() for i in range(b)]), wide)


@pytest.mark.parametrize("b,h,w", [(2, 4, 6), (1, 8, 66), (3, 16, 130)])
def test_tiled_pool_with_routing_codes_equals_torch_and_the_plain_pair(dev, b, h, w):
    """dd_pool4_bf16_fwd_idx / dd_pool4_idx_relu_bf16_bwd (64-window tiles through LDS, the backward from 4-bit routing codes): max_pool1d(4)
    of the NCHW-flattened feature and its gradient behind the ReLU exactly as torch computes them -- ties go to the earliest index, a
    non-positive maximum passes nothing -- on ragged last tiles too, and bit for bit what the plain kernels give."""
    from driving_dirty_amd import ops_bf16 as ob
    feat = bf16r(hu((b, 32, h, w), "featt", -1.0, 1.0))
    flat = feat.reshape(b, -1)
    flat[:, 0:4] = 0.25                       # a window of four equal values: the first takes the gradient
    flat[:, 4:8] = torch.tensor([-0.5, -0.25, -0.25, -1.0])      # negative maximum: the ReLU in front is closed
    flat[:, 8:12] = torch.tensor([0.0, -1.0, 0.0, -1.0])         # maximum exactly zero: closed as well
    flat[:, 12:16] = torch.tensor([0.5, 0.75, 0.75, 0.125])      # tie between positions 1 and 2
    fd = nhwc(feat).to(dev).to(torch.bfloat16)
    assert ob.pool4_has_idx(h, w, 32)
    pooled, codes = ob.pool4_fwd_idx(fd)
    ref = F.max_pool1d(feat.reshape(b, 1, -1), 4).squeeze(1)
    assert torch.equal(pooled.cpu(), ref) and torch.equal(pooled, ob.pool4_fwd(fd))
    gp = hu(tuple(ref.shape), "gpt")
    featr = feat.clone().requires_grad_(True)
    F.max_pool1d(F.relu(featr).reshape(b, 1, -1), 4).squeeze(1).backward(gp)
    got = ob.pool4_idx_relu_bwd(gp.to(dev), codes, tuple(fd.shape))
    assert torch.equal(nchw(got.float().cpu()), bf16r(featr.grad))
    assert torch.equal(got, ob.pool4_relu_bwd(gp.to(dev), fd))
    with pytest.raises(Exception):
        ob.pool4_idx_relu_bwd(gp.to(dev), codes[:-1], tuple(fd.shape))


def test_conv_stack_against_oracle(dev):
    """Whole stack forward + backward at a small ragged size; weights/bias gradients are fp32 sums."""
    from driving_dirty_amd import ops_bf16 as ob
    from oracle import bf16_parts
    b, h, w = 2, 16, 132
    c1, c2, c3 = torch.nn.Conv2d(3, 32, 3, padding=1), torch.nn.Conv2d(32, 32, 3, padding=1), torch.nn.Conv2d(32, 32, 3, stride=2, padding=1)
    for i, m in enumerate((c1, c2, c3)):
        synth.fill_module(m, seed=60 + i)
    x = bf16r(hu((b, 3, h, w), "img", 0.0, 1.0))
    pooled_ref, (a1r, a2r, a3r) = bf16_parts.conv_stack_pooled(x, c1, c2, c3)
    gp = hu(tuple(pooled_ref.shape), "gpool")
    pooled_ref.backward(gp.double())
    ref_grads = [p.grad.clone() for m in (c1, c2, c3) for p in (m.weight, m.bias)]
    for m in (c1, c2, c3):
        m.zero_grad()
        m.to(dev)
    xd = pad4(nhwc(x)).to(dev).to(torch.bfloat16)
    pooled = ob.encoder_conv_stack(xd, c1, c2, c3)
    pooled.backward(gp.to(dev))
    # two layers of one-ulp flips upstream perturb a3 slightly beyond one ulp in rare elements: judge the pooled
    # vector against its peak, the gradients against theirs
    assert float((pooled.detach().cpu().double() - pooled_ref.detach()).abs().max() / pooled_ref.detach().abs().max()) < 4e-3
    got = [p.grad for m in (c1, c2, c3) for p in (m.weight, m.bias)]
    for name, g, r in zip(["c1.w", "c1.b", "c2.w", "c2.b", "c3.w", "c3.b"], got, ref_grads):
        err = float((g.cpu().double() - r.double()).abs().max() / r.double().abs().max())
        assert err < 4e-3, (name, err)


def test_roadmap_step_in_bf16_against_oracle(dev):
    """RoadMapBCE(precision='bf16').training_step vs the oracle: conv stack in the mixed-precision contract, FC tail,
    head and loss in fp64.  Also: the bf16 step stays within bf16's resolution of the fp32 step."""
    from argparse import Namespace
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    from oracle import ae_parts, bf16_parts, steps
    hp = dict(hidden_dim=16, latent_dim=8, input_height=16, input_width=132, output_height=16, output_width=22)
    ae = BasicAE(Namespace(**hp))
    model = RoadMapBCE(Namespace(pretrained_ae=ae, precision="bf16", unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=10 ** 9))
    synth.fill_module(model, seed=23)
    for blk in (model.ae.encoder.fc1, model.ae.encoder.fc2):
        blk.drop_p = 0.0
    enc = ae_parts.EncoderNet(16, 8, 3, 16, 132).double()
    enc.load_state_dict(model.ae.encoder.state_dict())
    enc.fc1.drop_p = enc.fc2.drop_p = 0.0
    head = torch.nn.Linear(8, 640000).double()
    head.load_state_dict(model.fc1.state_dict())
    model = model.to(dev)
    views = synth.camera_batch(3, 16, 22, seed=23)
    road = synth.road_maps(3, seed=23)
    out = model.training_step((tuple(views.to(dev)), None, tuple(road.to(dev))), 0)
    out["loss"].backward()

    z = bf16_parts.encoder_latent(enc, bf16r(steps.wide_stitch(views)))
    logits = head(z)
    ref = F.binary_cross_entropy_with_logits(logits.reshape(3, -1), road.double().reshape(3, -1))
    ref.backward()
    assert abs(float(out["loss"].detach()) - float(ref.detach())) / float(ref.detach()) < 1e-4
    refs = dict(list(("ae.encoder." + k, p) for k, p in enc.named_parameters()) + [("fc1." + k, p) for k, p in head.named_parameters()])
    for k, p in model.named_parameters():
        r = refs[k].grad.double()
        floor = 1e-30
        if k.endswith("fc1.bias") and "encoder" in k:       # zero by construction in front of BatchNorm
            floor = float(refs[k[:-4] + "weight"].grad.abs().max())
        err = float((p.grad.cpu().double() - r).abs().max() / max(float(r.abs().max()), floor))
        # one-ulp rounding flips (4e-3 relative each) in three stacked bf16 layers, amplified by BatchNorm over a batch
        # of 3: the conv gradients agree to ~5e-3 of peak whichever way the fp32 sums are ordered
        assert err < 1e-2, (k, err)

    # the same model in fp32: bf16 changes the loss by no more than bf16 resolution
    model.zero_grad()
    model.ae.encoder.precision = "fp32"
    l32 = model.training_step((tuple(views.to(dev)), None, tuple(road.to(dev))), 0)["loss"]
    assert abs(float(l32.detach()) - float(out["loss"].detach())) / float(l32.detach()) < 1e-2


def test_c3_only_exit_in_bf16_against_oracle(dev):
    """Encoder(precision='bf16', c3_only=True): the conv feature the box heads consume (components.py:44-45), forward and the
    conv gradients for an upstream gradient on the feature, vs the mixed-precision oracle."""
    from driving_dirty_amd.components import Encoder
    from oracle import ae_parts, bf16_parts
    enc = synth.fill_module(Encoder(16, 8, 3, 16, 70), seed=37)
    ref = ae_parts.EncoderNet(16, 8, 3, 16, 70)
    ref.load_state_dict(enc.state_dict())
    enc = enc.to(dev)
    enc.precision = "bf16"
    enc.c3_only = True
    x = synth.hash_uniform((3, 3, 16, 70), synth.key_salt("bf_c3x"), 0.0, 1.0)
    feat = enc(x.to(dev))
    assert feat.shape == (3, 32, 8, 35) and feat.dtype == torch.float32
    wf = synth.hash_uniform(tuple(feat.shape), synth.key_salt("bf_c3w"))
    (feat * wf.to(dev)).sum().backward()
    _, (a1, a2, a3) = bf16_parts.conv_stack_pooled(bf16r(x), ref.c1, ref.c2, ref.c3)
    (a3 * wf.double()).sum().backward()
    # two layers of one-ulp flips upstream perturb a3 slightly beyond one ulp in rare elements (as in test_conv_stack_against_oracle):
    # the feature is judged against its peak, and almost all of it must be bit-identical
    d = (feat.detach().cpu().double() - a3.detach()).abs()
    assert float(d.max() / a3.detach().abs().max()) < 4e-3 and float((d > 0).double().mean()) < 2e-2
    for k in ("c1.weight", "c1.bias", "c2.weight", "c2.bias", "c3.weight", "c3.bias"):
        r = dict(ref.named_parameters())[k].grad.double()
        err = float((dict(enc.named_parameters())[k].grad.cpu().double() - r).abs().max() / r.abs().max())
        assert err < 4e-3, (k, err)
