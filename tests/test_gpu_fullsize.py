"""Full-size (BASELINE config 2: bs = 32, 256 x 1836) checks of the conv kernels through size-independent
properties, since the CPU oracle cannot finish these sizes in seconds:
  linearity     conv(a*x + b*z) = a*conv(x) + b*conv(z)                (no bias, no ReLU)
  adjointness   <dgrad(g), x> = <g, conv(x)>  and  <wgrad(x, g), W> = <g, conv(x)>   (fwd / dgrad / wgrad agree)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, H, W = 32, 256, 1836


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from driving_dirty_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("cin,stride", [(3, 1), (32, 1), (32, 2)])
def test_full_size_linearity_and_adjointness(dev, cin, stride):
    from driving_dirty_amd import ops
    g = torch.Generator(device=dev).manual_seed(cin * 10 + stride)
    cs = 4 if cin == 3 else 32
    d = ops.conv_desc(B, H, W, cin, stride)
    x = torch.randn(B, H, W, cs, device=dev, generator=g)
    z = torch.randn(B, H, W, cs, device=dev, generator=g)
    if cin == 3:
        x[..., 3] = 0
        z[..., 3] = 0
    wt = torch.randn(32, cin, 3, 3, device=dev, generator=g) * 0.1
    pk = ops.conv_pack(wt, d, ops.PACK_FWD)
    yx = ops.conv_fwd(x, pk, None, d, ops.EPI_NONE)
    yz = ops.conv_fwd(z, pk, None, d, ops.EPI_NONE)
    ylin = ops.conv_fwd(2.0 * x - 0.5 * z, pk, None, d, ops.EPI_NONE)
    err = float((ylin - (2.0 * yx - 0.5 * yz)).abs().max() / ylin.abs().max())
    assert err < 1e-5, err
    gy = torch.randn(yx.shape, device=dev, generator=g)
    ref = dot(gy, yx)
    dw, db = ops.conv_wgrad(x, gy, d)
    assert abs(dot(dw, wt) - ref) / abs(ref) < 1e-4
    assert abs(float(db.double().sum()) - float(gy.double().sum())) / float(gy.double().abs().sum()) < 1e-6
    if cin == 32:
        kind = ops.PACK_DGRAD_S1 if stride == 1 else ops.PACK_DGRAD_S2
        dx = ops.conv_dgrad(gy, ops.conv_pack(wt, d, kind), None, d)
        assert abs(dot(dx, x) - ref) / abs(ref) < 1e-4
        # ReLU-mask epilogue = elementwise product with (mask > 0)
        dxm = ops.conv_dgrad(gy, ops.conv_pack(wt, d, kind), z, d)
        assert torch.equal(dxm, dx * (z > 0))


def test_full_size_pool_roundtrip(dev):
    """pool backward routes each pooled gradient to exactly one element of its window (sum preserved)."""
    from driving_dirty_amd import ops
    feat = torch.rand(B, 128, 918, 32, device=dev) + 0.01
    pooled = ops.pool4_fwd(feat)
    assert pooled.shape == (B, 940032)
    gp = torch.rand_like(pooled) + 0.5          # strictly positive: every routed gradient is visible
    dfeat = ops.pool4_relu_bwd(gp, feat)
    assert abs(float(dfeat.double().sum()) - float(gp.double().sum())) / float(gp.double().sum()) < 1e-9
    assert int((dfeat != 0).sum()) == pooled.numel()
    # the maximum of every window, gathered back through the routing, reproduces the pooled values
    assert abs(dot(dfeat, feat) - dot(gp, pooled)) / dot(gp, pooled) < 1e-6
    pooled2, codes = ops.pool4_fwd_idx(feat)             # the routing-code form the encoder stack uses
    assert torch.equal(pooled2, pooled)
    assert torch.equal(ops.pool4_idx_relu_bwd(gp, codes, tuple(feat.shape)), dfeat)


def test_double_resolution_encoder_against_oracle(dev):
    """BASELINE config 5's input size (6 x 3 x 512 x 612 -> 512 x 3672 wide) in fp32, B = 3: latent and gradients vs the
    CPU oracle run in fp64."""
    from driving_dirty_amd import ops, synth
    from driving_dirty_amd.components import Encoder
    from oracle import ae_parts, steps
    torch.manual_seed(5)
    enc = Encoder(16, 8, 3, 512, 3672)
    ref = ae_parts.EncoderNet(16, 8, 3, 512, 3672).double()
    ref.load_state_dict(enc.state_dict())
    enc = enc.to(dev)
    for m in (enc.fc1, enc.fc2, ref.fc1, ref.fc2):
        m.drop_p = 0.0
    views = synth.camera_batch(3, 512, 612, seed=51)
    wz = synth.hash_uniform((3, 8), synth.key_salt("w2x"))
    z = enc.forward_nhwc4(ops.stitch6(views.to(dev))[0])
    (z * wz.to(dev)).sum().backward()
    torch.set_num_threads(16)
    zr = ref(steps.wide_stitch(views).double())
    (zr * wz.double()).sum().backward()

    def rel(a, b):
        return float((a.detach().double().cpu() - b.detach().double()).abs().max() / b.detach().double().abs().max())
    assert rel(z, zr) < 1e-3
    refp = dict(ref.named_parameters())
    # conv weight gradients are 1.4 M-term sums with heavy cancellation behind a small-batch BatchNorm1d: the
    # reference's own fp32 run is 5e-3..1e-2 off its fp64 run on them (tests/golden/full_roadmap.npz), so is any fp32 order
    for k in ("c1.weight", "c2.weight", "c3.weight", "c3.bias", "fc2.fc1.weight", "fc_z_out.weight"):
        assert rel(dict(enc.named_parameters())[k].grad, refp[k].grad) < 1e-2, k


def test_config5_bf16_conv_stack_against_oracle(dev):
    """BASELINE config 5 input size (6 x 3 x 512 x 612 -> 512 x 3672 wide image) through the bf16 conv stack, one
    scene, against the mixed-precision oracle (oracle/bf16_parts.py: fp64 between the bf16 rounding points): the pooled
    feature and all six parameter gradients.  Also the size-independent adjointness of the bf16 kernels at bs = 16:
    <dgrad(g), x> = <wgrad(x, g), W> (both sides are sums of exact bf16 products, accumulated in fp32)."""
    from driving_dirty_amd import ops, ops_bf16 as ob, synth
    from oracle import bf16_parts, steps
    h, w = 512, 6 * 612
    c1, c2, c3 = torch.nn.Conv2d(3, 32, 3, padding=1), torch.nn.Conv2d(32, 32, 3, padding=1), torch.nn.Conv2d(32, 32, 3, stride=2, padding=1)
    for i, m in enumerate((c1, c2, c3)):
        synth.fill_module(m, seed=70 + i)
    views = synth.camera_batch(1, 512, 612, seed=70)
    wide = steps.wide_stitch(views)
    pooled_ref, _ = bf16_parts.conv_stack_pooled(bf16_parts.bf16r(wide), c1, c2, c3)
    gp = synth.hash_uniform(tuple(pooled_ref.shape), synth.key_salt("gp5"))
    pooled_ref.backward(gp.double())
    ref = [p.grad.clone() for m in (c1, c2, c3) for p in (m.weight, m.bias)]
    for m in (c1, c2, c3):
        m.zero_grad()
        m.to(dev)
    pooled = ob.encoder_conv_stack(ob.stitch6_bf16(views.to(dev)), c1, c2, c3)
    pooled.backward(gp.to(dev))
    err = float((pooled.detach().cpu().double() - pooled_ref.detach()).abs().max() / pooled_ref.detach().abs().max())
    assert err < 8e-3, err                                   # at most a couple of bf16 ulps (2^-8) on the largest values
    frac_exact = float((pooled.detach().cpu().double() == pooled_ref.detach()).double().mean())
    assert frac_exact > 0.97, frac_exact                     # and bit-identical almost everywhere
    for name, g, r in zip(["c1.w", "c1.b", "c2.w", "c2.b", "c3.w", "c3.b"], [p.grad for m in (c1, c2, c3) for p in (m.weight, m.bias)], ref):
        e = float((g.cpu().double() - r.double()).abs().max() / r.double().abs().max())
        assert e < 5e-3, (name, e)

    b = 16
    d2 = ops.conv_desc(b, h, w, 32, 1)
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(b, h, w, 32, device=dev, generator=gen).to(torch.bfloat16)
    g = torch.randn(b, h, w, 32, device=dev, generator=gen).to(torch.bfloat16)
    wt = torch.randn(32, 32, 3, 3, device=dev, generator=gen) * 0.06
    wr = wt.to(torch.bfloat16).float()
    ones = torch.full((b, h, w), -1, device=dev, dtype=torch.int32)          # all ReLU signs set: no masking
    dx = ob.conv_dgrad(g, ob.conv_pack(wt, d2, ops.PACK_DGRAD_S1), ones, d2)
    dw, _ = ob.conv_wgrad(x, g, d2)
    lhs, rhs = dot(dx.float(), x.float()), dot(dw, wr)
    assert abs(lhs - rhs) / abs(rhs) < 2e-3, (lhs, rhs)      # dx is rounded to bf16 once per element (2^-9 relative, random sign)
