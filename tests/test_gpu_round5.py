"""GPU tests, fifth set: the rank-B optimizer pass (``dd_adam_step_rankb``: the weight gradient of a big Linear layer formed inside its
Adam pass, reference components.py:105 / roadmap_bce_v2.py:75,154-157) against fp64, and ``TrainStep`` with it against the materialised
gradient path."""
import os
import sys
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_round4 import _tiny_batch, _tiny_model  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from driving_dirty_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _adam64(p, m, v, g, lr, b1, b2, eps, step):
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)


# n x k (multiples of 4, as every Linear kernel of the path asks): whole tiles, ragged last n-tile (52 = 3 x 16 + 4) and n-group, ragged
# last k-tile (132 = 2 x 64 + 4), one k-tile (the head's shape class), less than one tile, more group-tiles than workgroups (350 > 256)
RANKB_SHAPES = [(128, 1024), (52, 132), (640, 64), (20, 4), (16 * 70, 64 * 20)]


@pytest.mark.parametrize("n,k", RANKB_SHAPES)
@pytest.mark.parametrize("rows,with_bias,scale", [(32, True, 1.0), (7, False, 1.0), (64, True, 0.5), (1, True, 1.0)])
@pytest.mark.parametrize("signed", [True, False])
def test_adam_rankb_matches_fp64(dev, n, k, rows, with_bias, scale, signed):
    """Three steps of dd_adam_step_rankb against torch.optim.Adam's arithmetic in fp64 on the fp64 gradient dy^T x (and dy's column sums
    for the bias): m, v to fp32 rounding of the gradient, p within the Adam kernel's own 1e-6.  Signed factors: the gradient elements
    that cancel to nearly zero carry the fp32 sum's ABSOLUTE rounding as a large RELATIVE error, and Adam's m / sqrt(v) turns that into
    the same fraction of a full-size update (any fp32 gradient does, torch's included): there p is held to a fifth of one update (the worst of 1.4 M elements measured 4 %), and the
    1e-6 is checked on the unsigned run, where every gradient element is well conditioned."""
    from driving_dirty_amd import ops
    g = torch.Generator().manual_seed(n * 7 + k + rows)
    p0 = (torch.rand(n, k, generator=g) - 0.5)
    b0 = (torch.rand(n, generator=g) - 0.5)
    p, m, v = p0.clone().to(dev), torch.zeros(n, k, device=dev), torch.zeros(n, k, device=dev)
    b, bm, bv = b0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pr, mr, vr = p0.double(), torch.zeros(n, k, dtype=torch.float64), torch.zeros(n, k, dtype=torch.float64)
    br, bmr, bvr = b0.double(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    # the kernels take beta1 / beta2 as fp32 arguments: 1 - fp32(0.999) is 1.3e-5 off 1 - 0.999, in v itself but not in v_hat (the bias
    # correction is computed from the same fp32 value), so the reference uses the rounded betas to judge m and v at rounding level
    b1, b2 = float(torch.tensor(0.9, dtype=torch.float32)), float(torch.tensor(0.999, dtype=torch.float32))
    for step in range(1, 4):
        x = torch.rand(rows, k, generator=g) - (0.5 if signed else 0.0)
        dy = (torch.rand(rows, n, generator=g) - (0.5 if signed else 0.0)) * 0.1 * step
        gw = dy.double().t() @ x.double() * scale
        _adam64(pr, mr, vr, gw, 1e-3, b1, b2, 1e-8, step)
        _adam64(br, bmr, bvr, dy.double().sum(0) * scale, 1e-3, b1, b2, 1e-8, step)
        ops.adam_step_rankb(p, m, v, dy.to(dev), x.to(dev), b if with_bias else None, bm if with_bias else None,
                            bv if with_bias else None, 1e-3, 0.9, 0.999, 1e-8, step, scale)
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    assert rel(m, mr) < 2e-6 and rel(v, vr) < 2e-6
    if signed:
        assert float((p.double().cpu() - pr).abs().max()) < 0.2 * 1e-3
    else:
        assert rel(p, pr) < 1e-6
    if with_bias:
        assert rel(bm, bmr) < 2e-6 and (signed or rel(b, br) < 1e-6)
    else:
        assert torch.equal(b.cpu(), b0) and float(bm.abs().max()) == 0.0


@pytest.mark.parametrize("rows,with_bias", [(16, True), (32, False), (5, True)])
def test_adam_rankb_wide_tiles_for_the_pass_by_itself_equal_the_budgeted_kernel(dev, rows, with_bias):
    """With more than one workgroup per CU (dd_set_adam_blocks_per_cu: the pass runs by itself, bf16 models) long rows take the 16-row x
    256-column tile form: every element gets the same contraction in the same order as in the 64 x 64 form, so p / m / v and the bias agree
    bit for bit; K not a multiple of 256 (a ragged last tile) and N not a multiple of 16 included."""
    from driving_dirty_amd import _lib, ops
    lib = _lib.lib()
    torch.manual_seed(rows)
    n, k = 136, 64 * 70 + 12
    p0 = torch.randn(n, k, device=dev) * 0.02
    x, dy = torch.randn(rows, k, device=dev), torch.randn(rows, n, device=dev) * 1e-2
    b0 = torch.randn(n, device=dev) * 0.02
    outs = []
    for blocks in (1, 4):
        assert lib.dd_set_adam_blocks_per_cu(blocks) == 0
        try:
            p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
            b, bm, bv = (b0.clone(), torch.zeros_like(b0), torch.zeros_like(b0)) if with_bias else (None, None, None)
            for step in (1, 2):
                ops.adam_step_rankb(p, m, v, dy, x, b, bm, bv, 1e-3, 0.9, 0.999, 1e-8, step)
            outs.append((p, m, v, b))
        finally:
            assert lib.dd_set_adam_blocks_per_cu(1) == 0
    for a, c in zip(outs[0], outs[1]):
        assert (a is None and c is None) or torch.equal(a, c)


def test_adam_rankb_refuses_bad_arguments(dev):
    from driving_dirty_amd import _lib, ops
    p = torch.zeros(16, 6, device=dev)
    with pytest.raises(_lib.HotpathError):      # K % 4 != 0
        ops.adam_step_rankb(p, p.clone(), p.clone(), torch.zeros(4, 16, device=dev), torch.zeros(4, 6, device=dev), None, None, None,
                            1e-3, 0.9, 0.999, 1e-8, 1)
    p = torch.zeros(16, 8, device=dev)
    with pytest.raises(_lib.HotpathError):      # factor shapes disagree with the weight
        ops.adam_step_rankb(p, p.clone(), p.clone(), torch.zeros(4, 16, device=dev), torch.zeros(5, 8, device=dev), None, None, None,
                            1e-3, 0.9, 0.999, 1e-8, 1)


def test_adam_rankb_spare_compute_units_change_the_grid_not_the_result(dev):
    """dd_set_adam_spare_cus: the pass launches that many fewer workgroups per dd_set_adam_blocks_per_cu (HipAdam's early slot leaves one
    CU per XCD to single-workgroup kernels of the backward); every element takes the same arithmetic whichever workgroup owns it."""
    from driving_dirty_amd import _lib, ops
    lib = _lib.lib()
    torch.manual_seed(5)
    n, k, rows = 1024, 2052, 32                                 # 256 row groups x 2 k-tiles... more tiles than workgroups either way
    p0 = torch.randn(n, k, device=dev) * 0.02
    x, dy = torch.randn(rows, k, device=dev), torch.randn(rows, n, device=dev) * 1e-2
    outs = []
    for spare in (0, 8, 128):
        assert lib.dd_set_adam_spare_cus(spare) == 0
        try:
            p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
            ops.adam_step_rankb(p, m, v, dy, x, None, None, None, 1e-3, 0.9, 0.999, 1e-8, 1)
            outs.append((p, m, v))
        finally:
            assert lib.dd_set_adam_spare_cus(0) == 0
    for p, m, v in outs[1:]:
        assert torch.equal(p, outs[0][0]) and torch.equal(m, outs[0][1]) and torch.equal(v, outs[0][2])
    assert lib.dd_set_adam_spare_cus(-1) != 0 and lib.dd_set_adam_spare_cus(129) != 0


def test_column_sum_matches_fp64(dev):
    from driving_dirty_amd import ops
    for m, n in ((32, 640), (7, 50), (3, 1027)):
        dy = torch.rand(m, n, device=dev) - 0.5
        got = ops.column_sum(dy)
        ref = dy.double().sum(0)
        assert float((got.double() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("overlap", [True, False])
def test_trainstep_rankb_matches_the_materialised_gradient_path(dev, overlap):
    """TrainStep(fuse_linear_wgrad=True): the two big Linear layers (head 640000 x 8, encoder fc1 16 x 1056) never get a ``.grad``; their
    Adam pass forms it.  After every step the parameters equal those of the materialised path (same gradient products, summed four
    batch rows per matrix instruction instead of two) within the Adam kernel's 1e-6; both start each step from the same parameters."""
    from driving_dirty_amd.train import TrainStep
    a, b = _tiny_model(dev), _tiny_model(dev)
    ta = TrainStep(a, lr=1e-2, adam_overlap=overlap, big_numel=4096, scheduler=False, fuse_linear_wgrad=False)
    tb = TrainStep(b, lr=1e-2, adam_overlap=overlap, big_numel=4096, scheduler=False)
    assert not ta.fused and {id(w) for w in tb.fused} == {id(b.fc1.weight), id(b.ae.encoder.fc1.fc1.weight)}
    for step in range(3):
        if step:
            with torch.no_grad():
                for p, q in zip(a.parameters(), b.parameters()):
                    p.copy_(q)
            for (_, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
                u.copy_(v)
        batch = _tiny_batch(dev, step, 0)
        la, lb = ta(batch, step)["loss"], tb(batch, step)["loss"]
        assert float(la.detach()) == float(lb.detach())
        assert a.fc1.weight.grad is not None and a.fc1.bias.grad is not None
        assert b.fc1.weight.grad is None and b.fc1.bias.grad is None and b.ae.encoder.fc1.fc1.weight.grad is None
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            if k.endswith("fc1.fc1.bias"):      # a Linear bias in front of a train-mode BatchNorm: its gradient is exactly zero, in fp32 it is
                continue                        # the rounding noise of the column sum, and Adam steps +-lr on the SIGN of that noise (either path)
            d = float((p.detach() - q.detach()).abs().max() / p.detach().abs().max().clamp_min(1e-30))
            assert d <= 1e-6, (step, k, d)
        sa, sb = ta.optimizer.state[a.fc1.weight], tb.optimizer.state[b.fc1.weight]
        assert sa["step"] == sb["step"] == step + 1
        assert tb.optimizer.state[b.fc1.bias]["step"] == step + 1
    ta.close()
    tb.close()
    from driving_dirty_amd import ops
    assert not ops.RANKB


def test_rankb_two_backwards_before_the_step_add_up(dev):
    """Two backwards through a registered layer before one optimizer step (gradient accumulation): the factors are concatenated, the
    update equals the materialised path's on the summed gradient."""
    from driving_dirty_amd import ops
    from driving_dirty_amd.optim import HipAdam
    torch.manual_seed(5)
    lin_a, lin_b = torch.nn.Linear(64, 4096).to(dev), torch.nn.Linear(64, 4096).to(dev)
    lin_b.load_state_dict(lin_a.state_dict())
    oa, ob = HipAdam(lin_a.parameters(), lr=1e-2), HipAdam(lin_b.parameters(), lr=1e-2)
    assert len(ob.fuse_linear_wgrad(lin_b, min_numel=1024)) == 1
    xs = [torch.rand(8, 64, device=dev), torch.rand(5, 64, device=dev)]
    for lin, opt in ((lin_a, oa), (lin_b, ob)):
        for x in xs:
            ops.linear(x, lin.weight, lin.bias).square().sum().backward()
        opt.step()
    assert lin_b.weight.grad is None and lin_a.weight.grad is not None
    for p, q in zip(lin_a.parameters(), lin_b.parameters()):
        assert float((p - q).abs().max() / p.abs().max()) <= 1e-6
    ob.close()


def test_factor_mode_starts_the_input_gather_in_the_forward(dev):
    """ddp.GradSync factor mode: ``ops.Linear.forward`` starts the all-gather of a big layer input right away -- decided in
    ``ops.linear()``, outside the autograd Function (inside ``Function.forward`` grad mode is always off, so the earlier
    ``torch.is_grad_enabled()`` test there never fired) -- and the backward picks that gather up instead of starting its own.  One-rank gloo communicator, force_collectives."""
    import torch.distributed as dist
    from _ports import free_port
    from driving_dirty_amd import ddp, ops
    from driving_dirty_amd.optim import HipAdam
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(3)
        lin = torch.nn.Linear(4096, 16).to(dev)
        ref = torch.nn.Linear(4096, 16).to(dev)
        ref.load_state_dict(lin.state_dict())
        sync = ddp.GradSync(lin, big_numel=4096, force_collectives=True, factor_linear=True)
        opt = HipAdam(lin.parameters(), lr=1e-2)
        opt.attach(sync)
        assert sync.factor and lin.weight.data_ptr() in ddp.FACTOR_SYNC
        x = torch.rand(2, 4096, device=dev)                  # 8192 elements >= big_numel: gathered from the forward
        with torch.no_grad():
            ops.linear(x, lin.weight, lin.bias)              # torch.no_grad(): validation starts no gather
        assert not sync._xwork and sync.early_input_gathers == 0
        ops.linear(x, lin.weight, lin.bias).square().sum().backward()
        assert sync.early_input_gathers == 1 and not sync._xwork
        sync.finish()
        opt.step(grad_scale=sync.grad_scale)
        ropt = HipAdam(ref.parameters(), lr=1e-2)
        ops.linear(x, ref.weight, ref.bias).square().sum().backward()
        ropt.step()
        for p, q in zip(lin.parameters(), ref.parameters()):
            assert float((p - q).abs().max() / q.abs().max()) <= 1e-6
        with pytest.raises(RuntimeError, match="second backward"):      # persistent gather buffers are not overwritten silently
            ops.linear(x, lin.weight, lin.bias).square().sum().backward()
            ops.linear(x, lin.weight, lin.bias).square().sum().backward()
        sync.remove()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ the six strip convs in one launch
def _strip_reference(views64, ws, bs):
    """The pre-ReLU mosaic [B,32,3 th,2 tw] in fp64 exactly as spatial_bb/components.py:34-73 builds it (torch ops on the CPU).
    ws / bs: bl, fl, b, f, br, fr."""
    import torch.nn.functional as F
    v = [views64[:, i] for i in range(6)]
    bl = F.conv2d(v[3], ws[0], bs[0], stride=(3, 2))
    fl = F.conv2d(v[0], ws[1], bs[1], stride=(3, 2))
    b = F.conv2d(torch.rot90(v[4], 1, [2, 3]), ws[2], bs[2], stride=(3, 2), padding=1)
    f = F.conv2d(torch.rot90(v[1], 1, [3, 2]), ws[3], bs[3], stride=(3, 2), padding=1)
    br = F.conv2d(torch.flip(v[5], [2, 3]), ws[4], bs[4], stride=(3, 2))
    fr = F.conv2d(torch.flip(v[2], [2, 3]), ws[5], bs[5], stride=(3, 2))
    rows = [torch.cat(p, dim=3) for p in ((bl, fl), (b, f), (br, fr))]
    return torch.cat(rows, dim=2)


@pytest.mark.parametrize("h,w,batch", [(256, 306, 3), (64, 114, 2), (256, 306, 66)])
def test_strip6_forward_and_weight_gradient_against_fp64(dev, h, w, batch):
    """dd_strip6_fwd / dd_strip6_wgrad (all six strip convs of SpatialMappingCNN in one launch each way, rot90 / flip / mosaic tiling as
    index arithmetic) against the reference's own sequence of torch ops in fp64: forward 2e-6 of peak, weight and bias gradients 2e-5
    (sums over up to 66 x 86 x 129 pixels).  64 x 114: other tile sizes (22 x 33); batch 66: two launches, the second accumulating."""
    from driving_dirty_amd import gconv
    assert gconv.strip6_supported(h, w)
    g = torch.Generator().manual_seed(h + batch)
    if batch > 8:      # one distinct sample repeated: the reference conv on the CPU stays cheap, the batch loop of the kernel is still exercised
        one = torch.rand(2, 6, 3, h, w, generator=g)
        views = one[torch.arange(batch) % 2].contiguous()
    else:
        views = torch.rand(batch, 6, 3, h, w, generator=g)
    shapes = [(32, 3, 1, 50), (32, 3, 1, 50), (32, 3, 52, 1), (32, 3, 52, 1), (32, 3, 1, 50), (32, 3, 1, 50)]
    ws = [(torch.rand(s, generator=g) - 0.5) * 0.2 for s in shapes]
    bs = [(torch.rand(32, generator=g) - 0.5) * 0.2 for _ in shapes]
    th, tw = (h - 1) // 3 + 1, (w - 50) // 2 + 1
    got = gconv.strip6_fwd(views.to(dev), [x.to(dev) for x in ws], [x.to(dev) for x in bs])
    assert tuple(got.shape) == (batch, 3 * th, 2 * tw, 32)
    w64 = [x.double().requires_grad_(True) for x in ws]
    b64 = [x.double().requires_grad_(True) for x in bs]
    nref = batch if batch <= 8 else 2
    pre = _strip_reference(views[:nref].double(), w64, b64)                 # [nref,32,3th,2tw]
    ref = pre.relu().permute(0, 2, 3, 1)
    ref = ref.detach()
    peak = float(ref.abs().max())
    for i in range(batch):
        assert float((got[i].double().cpu() - ref[i % 2 if batch > 8 else i]).abs().max()) <= 2e-6 * peak, i
    gm = (torch.rand(batch, 3 * th, 2 * tw, 32, generator=g) - 0.5)
    if batch > 8:
        gm = gm[torch.arange(batch) % 2].contiguous()
    dws, dbs = gconv.strip6_wgrad(views.to(dev), gm.to(dev))
    (pre * gm[:nref].double().permute(0, 3, 1, 2)).sum().backward()
    mult = batch / nref if batch > 8 else 1.0                              # the repeated samples: the same gradient batch / 2 times
    for k in range(6):
        rw, rb = w64[k].grad * mult, b64[k].grad * mult
        assert float((dws[k].double().cpu() - rw).abs().max()) <= 2e-5 * float(rw.abs().max()), k
        assert float((dbs[k].double().cpu() - rb).abs().max()) <= 2e-5 * float(rb.abs().max()), k
    again = gconv.strip6_wgrad(views.to(dev), gm.to(dev))                   # fixed-order partial sums: bit-reproducible
    assert all(torch.equal(a, b) for a, b in zip(again[0] + again[1], dws + dbs))


def test_strip6_reads_every_input_form_alike(dev):
    """fp32 [B,6,3,H,W], the collate's tuple of [6,3,H,W], uint8 [B,6,H,W,3] frames and a tuple of [6,H,W,3]: the same mosaic and the same
    gradients bit for bit (ToTensor's /255 is fused as a true division; the fp32 views here are what ToTensor makes of the frames)."""
    from driving_dirty_amd import gconv
    g = torch.Generator().manual_seed(9)
    frames = torch.randint(0, 256, (3, 6, 256, 306, 3), dtype=torch.uint8, generator=g)
    views = frames.permute(0, 1, 4, 2, 3).float().div(255).contiguous()     # on the CPU: a true division (data_helper.py:63-68)
    shapes = [(32, 3, 1, 50), (32, 3, 1, 50), (32, 3, 52, 1), (32, 3, 52, 1), (32, 3, 1, 50), (32, 3, 1, 50)]
    ws = [((torch.rand(s, generator=g) - 0.5) * 0.2).to(dev) for s in shapes]
    bs = [((torch.rand(32, generator=g) - 0.5) * 0.2).to(dev) for _ in shapes]
    gm = (torch.rand(3, 258, 258, 32, generator=g) - 0.5).to(dev)
    forms = {"tensor": views.to(dev), "tuple": tuple(views.to(dev)), "u8": frames.to(dev), "u8_tuple": tuple(frames.to(dev))}
    outs = {k: (gconv.strip6_fwd(v, ws, bs), gconv.strip6_wgrad(v, gm)) for k, v in forms.items()}
    base = outs["tensor"]
    for k, (mo, (dws, dbs)) in outs.items():
        assert torch.equal(mo, base[0]), k
        assert all(torch.equal(a, b) for a, b in zip(dws + dbs, base[1][0] + base[1][1])), k


def test_spatial_mapping_cnn_fused_strips_equal_the_generic_engine(dev):
    """SpatialMappingCNN forward + backward with the one-launch strip kernels (default) and with gconv.STRIP6 off (the round-4 path: six
    NHWC4 re-layouts, six launches of the generic engine each way): same map, same gradients to summation order."""
    from driving_dirty_amd import gconv, synth
    from driving_dirty_amd.spatial import SpatialMappingCNN
    m = SpatialMappingCNN()
    synth.fill_module(m, seed=13)
    m = m.to(dev)
    x = synth.camera_batch(2, seed=13).to(dev)
    res = {}
    for fused in (True, False):
        prev, gconv.STRIP6 = gconv.STRIP6, fused
        try:
            m.zero_grad(set_to_none=True)
            y = m(x)
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            res[fused] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
        finally:
            gconv.STRIP6 = prev
    ya, ga = res[True]
    yb, gb = res[False]
    assert float((ya - yb).abs().max()) <= 2e-6 * float(yb.abs().max())
    for k in gb:
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * float(gb[k].abs().max()), k


def test_out_conv_on_the_winograd_kernels_equals_the_engine(dev):
    """SpatialMappingCNN forward + backward with out_conv (32 -> 32, k3, padding 0) on the c2 layer's Winograd F(2x2,3x3) kernels (default:
    the padding-1 convolution of the mosaic, whose interior is the layer's output, returned as a view) and on the dilated-conv engine
    (heads.WINO_OUT off): same map, same gradients to the transforms' summation order -- and the merging head takes the view where it lies."""
    from driving_dirty_amd import heads, synth
    from driving_dirty_amd.spatial import SpatialMappingCNN
    m = SpatialMappingCNN()
    synth.fill_module(m, seed=17)
    m = m.to(dev)
    x = synth.camera_batch(3, seed=17).to(dev)
    res = {}
    for wino in (True, False):
        prev, heads.WINO_OUT = heads.WINO_OUT, wino
        try:
            m.zero_grad(set_to_none=True)
            y = m(x)
            assert (heads.padded_nhwc(y.permute(0, 2, 3, 1)) is not None) == wino
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view(y.shape)).sum().backward()
            res[wino] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
        finally:
            heads.WINO_OUT = prev
    ya, ga = res[True]
    yb, gb = res[False]
    assert ya.shape == yb.shape and float((ya - yb).abs().max()) <= 5e-6 * float(yb.abs().max())
    assert torch.equal(ya > 0, yb > 0) or float(((ya > 0) != (yb > 0)).float().mean()) < 1e-5
    # two fp32 paths, each within the kernels' 2e-5 of the exact gradient (tests/test_gpu_heads.py holds either to the fp64 oracle): their
    # difference may reach the sum of the two; measured 3.6e-5 on the strip weights, whose gradient sums 65 k positions of the data gradient
    for k in gb:
        assert float((ga[k] - gb[k]).abs().max()) <= 6e-5 * float(gb[k].abs().max()), k


@pytest.mark.parametrize("dh,dw", [(16, 20), (128, 153)])
def test_decoder_dc2_on_the_winograd_kernels_equals_the_engine(dev, dh, dw):
    """DecoderConvStack with dc2 (ConvTranspose2d 32 -> 32, k3, padding 1 = a padding-1 convolution with transposed, flipped weights) on the
    c2 layer's Winograd kernels (default) and on the dilated-conv engine (heads.WINO_DC2 off): output and every gradient to summation order."""
    from driving_dirty_amd import heads
    torch.manual_seed(23)
    b = 2
    h = torch.randn(b, 64 * dh * dw, device=dev, requires_grad=True)
    ws = [torch.randn(64, 32, 3, 3, device=dev) * 0.06, torch.randn(32, device=dev) * 0.1, torch.randn(32, 32, 3, 3, device=dev) * 0.08,
          torch.randn(32, device=dev) * 0.1, torch.randn(32, 32, 2, 2, device=dev) * 0.1, torch.randn(32, device=dev) * 0.1,
          torch.randn(32, 3, 1, 1, device=dev) * 0.2, torch.randn(3, device=dev) * 0.1]
    for t in ws:
        t.requires_grad_(True)
    res = {}
    for wino in (True, False):
        prev, heads.WINO_DC2 = heads.WINO_DC2, wino
        try:
            for t in [h] + ws:
                t.grad = None
            y = heads.DecoderConvStack.apply(h, dh, dw, *ws)
            (y * torch.linspace(-1, 1, y.numel(), device=dev).view(y.shape)).sum().backward()
            res[wino] = (y.detach().clone(), [t.grad.detach().clone() for t in [h] + ws])
        finally:
            heads.WINO_DC2 = prev
    ya, ga = res[True]
    yb, gb = res[False]
    assert float((ya - yb).abs().max()) <= 5e-6 * float(yb.abs().max())
    for i, (a, r) in enumerate(zip(ga, gb)):
        assert float((a - r).abs().max()) <= 4e-5 * float(r.abs().max()), i


@pytest.mark.parametrize("b,h,w", [(2, 8, 11), (3, 16, 34), (1, 128, 918)])
def test_joint_feature_gradient_in_one_pass_equals_the_three(dev, b, h, w):
    """dd_pool4_relu_bwd_add: the c3 feature's gradient when the pool AND the box heads consume it, (feat > 0) * (gfeat + routed dpooled),
    bit for bit what dd_relu_bwd + dd_pool4_relu_bwd + dd_add give (ties to the first index, zero maxima closed), ragged last tiles."""
    from driving_dirty_amd import ops
    torch.manual_seed(b * 100 + w)
    feat = torch.relu(torch.randn(b, h, w, 32, device=dev))
    feat = (feat * 4).round() / 4                       # ties and all-zero windows
    gfeat = torch.randn_like(feat)
    dpooled = torch.randn(b, 32 * h * w // 4, device=dev)
    want = ops.add(ops.relu_bwd(gfeat, feat), ops.pool4_relu_bwd(dpooled, feat))
    got = ops.pool4_relu_bwd_add(dpooled, feat, gfeat)
    assert torch.equal(got, want)


@pytest.mark.parametrize("b,sh,sw", [(2, 40, 46), (1, 268, 268), (3, 19, 33)])
def test_phase_major_rm_conv_1_and_the_scatter_gather_pair(dev, b, sh, sw):
    """dd_conv1ch_fwd_phase3 / dd_conv1ch_wgrad_phase3: rm_conv_1's output and gradient in the nine-residue-class layout -- the dense
    kernels' values bit for bit in their cells, zeros (and zero sign words) in the padding cells; dd_phase3_scatter / dd_phase3_gather are
    inverse on the cells that map to pixels and write zeros elsewhere."""
    from driving_dirty_amd import heads, ops
    torch.manual_seed(sh)
    rm4 = torch.zeros(b, sh, sw, 4, device=dev)
    rm4[..., 0] = torch.rand(b, sh, sw, device=dev)
    w = torch.randn(32, 1, 7, 7, device=dev) * 0.2
    bias = torch.randn(32, device=dev) * 0.2
    dense = ops.conv1ch_fwd(rm4, w, bias, relu=True)
    oh, ow = dense.shape[1:3]
    yp, bits = ops.conv1ch_fwd_phase3(rm4, w, bias, relu=True)
    ph, pw = yp.shape[1:3]
    assert (ph, pw) == ((oh + 2) // 3, (ow + 2) // 3)
    assert torch.equal(heads.MergeFn._dense_from_phase3(yp, oh, ow), dense)
    pv = yp.view(b, 3, 3, ph, pw, 32)
    for a in range(3):
        for c in range(3):
            na, nc = (oh - a + 2) // 3, (ow - c + 2) // 3
            assert float(pv[:, a, c, na:].abs().sum()) == 0.0 and float(pv[:, a, c, :, nc:].abs().sum()) == 0.0
    assert torch.equal(bits, ops.relu_sign_bits(yp))
    g = torch.randn_like(dense)
    gp = ops.phase3_gather(g, 0, ph, pw, 0)
    assert torch.equal(heads.MergeFn._dense_from_phase3(gp, oh, ow), g)
    dw, db = ops.conv1ch_wgrad(rm4, g)
    dwp, dbp = ops.conv1ch_wgrad_phase3(rm4, gp)
    assert torch.equal(dwp, dw) and torch.equal(dbp, db)
    # scatter with an offset into a channel slice, gather back
    wide = torch.full((b, oh, ow, 96), -2.0, device=dev)
    padded = torch.randn(9 * b, ph + 2, pw + 2, 32, device=dev)
    ops.phase3_scatter(padded, wide, 64, 1)
    assert torch.equal(wide[..., 64:], heads.MergeFn._dense_from_phase3(padded[:, 1:, 1:].contiguous(), oh, ow)) and float(wide[..., :64].max()) == -2.0
    back = ops.phase3_gather(wide, 64, ph + 2, pw + 2, 1)
    assert float(back[:, 0].abs().sum()) == 0.0 and float(back[:, :, 0].abs().sum()) == 0.0
    assert torch.equal(heads.MergeFn._dense_from_phase3(back[:, 1:, 1:].contiguous(), oh, ow), wide[..., 64:])


def test_rm_conv_2_on_the_winograd_kernels_equals_the_engine(dev):
    """RoadMapBoxesMergingCNN forward + backward with rm_conv_2 as nine plain 3x3 convolutions on phase images (default) and on the
    dilated-conv engine (heads.WINO_RM2 off): same probabilities, same gradients to summation order."""
    from driving_dirty_amd import heads, synth
    from driving_dirty_amd.spatial import RoadMapBoxesMergingCNN
    m = RoadMapBoxesMergingCNN()
    synth.fill_module(m, seed=29)
    m = m.to(dev)
    torch.manual_seed(4)
    ssr = torch.rand(2, 32, 128, 918, device=dev)
    space = torch.rand(2, 32, 256, 256, device=dev)
    rm = (torch.rand(2, 1, 800, 800, device=dev) < 0.3).float()
    res = {}
    for wino in (True, False):
        prev, heads.WINO_RM2 = heads.WINO_RM2, wino
        try:
            m.zero_grad(set_to_none=True)
            y = m(ssr, space, rm)
            # (positive weights: with a zero-mean weighting the last layer's bias gradient -- one number, the sum of 1.3 M terms -- cancels to
            # 1e-4 of its terms and its RELATIVE difference between two fp32 paths says nothing)
            (y * torch.linspace(0.5, 1.5, y.numel(), device=dev).view(y.shape)).sum().backward()
            res[wino] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
        finally:
            heads.WINO_RM2 = prev
    ya, ga = res[True]
    yb, gb = res[False]
    assert float((ya - yb).abs().max()) <= 5e-6 * float(yb.abs().max())
    for k in gb:      # two fp32 paths: their difference may reach the sum of the two kernel-level bounds
        assert float((ga[k] - gb[k]).abs().max()) <= 6e-5 * float(gb[k].abs().max()), k


def test_sign_words_padded_relu_backward_and_window_copy(dev):
    """The three helpers behind it: dd_relu_sign_bits, dd_relu_bwd_pad_bits, dd_copy_channels_window against torch, ragged sizes."""
    from driving_dirty_amd import heads, ops
    from driving_dirty_amd.gconv import View, copy_channels
    torch.manual_seed(3)
    for b, h, w in ((2, 5, 7), (3, 19, 33)):
        x = torch.randn(b, h + 2, w + 2, 32, device=dev)
        x[0, 0, 0, :] = 0.0                                   # zero is not positive
        bits = ops.relu_sign_bits(x)
        ref = ((x > 0).to(torch.int64) << torch.arange(32, device=dev)).sum(-1)
        assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, ref)
        dy = torch.randn(b, h, w, 32, device=dev)
        got = ops.relu_bwd_pad_bits(dy, bits)
        want = torch.zeros_like(x)
        want[:, 1:-1, 1:-1] = dy * (x[:, 1:-1, 1:-1] > 0)
        assert torch.equal(got, want)
        wide = torch.randn(b, h, w, 96, device=dev)           # the same from a channel slice of a wider buffer, read where it lies
        wide[..., 32:64] = dy
        assert ops.channel_slice(wide[..., 32:64])[1] == 32 and torch.equal(ops.relu_bwd_pad_bits(wide[..., 32:64], bits), want)
        dst = torch.full((b, h, w, 64), -1.0, device=dev)
        inner = x[:, 1:-1, 1:-1, :]
        base = heads.padded_nhwc(inner)
        assert base is not None and base[1:] == (1, 1) and base[0].data_ptr() == x.data_ptr() and tuple(base[0].shape) == tuple(x.shape)
        copy_channels(View(base[0], 0, 32, 1, 1, h, w), View(dst, 32, 32))
        assert torch.equal(dst[..., 32:], inner) and float(dst[..., :32].max()) == -1.0
    assert heads.padded_nhwc(torch.zeros(2, 4, 4, 32, device=dev)) is None
    assert heads.padded_nhwc(torch.zeros(2, 4, 4, 64, device=dev)[..., :32]) is None      # a channel slice is not a spatial window


def test_factor_mode_over_64_gathered_rows_falls_back_to_the_materialised_gradient(dev):
    """dd_adam_step_rankb takes at most 64 rows.  In ddp.GradSync factor mode the rows are world x batch: past 64 the optimizer forms the
    gathered gradient with dd_linear_wgrad as in round 4, and ``factor_bias`` tells ops.Linear.backward that the bias gradient is still
    owed (a stub stands in for the GradSync: the decision and the fallback are the optimizer's)."""
    from driving_dirty_amd import ddp
    from driving_dirty_amd.optim import HipAdam
    torch.manual_seed(2)
    lin = torch.nn.Linear(64, 4096).to(dev)
    ref = torch.nn.Linear(64, 4096).to(dev)
    ref.load_state_dict(lin.state_dict())
    opt = HipAdam(lin.parameters(), lr=1e-2)
    assert len(opt.fuse_linear_wgrad(lin, min_numel=1024)) == 1
    rows = 80
    x, dy = torch.rand(rows, 64, device=dev), torch.rand(rows, 4096, device=dev) - 0.5

    class Stub:
        factor, active, shard = True, True, False

        def take_factors(self, p):
            return ddp.Factors([], x, dy, rows)

        def has_factors(self, p):
            return False
    opt._sync = Stub()
    assert opt.factor_bias(lin.weight, 64) and not opt.factor_bias(lin.weight, rows)
    assert opt._update_factored(lin.weight, opt.param_groups[0], 0.5)
    assert lin.weight.grad is not None                         # the materialised path
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    ref.weight.grad = (dy.t() @ x) * 0.5
    ref.bias.grad = torch.zeros_like(ref.bias)
    ropt.step()
    assert float((lin.weight - ref.weight).abs().max() / ref.weight.abs().max()) <= 1e-6
    opt._sync = None
    opt.close()
