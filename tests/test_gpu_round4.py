"""GPU tests, fourth set: the sharded optimizer on the real Adam kernel (two and four ranks on one card over gloo), the
``TrainStep`` helper against ``configure_optimizers()``'s torch Adam, the N > 1 rehearsals of ``bench.py`` for configs 3 / 4 / 5 and
of the reduce-scatter / all-gather call pattern on a one-rank RCCL communicator."""
import json
import os
import subprocess
import sys
from argparse import Namespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from driving_dirty_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _tiny_model(dev, frozen_epochs=0):
    from driving_dirty_amd import synth
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    ae = BasicAE(Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132))
    model = RoadMapBCE(Namespace(pretrained_ae=ae, unfreeze_epoch_no=frozen_epochs, learning_rate=1e-2, output_img_freq=500))
    synth.fill_module(model, seed=77)
    model = model.to(dev)
    model.ae.encoder.fc1.drop_p = model.ae.encoder.fc2.drop_p = 0.0
    return model


def _tiny_batch(dev, step, rank):
    from driving_dirty_amd import synth
    views = synth.camera_batch(3, 16, 22, seed=100 + 10 * step + rank).to(dev)
    road = synth.road_maps(3, seed=100 + 10 * step + rank).to(dev)
    return (tuple(views), None, tuple(road))


# ------------------------------------------------------------------------------------------------ sharded optimizer, real kernel
def _shard_worker(rank, world, port, out, shards, overlap):
    """One process per rank runs the modes in ``shards`` one after the other (one spawn, one process group: the second spawn of a test was
    3-4 s of interpreter, HIP and gloo start-up per parametrisation)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for shard in shards:
        _shard_body(rank, world, out, shard, overlap)
        dist.barrier()
    dist.destroy_process_group()


def _shard_body(rank, world, out, shard, overlap):
    from driving_dirty_amd import ddp
    from driving_dirty_amd.train import TrainStep
    dev = torch.device("cuda:0")
    model = _tiny_model(dev, frozen_epochs=1)                 # the extractor stays frozen through epoch 0 (roadmap_bce_v2.py:127-129)
    # head fc1.weight: 640000 x 8 = 5.12 M elements; chunk 1 << 20 -> 5 pieces; encoder fc1.fc1.weight 16 x 1056 = 16,896 -> big too
    ts = TrainStep(model, lr=1e-2, adam_overlap=overlap, shard_optimizer=shard, big_numel=4096, chunk_numel=1 << 20, scheduler=False)
    assert ts.sync.shard == shard
    for step in range(4):
        if step == 1:
            model.current_epoch = 1                           # training_step unfreezes the extractor: GradSync / HipAdam re-arm
        ts(_tiny_batch(dev, step, rank), step)
        if shard:
            sh = ts.sync.shards(model.fc1.weight)
            assert sh is not None and len(sh) == 5
            if step >= 1:
                assert ts.sync.shards(model.ae.encoder.fc1.fc1.weight) is not None
    if shard:
        assert ddp.PARAM_WAITS or world == 1                  # the last step's all-gathers are still registered ...
    ts.sync_params()
    assert not ddp.PARAM_WAITS                                # ... and gone once waited for
    if shard:
        st = ts.optimizer.state[model.fc1.weight]
        assert "exp_avg" not in st and sum(m.numel() for m, _ in st["shards"].values()) * world == model.fc1.weight.numel()
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, f"{out}.{int(shard)}.{rank}")
    ts.close()


@pytest.mark.parametrize("world,overlap", [(2, True), (2, False), (4, True)])
def test_sharded_hipadam_is_bit_identical_to_the_all_reduce_path(tmp_path, dev, world, overlap):
    """ddp.GradSync(shard_optimizer=True) + optim.HipAdam on the real kernel, `world` ranks on one card over gloo, a frozen extractor
    unfrozen after step 0, Adam beside the backward or after it: every replica holds bit for bit the parameters the all-reduce path
    leaves (reduce-scatter over gloo adds in its all-reduce's order; the update is elementwise)."""
    out = str(tmp_path / "s.pt")
    mp.spawn(_shard_worker, args=(world, free_port(), out, (False, True), overlap), nprocs=world, join=True)
    ref = torch.load(f"{out}.0.0")
    start = {k: v.cpu() for k, v in _tiny_model(dev, 1).state_dict().items()}
    assert float((ref["ae.encoder.c2.weight"] - start["ae.encoder.c2.weight"]).abs().max()) > 0      # the unfrozen extractor trained
    assert float((ref["fc1.weight"] - start["fc1.weight"]).abs().max()) > 0
    buffers = {k for k in ref if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}      # BatchNorm statistics stay per-rank
    for rank in range(world):
        plain, sharded = torch.load(f"{out}.0.{rank}"), torch.load(f"{out}.1.{rank}")
        for k in ref:
            assert torch.equal(sharded[k], plain[k]), f"rank {rank}: {k} differs between the sharded and the all-reduce run"
            if k not in buffers:
                assert torch.equal(plain[k], ref[k]), f"rank {rank}: parameter {k} differs from rank 0's"


def _factor_worker(rank, world, port, out, modes, overlap):
    """``modes``: (factor, fuse) pairs run one after the other in one process per rank (one spawn, one process group)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for factor, fuse in modes:
        _factor_body(rank, world, out, factor, overlap, fuse)
        dist.barrier()
    dist.destroy_process_group()


def _factor_body(rank, world, out, factor, overlap, fuse):
    from driving_dirty_amd import ddp
    from driving_dirty_amd.train import TrainStep
    dev = torch.device("cuda:0")
    model = _tiny_model(dev, frozen_epochs=1)
    # fuse: rank-B mode of the optimizer -- the gathered factors go straight into dd_adam_step_rankb, no gradient tensor is formed
    ts = TrainStep(model, lr=1e-2, adam_overlap=overlap, factor_linear=factor, big_numel=4096, chunk_numel=1 << 20, scheduler=False,
                   fuse_linear_wgrad=fuse)
    assert ts.sync.factor == factor and bool(ddp.FACTOR_SYNC) == factor
    big = {"fc1.weight": model.fc1.weight, "ae.encoder.fc1.fc1.weight": model.ae.encoder.fc1.fc1.weight}
    grads, losses = {}, []
    for step in range(3):
        if step == 1:
            model.current_epoch = 1                           # the extractor (and its fc1) joins: its weight is registered at the unfreeze
        losses.append(float(ts(_tiny_batch(dev, step, rank), step)["loss"]))
        for name, q in big.items():
            if q.grad is not None:
                grads[f"{name}.{step}"] = q.grad.detach().cpu().clone()      # the SUM over ranks in both modes (1 / world is folded into Adam)
    if factor:
        assert set(ddp.FACTOR_SYNC) == {q.data_ptr() for q in big.values()}
    torch.cuda.synchronize()
    torch.save({"state": {k: v.cpu() for k, v in model.state_dict().items()}, "grads": grads, "losses": losses}, f"{out}.{int(factor)}.{rank}")
    ts.close()
    assert not ddp.FACTOR_SYNC


@pytest.mark.parametrize("world,overlap", [(2, True), (2, False)])
def test_factor_gather_with_the_rankb_pass_keeps_the_replicas_identical(tmp_path, dev, world, overlap):
    """Factor gather + rank-B optimizer pass (TrainStep's defaults at N = 2): the gathered factors feed dd_adam_step_rankb directly, no
    gradient tensor exists for the two big Linear layers (``.grad`` stays None, bias included).  Replicas bit-identical to each other;
    losses of three steps equal to the all-reduce path's to 1e-4."""
    out = str(tmp_path / "f.pt")
    mp.spawn(_factor_worker, args=(world, free_port(), out, ((False, False), (True, True)), overlap), nprocs=world, join=True)
    plain = [torch.load(f"{out}.0.{r}") for r in range(world)]
    fact = [torch.load(f"{out}.1.{r}") for r in range(world)]
    assert len(plain[0]["grads"]) == 5 and not fact[0]["grads"]      # no .grad on the fused layers, ever
    buffers = {k for k in fact[0]["state"] if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    for r in range(1, world):
        for k, v in fact[0]["state"].items():
            if k not in buffers:
                assert torch.equal(fact[r]["state"][k], v), f"rank {r}: parameter {k} differs from rank 0's"
    moved = float((fact[0]["state"]["fc1.weight"] - _tiny_model(dev, 1).state_dict()["fc1.weight"].cpu()).abs().max())
    assert moved > 0
    for a, b in zip(fact[0]["losses"], plain[0]["losses"]):
        assert abs(a - b) <= 1e-4 * abs(b)


@pytest.mark.parametrize("world,overlap", [(2, True), (3, False)])
def test_factor_gather_forms_the_global_batch_gradient_on_every_rank(tmp_path, dev, world, overlap):
    """ddp.GradSync(factor_linear=True): the two big Linear layers all-gather (input, output gradient) instead of all-reducing their
    weight gradients, HipAdam forms dY_all^T X_all on every rank (beside the backward or after it).  The gradient equals the
    all-reduce path's sum to rounding (one GEMM over the global batch against a sum of per-rank GEMMs), the replicas are bit-identical
    to each other, the losses of three steps agree."""
    out = str(tmp_path / "f.pt")
    mp.spawn(_factor_worker, args=(world, free_port(), out, ((False, False), (True, False)), overlap), nprocs=world, join=True)
    plain = [torch.load(f"{out}.0.{r}") for r in range(world)]
    fact = [torch.load(f"{out}.1.{r}") for r in range(world)]
    assert set(fact[0]["grads"]) == set(plain[0]["grads"]) and len(fact[0]["grads"]) == 5      # head: 3 steps, encoder fc1: after the unfreeze
    for key, g in fact[0]["grads"].items():      # step 0: same parameters in both runs; later the two runs have taken Adam steps apart
        ref = plain[0]["grads"][key].double()
        tol = 2e-6 if key.endswith(".0") else 1e-3
        assert (g.double() - ref).abs().max().item() <= tol * ref.abs().max().item(), key
    buffers = {k for k in fact[0]["state"] if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    for r in range(1, world):
        for k, v in fact[0]["state"].items():
            if k not in buffers:
                assert torch.equal(fact[r]["state"][k], v), f"rank {r}: parameter {k} differs from rank 0's"
    for a, b in zip(fact[0]["losses"], plain[0]["losses"]):
        assert abs(a - b) <= 1e-4 * abs(b)


# ------------------------------------------------------------------------------------------------ TrainStep == the reference's loop
def test_trainstep_matches_configure_optimizers_loop(dev):
    """Two steps through train.TrainStep (HipAdam beside the backward, ReduceLROnPlateau attached) leave the parameters that two steps
    of the reference's loop -- zero_grad / training_step / backward / step with configure_optimizers()'s torch.optim.Adam
    (roadmap_bce_v2.py:154-157) -- leave, within the Adam kernel's own 1e-6."""
    from driving_dirty_amd.train import TrainStep
    a, b = _tiny_model(dev), _tiny_model(dev)
    (opt,), (sched,) = a.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau)
    ts = TrainStep(b)                                         # lr from hparams.learning_rate, scheduler because the module returns one
    assert ts.scheduler is not None and ts.lr == 1e-2
    # step 0: both models hold the same parameters, the same kernels give the same gradients, only the optimizer differs: 1e-6.
    # Before step 1 the torch-Adam model takes over the TrainStep model's parameters (they differ by that rounding, 2e-7 of an update,
    # and Adam's m / sqrt(v) turns the resulting relative change of a near-zero gradient element into the same relative change of a
    # full-size update: measured 9e-4 of peak on c2.weight); each optimizer keeps its OWN moments and step count from step 0, so step 1
    # still compares the two optimizers' second steps -- on identical gradients.
    for step, tol in ((0, 1e-6), (1, 1e-6)):
        if step == 1:
            with torch.no_grad():
                for p, q in zip(a.parameters(), b.parameters()):
                    p.copy_(q)
            for (_, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
                u.copy_(v)
        batch = _tiny_batch(dev, step, 0)
        opt.zero_grad()
        la = a.training_step(batch, step)["loss"]
        la.backward()
        opt.step()
        lb = ts(batch, step)["loss"]
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la)), (step, float(la), float(lb))
        ts.sync_params()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            d = float((p.detach() - q.detach()).abs().max() / p.detach().abs().max().clamp_min(1e-30))
            assert d <= tol, (step, k, d)
    # the plateau scheduler: 11 epochs without improvement cut the rate by 10 (patience 10), on both optimizers alike
    for _ in range(12):
        sched.step(1.0)
        ts.validation_epoch_end(1.0)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-3) and ts.lr == pytest.approx(1e-3)
    ts.close()


def test_trainstep_has_no_scheduler_for_the_autoencoder(dev):
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.train import TrainStep
    ae = BasicAE(Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132, output_height=16, output_width=22,
                           learning_rate=1e-3, output_img_freq=500)).to(dev)
    ts = TrainStep(ae)
    assert ts.scheduler is None                               # autoencoder.py:119-120 returns the bare optimizer
    ts.close()


# ------------------------------------------------------------------------------------------------ bench.py rehearsals
def _bench(env_extra, *args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--no-others", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"bench.py owes its caller ONE line on stdout, got {len(lines)}: {[ln[:60] for ln in lines]}"      # e.g. RCCL's banner at NCCL_DEBUG=WARN
    return json.loads(lines[0]), r.stderr


def test_bench_sharded_step_over_a_one_rank_rccl_communicator(dev):
    """The sharded step's call pattern on the real backend: RCCL reduce-scatter per gradient piece from the autograd hooks, Adam on
    the owned slices on the side stream, in-place all-gather behind each, the wait where the next forward first touches the
    parameter (1-rank collectives are copies: the loss must equal the replicated step's)."""
    line, err = _bench({"DD_REHEARSE_RCCL": "1", "MASTER_PORT": str(free_port())}, "--steps", "3", "--warmup", "2",
                       "--shard-optimizer", "on")
    assert line["n_ranks_seen"] == 1 and "rehearsal" in line and line["config"]["optimizer"].startswith("sharded")
    assert line["preflight"]["backend"] == "nccl" and "rccl_version" in line["preflight"] and "bench.py preflight:" in err
    plain, _ = _bench({"DD_REHEARSE_RCCL": "1", "MASTER_PORT": str(free_port())}, "--steps", "3", "--warmup", "2",
                      "--shard-optimizer", "off")
    assert plain["config"]["optimizer"] == "replicated"
    assert abs(line["config"]["final_loss"] - plain["config"]["final_loss"]) <= 2e-6 * abs(plain["config"]["final_loss"])


def test_bench_factor_gather_over_a_one_rank_rccl_communicator(dev):
    """The factor gather's call pattern on the real backend: RCCL all-gathers of the two Linear layers' inputs and output gradients from
    inside the backward, the weight-gradient kernel and the Adam pass behind them on the side stream (a 1-rank gather is a copy and the
    gradient formed from the gathered factors is the local one: the loss must equal the plain step's)."""
    line, _ = _bench({"DD_REHEARSE_RCCL": "1", "MASTER_PORT": str(free_port())}, "--steps", "3", "--warmup", "2", "--factor-linear", "on")
    assert line["n_ranks_seen"] == 1 and "rehearsal" in line and line["config"]["optimizer"].startswith("replicated; the big Linear layers all-gather")
    plain, _ = _bench({}, "--steps", "3", "--warmup", "2")
    assert abs(line["config"]["final_loss"] - plain["config"]["final_loss"]) <= 2e-6 * abs(plain["config"]["final_loss"])


@pytest.mark.parametrize("config", [3, 4, 5])
def test_bench_two_ranks_over_gloo_on_one_card(dev, config):
    """`bench.py --gpus 2 --config C` starting its own ranks, both on this card, gradients over gloo (DD_DIST_BACKEND=gloo): the N > 1
    control flow of every BASELINE configuration that exists only on several GPUs -- config 4 on the sharded optimizer, config 5 on the
    factor gather (the N = 2 default)."""
    # N = 2 defaults to the factor gather (config 5 here); config 4: the sharded optimizer, without the all-reduce step in front of it (the
    # two-mode line is asserted on config 5 here and on config 2 under torch.distributed.run below; 3 s of a 64 ms step saved)
    extra = ("--factor-linear", "off", "--alt-all-reduce", "off") if config == 4 else ()
    line, err = _bench({"DD_DIST_BACKEND": "gloo", "DD_RESERVED_CUS": "0"}, "--gpus", "2", "--config", str(config), "--steps", "2", "--warmup", "1", *extra)
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["config"]["baseline_config"] == config
    assert line["config"]["optimizer"].startswith({3: "replicated", 4: "sharded", 5: "replicated; the big Linear layers all-gather"}[config])
    assert line["value"] > 0 and line["config"]["final_loss"] == line["config"]["final_loss"]
    assert line["roofline"] is not None and line["roofline"]["frac"] > 0
    assert "[rank 0] bench.py preflight:" in err and line["collective_probe"]["ms"] > 0
    if config == 5:
        assert line["config"]["alt_all_reduce"]["ms_per_step"] > 0 and "[rank 0] bench.py alt_all_reduce:" in err
    elif config == 3:
        assert line["config"]["alt_all_reduce_ms"] == line["ms_per_step"]      # the default mode IS the plain all-reduce


def test_bench_under_torch_distributed_run_as_the_driver_launches_it(dev):
    """The driver's N > 1 command line, verbatim, for the headline configuration: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W` -- both ranks on this card,
    gradients over gloo.  One JSON line on stdout (rank 0's), the N = 2 default (factor gather) in it, a roofline and a preflight."""
    env = dict(os.environ, DD_DIST_BACKEND="gloo", DD_RESERVED_CUS="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, [ln[:80] for ln in r.stdout.splitlines()]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 2
    assert line["config"]["baseline_config"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 64
    assert line["config"]["optimizer"].startswith("replicated; the big Linear layers all-gather")
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["scaling"] == "weak"
    assert "preflight" in line and "cpu_baseline" not in line          # the CPU baseline is timed at N = 1 only
    # the un-losable part: the plain all-reduce step timed FIRST and carried in the line, its figure on stderr before the
    # default mode starts, the communicator's measured bus bandwidth beside it; the default mode starts from the same parameters
    alt = line["config"]["alt_all_reduce"]
    assert line["config"]["alt_all_reduce_ms"] == alt["ms_per_step"] > 0 and alt["n_ranks_seen"] == 2 and alt["value"] > 0
    assert alt["optimizer"].startswith("replicated (plain all-reduce")
    assert "bench.py alt_all_reduce: " in r.stderr and "bench.py collective probe: " in r.stderr
    probe = line["collective_probe"]
    assert probe["message_bytes"] > 0 and probe["ms"] > 0 and probe["busbw_GBps"] > 0
    assert abs(alt["final_loss"] - line["config"]["final_loss"]) <= 1e-3 * abs(alt["final_loss"])      # same start, same batches, same steps
    assert line["config"]["linear_wgrad"].startswith("formed inside the Adam pass")      # factor gather + rank-B: gathered factors -> the pass


def test_bench_watchdog_prints_the_all_reduce_line_when_the_default_mode_stalls(dev):
    """A rank that hangs in the DEFAULT mode's timed region (fault injection at a step index only that phase reaches is not possible --
    both phases count from 0 -- so the hang is injected by step AND phase): the watchdog fires, and rank 0's last words are a complete
    JSON line for the all-reduce step that did run, marked `default_mode_failed`."""
    env = dict(os.environ, DD_DIST_BACKEND="gloo", DD_RESERVED_CUS="0", PYTHONPATH=ROOT, DD_WATCHDOG_S="10", DD_BENCH_FAULT="1:3:hang:default")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-others", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout[-500:], r.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["config"]["default_mode_failed"].startswith("watchdog") and line["value"] > 0 and line["n_gpus"] == 2
    assert line["config"]["optimizer"].startswith("replicated (plain all-reduce") and line["collective_probe"]["ms"] > 0


def test_simulated_shard_step_runs(dev):
    """`--simulate-shard 8`: the compute side of an 8-GPU sharded step on this one GPU (a timing aid; labelled as such)."""
    line, _ = _bench({}, "--steps", "3", "--warmup", "2", "--simulate-shard", "8")
    assert "simulated" in line and line["config"]["optimizer"].startswith("sharded") and line["value"] > 0


# ------------------------------------------------------------------------------------------------ uint8 frames end to end (row f4)
def _frames(b, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (b, 6, h, w, 3), dtype=torch.uint8, generator=g)


def _as_views(frames):
    """What the reference's dataset makes of decoded frames: ToTensor per camera (HWC uint8 -> CHW float / 255), stacked
    (data_helper.py:63-68) -- ON THE CPU, as in a DataLoader worker: there torch divides; on a GPU tensor `x / 255` is computed as
    x * (1 / 255), one ulp off for some of the 256 values."""
    return frames.cpu().permute(0, 1, 4, 2, 3).float().div(255).contiguous().to(frames.device)


def _grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _same(ga, gb, tol=2e-5):
    assert ga.keys() == gb.keys() and len(ga) > 0
    for k in ga:
        d = float((ga[k] - gb[k]).abs().max() / ga[k].abs().max().clamp_min(1e-30))
        assert d <= tol, (k, d)


def _full_ae(dev, hidden=128, latent=64):
    from driving_dirty_amd.autoencoder import BasicAE
    torch.manual_seed(5)
    return BasicAE(Namespace(hidden_dim=hidden, latent_dim=latent, learning_rate=1e-3, output_img_freq=10 ** 9))


@pytest.mark.parametrize("form", ["tuple", "tensor"])
def test_roadmap_step_on_uint8_frames_equals_the_fp32_step(dev, form):
    """RoadMapBCE.training_step at 256 x 306, B = 32 on uint8 frames (the collate's tuple of [6,H,W,3] or one [B,6,H,W,3] tensor)
    against the same step on the fp32 views ToTensor makes of them: loss 1e-6, every gradient 2e-5."""
    from driving_dirty_amd.roadmap import RoadMapBCE
    b = 32
    frames = _frames(b, 256, 306, 1).to(dev)
    road = tuple((torch.rand(b, 800, 800, generator=torch.Generator().manual_seed(2)) < 0.3).to(dev))
    model = RoadMapBCE(Namespace(pretrained_ae=_full_ae(dev), unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=10 ** 9)).to(dev)
    model.ae.encoder.fc1.drop_p = model.ae.encoder.fc2.drop_p = 0.0
    out = {}
    for kind in ("fp32", "u8"):
        sample = tuple(_as_views(frames)) if kind == "fp32" else (tuple(frames) if form == "tuple" else frames)
        model.zero_grad(set_to_none=True)
        loss = model.training_step((sample, None, road), 0)["loss"]
        loss.backward()
        out[kind] = (float(loss), _grads(model))
    assert abs(out["u8"][0] - out["fp32"][0]) <= 1e-6 * abs(out["fp32"][0])
    _same(out["fp32"][1], out["u8"][1])


def test_autoencoder_step_on_uint8_frames_equals_the_fp32_step(dev):
    """BasicAE.training_step (masked-view task included: the same np.random draw, the blanked slot and the target view taken from
    the uint8 frames) at 256 x 306, B = 32."""
    import numpy as np
    b = 32
    frames = _frames(b, 256, 306, 3).to(dev)
    model = _full_ae(dev).to(dev)
    for blk in (model.encoder.fc1, model.encoder.fc2, model.decoder.fc1, model.decoder.fc2):
        blk.drop_p = 0.0
    out = {}
    for kind in ("fp32", "u8"):
        np.random.seed(20200505)
        model.zero_grad(set_to_none=True)
        loss = model.training_step(_as_views(frames) if kind == "fp32" else frames, 0)["loss"]
        loss.backward()
        out[kind] = (float(loss), _grads(model))
    assert abs(out["u8"][0] - out["fp32"][0]) <= 1e-6 * abs(out["fp32"][0])
    _same(out["fp32"][1], out["u8"][1])
    np.random.seed(7)
    wide_a, y_a = model.six_to_one_task(_as_views(frames[:2]))
    np.random.seed(7)
    wide_b, y_b = model.six_to_one_task(frames[:2])
    assert torch.equal(wide_a, wide_b) and torch.equal(y_a, y_b)


@pytest.mark.parametrize("cls", ["spatial", "joint"])
def test_box_and_joint_steps_on_uint8_frames_equal_the_fp32_steps(dev, cls):
    """BBSpatialRoadMap (frozen encoder) and JointRoadMapBBox at 256 x 306, B = 32: SpatialMappingCNN reads its six views from
    the uint8 frames (dd_view_to_nhwc4_u8_ptrs), the encoder its wide image (dd_stitch6_u8_ptrs)."""
    from driving_dirty_amd.joint import JointRoadMapBBox
    from driving_dirty_amd.spatial import BBSpatialRoadMap
    b = 32
    frames = _frames(b, 256, 306, 4).to(dev)
    g = torch.Generator().manual_seed(6)
    road = tuple((torch.rand(b, 800, 800, generator=g) < 0.3).to(dev))
    tgt = tuple({"bb_map": (torch.rand(800, 800, generator=g) < 0.02).float().to(dev)} for _ in range(b))
    if cls == "spatial":
        model = BBSpatialRoadMap(Namespace(pretrained_ae=_full_ae(dev), unfreeze_epoch_no=10 ** 9, learning_rate=1e-3, output_img_freq=10 ** 9,
                                           mse_loss=False)).to(dev)
    else:
        model = JointRoadMapBBox(Namespace(pretrained_ae=_full_ae(dev), learning_rate=1e-3, output_img_freq=10 ** 9)).to(dev)
        model.ae.encoder.fc1.drop_p = model.ae.encoder.fc2.drop_p = 0.0
    out = {}
    for kind in ("fp32", "u8"):
        sample = tuple(_as_views(frames)) if kind == "fp32" else tuple(frames)
        model.zero_grad(set_to_none=True)
        loss = model.training_step((sample, tgt, road), 0)["loss"]
        loss.backward()
        out[kind] = (float(loss), _grads(model))
    assert abs(out["u8"][0] - out["fp32"][0]) <= 1e-6 * abs(out["fp32"][0])
    _same(out["fp32"][1], out["u8"][1])


def test_uint8_wide_images_are_bit_identical_to_totensor(dev):
    """The three uint8 entry points against torch on frames / 255 (a true division, as ToTensor's): fp32 wide image, bf16 wide image,
    every (view, transform) SpatialMappingCNN uses -- ragged size, B > 64 (two launches of the pointer table)."""
    from driving_dirty_amd import gconv, ops, ops_bf16
    frames = _frames(70, 9, 13, 9).to(dev)
    views = _as_views(frames)
    assert torch.equal(ops.wide_image(tuple(frames)), ops.stitch6(views)[0])
    assert torch.equal(ops.wide_image(frames), ops.stitch6(views)[0])
    assert torch.equal(ops.stitch6_u8(frames), ops.stitch6(views)[0])                      # the contiguous entry point: same division
    assert torch.equal(ops.wide_image(frames, "bf16").view(torch.int16), ops_bf16.stitch6_bf16(views).view(torch.int16))
    w_u8, y_u8 = ops.wide_image(frames, mask_slot=3, want_target=True)
    w_f, _, y_f = ops.stitch6(views, mask_slot=3, want_target=True)
    assert torch.equal(w_u8, w_f) and torch.equal(y_u8, y_f)
    for view in range(6):
        for tf in range(4):
            assert torch.equal(gconv.view_to_nhwc4(tuple(frames), view, tf), gconv.view_to_nhwc4(views, view, tf)), (view, tf)
    with pytest.raises(Exception):
        ops.wide_image(frames.cpu())                                                       # no CPU path


def test_device_prefetcher_delivers_batches_in_order(dev):
    """prefetch.DevicePrefetcher: host batches (the collate's nested tuples, pinned) arrive in HBM intact and in order while a
    long-running kernel keeps the compute stream busy; two device slots are reused."""
    from driving_dirty_amd.prefetch import DevicePrefetcher
    host = []
    for i in range(5):
        g = torch.Generator().manual_seed(40 + i)
        host.append((tuple(torch.randint(0, 256, (6, 16, 22, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(3)),
                     tuple({"id": i} for _ in range(3)),
                     tuple((torch.rand(800, 800, generator=g) < 0.3).pin_memory() for _ in range(3))))
    pf = DevicePrefetcher(host, dev)
    seen, ptrs = 0, set()
    busy = torch.zeros(1 << 24, device=dev)
    for i, (sample, target, road) in enumerate(pf):
        for _ in range(20):
            busy.add_(1.0)                                  # compute-stream work the next copy runs beside
        assert all(t.is_cuda for t in sample) and target[0]["id"] == i
        for a, b in zip(sample, host[i][0]):
            assert torch.equal(a.cpu(), b)
        for a, b in zip(road, host[i][2]):
            assert torch.equal(a.cpu(), b)
        ptrs.add(sample[0].data_ptr())
        seen += 1
    assert seen == 5 and len(ptrs) == 2


# ------------------------------------------------------------------------------------------------ config 5 at its own size
def test_full_size_bf16_roadmap_step_against_oracle(dev):
    """BASELINE config 5's step at its real shapes -- RoadMapBCE(precision='bf16'), 6 x 3 x 512 x 612 views (wide image
    512 x 3672), hidden 128 / latent 64, B = 4: the pool over 15 M features, fc1 at K = 3,760,128 and its 481 M-element weight
    gradient included -- against oracle/bf16_parts.py evaluated in fp64 ON THE DEVICE (conv stack in the mixed-precision
    contract: bf16-rounded operands, exact accumulation, one rounding per stored activation / activation gradient; FC tail, head
    and loss in fp64).  "Parity unpinned" by construction (the reference has no mixed precision); this holds the step to the
    stated contract at size.  Loss 1e-4; gradients within a per-tensor budget of bf16 rounding flips (each flip is 4e-3 of one
    activation; BatchNorm over 4 rows amplifies the few that matter)."""
    import json
    from torch.nn import functional as F
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    from oracle import ae_parts, bf16_parts, steps
    b, h, w = 4, 512, 612
    torch.manual_seed(20200505)
    ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64, input_height=h, input_width=6 * w, output_height=h, output_width=w))
    ae.decoder = None
    model = RoadMapBCE(Namespace(pretrained_ae=ae, precision="bf16", unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=10 ** 9))
    for blk in (model.ae.encoder.fc1, model.ae.encoder.fc2):
        blk.drop_p = 0.0
    enc = ae_parts.EncoderNet(128, 64, 3, h, 6 * w)
    enc.load_state_dict(model.ae.encoder.state_dict())
    enc = enc.double().to(dev)
    enc.fc1.drop_p = enc.fc2.drop_p = 0.0
    enc.train()
    head = torch.nn.Linear(64, 640000)
    head.load_state_dict(model.fc1.state_dict())
    head = head.double().to(dev)
    model = model.to(dev)
    g = torch.Generator().manual_seed(11)
    views = torch.rand(b, 6, 3, h, w, generator=g)
    road = torch.rand(b, 800, 800, generator=g) < 0.3
    out = model.training_step((tuple(views.to(dev)), None, tuple(road.to(dev))), 0)
    out["loss"].backward()
    torch.cuda.synchronize()

    wide = bf16_parts.bf16r(steps.wide_stitch(views)).to(dev)
    z = bf16_parts.encoder_latent(enc, wide)
    ref = F.binary_cross_entropy_with_logits(head(z).reshape(b, -1), road.to(dev).double().reshape(b, -1))
    ref.backward()
    loss_err = abs(float(out["loss"].detach()) - float(ref.detach())) / abs(float(ref.detach()))
    refs = dict([("ae.encoder." + k, p) for k, p in enc.named_parameters()] + [("fc1." + k, p) for k, p in head.named_parameters()])
    errs = {}
    for k, p in model.named_parameters():
        r = refs[k].grad
        floor = 1e-30
        if k.endswith("fc1.bias") and "encoder" in k:       # exactly zero in front of a train-mode BatchNorm: judged against the weight gradient
            floor = float(refs[k[:-4] + "weight"].grad.abs().max())
        errs[k] = float((p.grad.double() - r).abs().max() / max(float(r.abs().max()), floor))
    dump = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(dump):
        json.dump({"loss": float(out["loss"].detach()), "oracle_loss": float(ref.detach()), "loss_rel_err": loss_err, "grad_rel_err_of_peak": errs},
                  open(os.path.join(dump, "r4_bf16_fullsize_errors.json"), "w"), indent=1)
    assert loss_err < 1e-4, loss_err
    bad = {k: e for k, e in errs.items() if e > BF16_FULLSIZE_BUDGET.get(k, 1e-2)}
    assert not bad, bad


# per-tensor budgets of the full-size bf16 step (2x the errors measured on the MI355X, profiles/r04_bf16_fullsize_errors.json)
BF16_FULLSIZE_BUDGET = {
    "ae.encoder.c1.weight": 0.0089,
    "ae.encoder.c1.bias": 0.014,
    "ae.encoder.c2.weight": 0.008,
    "ae.encoder.c2.bias": 0.0069,
    "ae.encoder.c3.weight": 0.013,
    "ae.encoder.c3.bias": 0.007,
    "ae.encoder.fc1.fc1.weight": 0.018,
    "ae.encoder.fc1.fc1.bias": 0.001,
    "ae.encoder.fc1.fc_bn.weight": 0.0033,
    "ae.encoder.fc1.fc_bn.bias": 0.0024,
    "ae.encoder.fc2.fc1.weight": 0.006,
    "ae.encoder.fc2.fc1.bias": 0.001,
    "ae.encoder.fc2.fc_bn.weight": 0.0012,
    "ae.encoder.fc2.fc_bn.bias": 0.001,
    "ae.encoder.fc_z_out.weight": 0.0056,
    "ae.encoder.fc_z_out.bias": 0.001,
    "fc1.weight": 0.001,
    "fc1.bias": 0.001
}


# ------------------------------------------------------------------------------------------------ EXPERIMENT: bf16 x 3 split products
_SPLIT_CASES = [
    ("up1_96_64_small", 96, 64, (1, 9, 20)),
    ("up1_full_width", 96, 64, (2, 9, 256)),            # 8 m-tiles: one per wave
    ("up1_ragged_250", 96, 64, (1, 5, 250)),            # last m-tile 26 pixels: the DMA's range check supplies the zeros
    ("up2_full_width", 64, 32, (2, 9, 298)),            # 70 tiles over 8 waves
    ("up2_257_nine_tiles", 64, 32, (1, 3, 257)),
    ("up2_320_ten_full", 64, 64, (1, 2, 320)),          # widest row, two column tiles
    ("up2_tall", 64, 32, (2, 40, 100)),                 # every tap-row range of the residue classes
    ("up3_full_width", 32, 16, (1, 9, 340)),            # Cout 16: forward and weight gradient stay exact, the data gradient on the 11-wave form
    ("up3_352_widest", 32, 16, (1, 2, 352)),
]


@pytest.mark.parametrize("name,cin,cout,shape", _SPLIT_CASES, ids=[c[0] for c in _SPLIT_CASES])
def test_split_bf16_forward_and_data_gradient_against_fp64_and_the_exact_kernels(dev, name, cin, cout, shape):
    """csrc/dconv_split.hip (off by default): the k7 d7 transposed forward, its data gradient (with and without the fused ReLU mask
    of the producer) and its weight gradient with every fp32 product as six bf16 x bf16 products on the bf16 matrix pipe.  Held to the exact kernels' own
    bound -- 2e-5 of peak against fp64 torch -- and compared with the exact fp32 kernels on the same operands (the split path must
    not be the worse one by more than 4x)."""
    from torch import nn
    from torch.nn import functional as F
    from driving_dirty_amd import gconv, synth
    b, h, w = shape
    mod = synth.fill_module(nn.ConvTranspose2d(cin, cout, 7, dilation=7), seed=21).double()
    x = synth.hash_uniform((b, cin, h, w), synth.key_salt("spx" + name), -1.0, 1.0).double().requires_grad_(True)
    y_ref = F.relu(mod(x))
    gy = synth.hash_uniform(tuple(y_ref.shape), synth.key_salt("spg" + name), -1.0, 1.0).double()
    y_ref.backward(gy * (y_ref > 0))
    layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
    oh, ow = layer.out_hw(h, w)
    xb = x.detach().float().permute(0, 2, 3, 1).contiguous().to(dev)
    gb = (gy * (y_ref > 0)).detach().float().permute(0, 2, 3, 1).contiguous().to(dev)
    mb = (x.detach() - 0.25).float().permute(0, 2, 3, 1).contiguous().to(dev)          # a stand-in producer activation for the mask
    wd, bd = mod.weight.detach().float().to(dev), mod.bias.detach().float().to(dev)
    errs = {}
    for split in (False, True):
        yb = torch.full((b, oh, ow, cout), float("nan"), device=dev)
        dxb = torch.full((b, h, w, cin), float("nan"), device=dev)
        dxm = torch.full((b, h, w, cin), float("nan"), device=dev)
        old = gconv.SPLIT_BF16
        gconv.SPLIT_BF16 = split
        try:
            layer.forward(wd, bd, gconv.View(xb), gconv.View(yb), gconv.EPI_BIAS_RELU)
            layer.backward_data(wd, gconv.View(gb), gconv.View(dxb))
            layer.backward_data(wd, gconv.View(gb), gconv.View(dxm), relu_src=mb)
            assert layer.split_wgrad_ok(gconv.View(xb), gconv.View(gb)) == (split and (cin, cout) in ((96, 64), (64, 32)))
            dw, db = layer.backward_weight(gconv.View(xb), gconv.View(gb))
        finally:
            gconv.SPLIT_BF16 = old
        got = yb.permute(0, 3, 1, 2).double().cpu()
        gdx = dxb.permute(0, 3, 1, 2).double().cpu()
        gdm = dxm.permute(0, 3, 1, 2).double().cpu()
        assert torch.isfinite(got).all() and torch.isfinite(gdx).all() and torch.isfinite(gdm).all()
        gw_ref = mod.weight.grad
        errs[split] = (float((got - y_ref).abs().max() / y_ref.abs().max()), float((gdx - x.grad).abs().max() / x.grad.abs().max()),
                       float((gdm - x.grad * (x.detach() > 0.25)).abs().max() / x.grad.abs().max()),
                       float((dw.double().cpu() - gw_ref).abs().max() / gw_ref.abs().max()),
                       float((db.double().cpu() - mod.bias.grad).abs().max() / mod.bias.grad.abs().max()))
    for e_split, e_exact in zip(errs[True], errs[False]):
        assert e_split < 2e-5, errs
        assert e_split <= 4.0 * e_exact + 1e-7, errs


def test_split_bf16_data_gradient_mask_pass_and_channel_slice(dev):
    """The split data gradient writes a channel slice of a wider buffer and exempts `mask_pass` channels from the ReLU mask exactly as
    the exact kernel does (the 96-channel concat buffer of RoadMapBoxesMergingCNN: components.py:159)."""
    from driving_dirty_amd import gconv, synth
    b, h, w, cin, cout = 1, 9, 64, 96, 64
    layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
    oh, ow = layer.out_hw(h, w)
    wd = synth.hash_uniform((cin, cout, 7, 7), synth.key_salt("mpw"), -0.05, 0.05).to(dev)
    g = synth.hash_uniform((b, oh, ow, cout), synth.key_salt("mpg")).to(dev)
    msk = synth.hash_uniform((b, h, w, 128), synth.key_salt("mpm")).to(dev)
    outs = []
    for split in (False, True):
        dx = torch.zeros(b, h, w, 128, device=dev)
        old = gconv.SPLIT_BF16
        gconv.SPLIT_BF16 = split
        try:
            layer.backward_data(wd, gconv.View(g), gconv.View(dx, 16, cin), relu_src=msk, mask_pass=(48, 80))
        finally:
            gconv.SPLIT_BF16 = old
        outs.append(dx)
    a, c = outs
    assert float(a[..., :16].abs().max()) == 0 and float(c[..., :16].abs().max()) == 0 and float(c[..., 112:].abs().max()) == 0
    assert torch.equal(a == 0, c == 0)                       # the same elements masked
    assert float((a - c).abs().max() / a.abs().max()) < 2e-5


def test_split_bf16_pieces_reassemble_the_operand(dev):
    """dd_dconv_split_input: hi + mid + lo == x to 2^-26 |x| (three roundings to nearest, each of the exact remainder of the one
    before), every piece a bf16, and the pieces carry no sign bias (round to nearest: `mid` and `lo` are as often opposite in sign to x
    as not -- what keeps the dropped cross products zero-mean)."""
    import ctypes as C
    from driving_dirty_amd import _lib, gconv
    b, h, w, c = 1, 6, 40, 32                                 # 7680 values: enough pairs for the statistics of the dropped terms below
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(b, h, w, c, generator=g) * torch.logspace(-6, 3, c)).to(dev)
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.0])
    layer = gconv.Layer(c, 32, 7, dil=7, transposed=True)
    oh, ow = layer.out_hw(h, w)
    y = torch.empty(b, oh, ow, 32, device=dev)
    d = gconv._desc(b, gconv.View(x), gconv.View(y), c, 32, (7, 7), (1, 1), (7, 7), (42, 42))
    lib = _lib.lib()
    assert lib.dd_dconv_split_supported(C.byref(d)) == 1
    xs = torch.empty(lib.dd_dconv_split_input_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
    _lib.check(lib.dd_dconv_split_input(C.c_void_p(x.data_ptr()), C.c_void_p(xs.data_ptr()), C.byref(d), None), "split_input")
    torch.cuda.synchronize()
    assert torch.equal(xs, gconv.split_rows(gconv.View(x)))                                        # the descriptor-free entry point: same image
    img = xs.view(torch.int16).view(b, h, c // 16, w, 56)[..., :48].reshape(b, h, c // 16, w, 3, 16)      # [b, y, q, px, plane, ch]
    planes = (img.to(torch.int32) << 16).view(torch.float32)
    total = planes[..., 0, :].double() + planes[..., 1, :].double() + planes[..., 2, :].double()
    back = total.permute(0, 1, 3, 2, 4).reshape(b, h, w, c)
    err = (back - x.double()).abs()
    assert bool((err <= x.double().abs() * 2.0 ** -26).all())
    hi = planes[..., 0, :].permute(0, 1, 3, 2, 4).reshape(b, h, w, c)
    assert torch.equal(hi, x.to(torch.bfloat16).float())                                           # hi = the bf16 nearest x
    mid = planes[..., 1, :].permute(0, 1, 3, 2, 4).reshape(b, h, w, c)
    opposite = float(((mid * x) < 0).float().mean())
    assert 0.35 < opposite < 0.65, opposite
    # ---- the error model of the six-product form, asserted on these very pieces ---------------------------------
    # pieces: |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x| (each the round-to-nearest bf16 of an exact remainder)
    lo = planes[..., 2, :].permute(0, 1, 3, 2, 4).reshape(b, h, w, c)
    xa = x.double().abs()
    assert bool((mid.double().abs() <= xa * 2.0 ** -8).all()) and bool((lo.double().abs() <= xa * 2.0 ** -16).all())
    # a second operand: the same values in another order -- products a_k b_k over 3840 pairs of magnitudes 1e-6 .. 1e3
    a3 = torch.stack([hi.double().flatten(), mid.double().flatten(), lo.double().flatten()])            # [3, n]
    perm = torch.randperm(a3.shape[1], generator=torch.Generator().manual_seed(5)).to(dev)
    b3 = a3[:, perm]
    full = a3.sum(0) * b3.sum(0)
    H, M, L = 0, 1, 2
    kept = (a3[H] * b3[H] + a3[H] * b3[M] + a3[M] * b3[H] + a3[H] * b3[L] + a3[L] * b3[H] + a3[M] * b3[M])
    dropped = a3[M] * b3[L] + a3[L] * b3[M] + a3[L] * b3[L]
    assert float((full - kept - dropped).abs().max()) <= 1e-30 + 2.0 ** -50 * float(full.abs().max())      # nine terms, six issued, three dropped
    live = full.abs() > 0
    rel = (dropped[live] / full[live].abs())
    # WORST CASE per product: |a_m b_l| + |a_l b_m| + |a_l b_l| <= (2 x 2^-24 + 2^-32) |a b| = 2^-23 (1 + 2^-9) |a b|: fp32's own product rounding
    assert float(rel.abs().max()) <= 2.0 ** -23 * (1.0 + 2.0 ** -9)
    # EXPECTED: zero-mean (rounded pieces carry no sign bias), rms <= 2^-24 |a b| (uniform remainders: sqrt(2)/3 x 2^-24 = 2.8e-8)
    n = rel.numel()
    rms = float(rel.pow(2).mean().sqrt())
    assert rms <= 2.0 ** -24 and abs(float(rel.mean())) <= 5.0 * rms / n ** 0.5, (rms, float(rel.mean()))
    # a K-term dot product: the dropped terms add like a random walk -- |sum dropped| against the bound 2^-23 sum |a b| and the expectation
    K = 4704                                                                                          # up_conv_1's forward: 96 channels x 49 taps
    prod, drop = full[:K * (n // K)].view(-1, K), dropped[:K * (n // K)].view(-1, K)
    walk = drop.sum(1).abs() / prod.pow(2).sum(1).sqrt()
    assert bool((drop.sum(1).abs() <= 2.0 ** -23 * (1.0 + 2.0 ** -9) * prod.abs().sum(1)).all()) and float(walk.max()) <= 4.0 * 2.0 ** -24


def test_box_head_model_tests_pass_on_the_split_product_path(dev, golden):
    """The model-level tests of the box heads -- the three-way checks against the fp64 oracle (same-branch 2e-4, flip census, reference
    fixtures), the B = 32 step against the mean of 32 single-scene steps, adjointness of every layer at bs 32 -- re-run with the
    precision mode "fp32x3" as the process-wide default (``gconv.split_products(True)``: what ``hparams.precision = "fp32x3"`` sets
    per module): the forwards, data gradients and weight gradients of up_conv_1 / up_conv_2 on the split-product kernels.  Same test
    functions, same tolerances, no allowance for the mode.  (In this process: round 4 ran them in a child pytest, 8 s of imports.)"""
    import test_gpu_heads
    import test_gpu_round2
    import test_gpu_round3
    from driving_dirty_amd import _lib, gconv
    calls = []
    real = _lib.lib().dd_dconv_fwd_split

    def counted(*a):
        calls.append(1)
        return real(*a)
    with gconv.split_products(True):
        _lib.lib().dd_dconv_fwd_split = counted
        try:
            test_gpu_heads.test_spatial_heads_three_way(dev, golden)
            for mse in (False, True):
                test_gpu_heads.test_bbox_training_step_three_way(dev, mse)
            for variant in ("rboxm", "boxm"):
                test_gpu_round2.test_merging_heads_signed_inputs_three_way(dev, golden, variant)
            test_gpu_round3.test_bbox_step_b32_equals_mean_of_single_scene_steps(dev)
            for name in test_gpu_round3._BOX_LAYERS:
                test_gpu_round3.test_box_head_layers_full_size_linearity_and_adjointness(dev, name)
        finally:
            _lib.lib().dd_dconv_fwd_split = real
    assert len(calls) >= 20, "the split-product kernels were never reached"
    assert gconv.SPLIT_BF16 is False


def test_precision_mode_fp32x3_on_the_module_surface(dev):
    """``hparams.precision = "fp32x3"`` on BBSpatialRoadMap / JointRoadMapBBox: the step runs on the split-product kernels (forward AND
    backward, although the backward is started outside the module's ``with``), a module built without it stays on the exact kernels in
    the same process, and the two losses agree to the mode's bound."""
    from driving_dirty_amd import _lib, synth
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.joint import JointRoadMapBBox
    from driving_dirty_amd.spatial import BBSpatialRoadMap
    lib = _lib.lib()
    seen = {"fwd_split": 0, "wgrad_split": 0}
    real_f, real_w = lib.dd_dconv_fwd_split, lib.dd_dconv_wgrad_split

    def count(key, fn):
        def wrapped(*a):
            seen[key] += 1
            return fn(*a)
        return wrapped
    lib.dd_dconv_fwd_split, lib.dd_dconv_wgrad_split = count("fwd_split", real_f), count("wgrad_split", real_w)
    try:
        b = 2
        views = synth.camera_batch(b, seed=41).to(dev)
        road = synth.road_maps(b, seed=41).to(dev)
        tgt = tuple({"bb_map": (synth.hash_uniform((800, 800), 300 + i, 0.0, 1.0) < 0.02).float().to(dev)} for i in range(b))
        losses = {}
        for precision in ("fp32", "fp32x3"):
            ae = BasicAE(Namespace(hidden_dim=128, latent_dim=64))
            m = BBSpatialRoadMap(Namespace(pretrained_ae=ae, unfreeze_epoch_no=10 ** 9, learning_rate=1e-3, output_img_freq=500, precision=precision))
            synth.fill_module(m, seed=41)
            m = m.to(dev)
            assert m.box_merge.precision == precision and m.ae.encoder.precision == "fp32"
            before = dict(seen)
            out = m.training_step((tuple(views), tgt, tuple(road)), 0)
            mid = dict(seen)
            out["loss"].backward()
            if precision == "fp32":
                assert seen == before
            else:
                assert mid["fwd_split"] - before["fwd_split"] == 2                   # up_conv_1, up_conv_2 forward
                assert seen["fwd_split"] - mid["fwd_split"] >= 2 and seen["wgrad_split"] - mid["wgrad_split"] == 2      # data + weight gradients
            losses[precision] = float(out["loss"].detach())
            del m, ae
        assert abs(losses["fp32"] - losses["fp32x3"]) <= 2e-5 * abs(losses["fp32"])
        with pytest.raises(ValueError):
            BBSpatialRoadMap(Namespace(pretrained_ae=BasicAE(Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132)),
                                       unfreeze_epoch_no=0, learning_rate=1e-3, precision="fp16"))
        j = JointRoadMapBBox(Namespace(pretrained_ae=BasicAE(Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132)),
                                       learning_rate=1e-3, output_img_freq=500, precision="fp32x3"))
        assert j.box_merge.precision == "fp32x3"
    finally:
        lib.dd_dconv_fwd_split, lib.dd_dconv_wgrad_split = real_f, real_w


def test_split_kernels_emit_the_planes_of_their_output(dev):
    """The forward and the data gradient can write the three-plane bf16 image of their OUTPUT from the epilogue (the next layer's
    operand): bit for bit what a split pass over the fp32 output gives (the 16 pad bytes of a pixel record are never read and not
    compared)."""
    from driving_dirty_amd import gconv, synth
    for cin, cout, b, h, w in ((96, 64, 2, 9, 256), (64, 32, 1, 5, 298)):
        layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
        oh, ow = layer.out_hw(h, w)
        wd = synth.hash_uniform((cin, cout, 7, 7), synth.key_salt("epw"), -0.05, 0.05).to(dev)
        bd = synth.hash_uniform((cout,), synth.key_salt("epb")).to(dev)
        x = synth.hash_uniform((b, h, w, cin), synth.key_salt("epx")).to(dev)
        g = synth.hash_uniform((b, oh, ow, cout), synth.key_salt("epg")).to(dev)
        old = gconv.SPLIT_BF16
        gconv.SPLIT_BF16 = True
        try:
            y = torch.empty(b, oh, ow, cout, device=dev)
            ef, eb = {}, {}
            layer.forward(wd, bd, gconv.View(x), gconv.View(y), gconv.EPI_BIAS_RELU, emit=ef)
            dx = torch.empty(b, h, w, cin, device=dev)
            layer.backward_data(wd, gconv.View(g), gconv.View(dx), relu_src=x, emit=eb)
            for planes, t in ((ef["ys"], y), (eb["ys"], dx)):
                want = gconv.split_rows(gconv.View(t)).view(-1, 112)[:, :96]
                assert torch.equal(planes.view(-1, 112)[:, :96], want)
        finally:
            gconv.SPLIT_BF16 = old


def test_split_product_kernels_are_deterministic(dev):
    """Two launches on the same operands agree bit for bit: the forward adds its partial rows in barrier-separated passes, the data
    gradient has no cross-wave sums, the weight gradient sums its per-workgroup partials in a fixed-order fp64 second stage."""
    from driving_dirty_amd import gconv, synth
    cin, cout, b, h, w = 96, 64, 2, 16, 256
    layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
    oh, ow = layer.out_hw(h, w)
    wd = synth.hash_uniform((cin, cout, 7, 7), synth.key_salt("dtw"), -0.05, 0.05).to(dev)
    bd = synth.hash_uniform((cout,), synth.key_salt("dtb")).to(dev)
    x = synth.hash_uniform((b, h, w, cin), synth.key_salt("dtx")).to(dev)
    g = synth.hash_uniform((b, oh, ow, cout), synth.key_salt("dtg")).to(dev)
    old = gconv.SPLIT_BF16
    gconv.SPLIT_BF16 = True
    try:
        outs = []
        for _ in range(2):
            y = torch.empty(b, oh, ow, cout, device=dev)
            layer.forward(wd, bd, gconv.View(x), gconv.View(y), gconv.EPI_BIAS_RELU)
            dx = torch.empty(b, h, w, cin, device=dev)
            layer.backward_data(wd, gconv.View(g), gconv.View(dx), relu_src=x)
            dw, db = layer.backward_weight(gconv.View(x), gconv.View(g))
            outs.append((y, dx, dw.clone(), db.clone()))
    finally:
        gconv.SPLIT_BF16 = old
    for a, c in zip(*outs):
        assert torch.equal(a, c)


# ---- round 4, late: the multi-row gather kernel (csrc/dconv_m.hip), ss_conv's one-launch data gradient (csrc/ssconv.hip)

@pytest.mark.parametrize("cin,cout,b,h,w", [(64, 32, 2, 13, 298), (64, 32, 1, 50, 77), (32, 16, 2, 9, 340), (32, 16, 3, 47, 45)])
def test_multi_row_data_gradient_against_fp64_and_the_one_row_kernels(dev, cin, cout, b, h, w):
    """up_conv_2 / up_conv_3 data gradients on the multi-row kernel (tiles that straddle rows, tasks cut at image ends and at the ends of a
    workgroup's row range, a channel slice with an exempt mask range) against torch fp64 at 2e-5 of the largest value; unwritten
    elements of the output buffer keep their canary."""
    import torch.nn.functional as F
    from driving_dirty_amd import gconv, synth
    layer = gconv.Layer(cin, cout, 7, dil=7, transposed=True)
    oh, ow = layer.out_hw(h, w)
    wd = synth.hash_uniform((cin, cout, 7, 7), synth.key_salt("mrw"), -0.05, 0.05).to(dev)
    g = synth.hash_uniform((b, oh, ow, cout), synth.key_salt("mrg"), -1.0, 1.0).to(dev)
    x = synth.hash_uniform((b, h, w, cin), synth.key_salt("mrx"), -1.0, 1.0).to(dev)
    ref = F.conv2d(g.permute(0, 3, 1, 2).double(), wd.double(), dilation=7).permute(0, 2, 3, 1)
    for masked in (False, True):
        buf = torch.full((b, h, w, cin + 8), 7.0, device=dev)                      # the layer's slice sits at channel offset 4
        src = torch.zeros_like(buf)
        src[..., 4:4 + cin] = x
        lo, hi = 4 + cin // 2, 4 + cin                                             # the upper half of the slice is exempt from the mask
        csum = torch.full((cin,), float("nan"), device=dev)
        took = layer.backward_data(wd, gconv.View(g), gconv.View(buf, 4, cin), relu_src=src if masked else None,
                                   mask_pass=(lo, hi) if masked else (0, 0), colsum=csum)
        assert took                                                                # these layers run on the windowed kernel, which sums its output
        sums = buf[..., 4:4 + cin].double().sum(dim=(0, 1, 2))
        assert (csum.double() - sums).abs().max().item() <= 1e-5 * max(sums.abs().max().item(), 1.0), "per-channel sums of the written gradient"
        want = ref.clone()
        if masked:
            keep = (x > 0).double()
            keep[..., cin // 2:] = 1.0
            want = want * keep
        err = (buf[..., 4:4 + cin].double() - want).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (masked, err)
        assert torch.all(buf[..., :4] == 7.0) and torch.all(buf[..., 4 + cin:] == 7.0)


@pytest.mark.parametrize("b,h,xw", [(2, 5, 918), (3, 7, 311), (1, 128, 24), (2, 3, 919)])
def test_ss_conv_data_gradient_in_one_launch(dev, b, h, xw):
    """csrc/ssconv.hip against torch fp64 (the transposed convolution autograd runs for F.conv2d's input, components.py:129) and against
    the seven phase launches of the generic engine: full width, a narrow image, a single tap position, trailing pixels no tap reaches."""
    import torch.nn.functional as F
    from driving_dirty_amd import gconv, synth
    layer = gconv.Layer(32, 32, (1, 24), stride=(1, 7))
    gw = (xw - 24) // 7 + 1
    wt = synth.hash_uniform((32, 32, 1, 24), synth.key_salt("ssw"), -0.1, 0.1).to(dev)
    g = synth.hash_uniform((b, h, gw, 32), synth.key_salt("ssg"), -1.0, 1.0).to(dev)
    ref = F.conv_transpose2d(g.permute(0, 3, 1, 2).double(), wt.double(), stride=(1, 7))
    ref = F.pad(ref, (0, xw - ref.shape[3])).permute(0, 2, 3, 1)
    outs = []
    old = gconv.SSCONV_DGRAD
    try:
        for on in (True, False):
            gconv.SSCONV_DGRAD = on
            dx = torch.full((b, h, xw, 32), float("nan"), device=dev)
            layer.backward_data(wt, gconv.View(g), gconv.View(dx))
            outs.append(dx)
    finally:
        gconv.SSCONV_DGRAD = old
    scale = ref.abs().max().item()
    for dx in outs:
        assert (dx.double() - ref).abs().max().item() / scale < 2e-6
    assert (outs[0] - outs[1]).abs().max().item() / scale < 2e-6


@pytest.mark.parametrize("b,h,w,gcs,coff", [(2, 5, 7, 32, 0), (3, 16, 16, 96, 0), (1, 9, 3, 40, 8)])
def test_k2s2_weight_gradient_in_one_launch(dev, b, h, w, gcs, coff):
    """ConvTranspose2d(32, 32, k2, s2)'s weight / bias gradient (ss_deconv, the decoder's dc3) by `dd_deconv2x2_c32_wgrad` against torch
    fp64 autograd and against the four phase launches of the generic engine; an odd pixel count, a channel slice of a wider buffer."""
    import torch.nn.functional as F
    from driving_dirty_amd import gconv, synth
    layer = gconv.Layer(32, 32, 2, stride=2, transposed=True)
    x = synth.hash_uniform((b, h, w, 32), synth.key_salt("k2x"), -1.0, 1.0).to(dev)
    gbuf = synth.hash_uniform((b, 2 * h, 2 * w, gcs), synth.key_salt("k2g"), -1.0, 1.0).to(dev)
    wt = torch.zeros(32, 32, 2, 2, dtype=torch.float64, device=dev, requires_grad=True)
    bias = torch.zeros(32, dtype=torch.float64, device=dev, requires_grad=True)
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt, bias, stride=2)
    (y * gbuf[..., coff:coff + 32].permute(0, 3, 1, 2).double()).sum().backward()
    outs = []
    old = gconv.K2S2_WGRAD
    try:
        for on in (True, False):
            gconv.K2S2_WGRAD = on
            outs.append(layer.backward_weight(gconv.View(x), gconv.View(gbuf, coff, 32)))
    finally:
        gconv.K2S2_WGRAD = old
    for dw, db in outs:
        assert (dw.double() - wt.grad).abs().max().item() <= 2e-6 * wt.grad.abs().max().item()
        assert (db.double() - bias.grad).abs().max().item() <= 2e-6 * bias.grad.abs().max().item()


@pytest.mark.parametrize("b,h,xw", [(2, 5, 918), (3, 7, 311), (1, 4, 24), (2, 3, 919)])
def test_ss_conv_forward_in_one_launch(dev, b, h, xw):
    """csrc/ssconv.hip's forward against torch fp64 and the generic engine: full width, a narrow image (a ragged last m-tile), a single
    output pixel, trailing input pixels no tap reaches."""
    import torch.nn.functional as F
    from driving_dirty_amd import gconv, synth
    layer = gconv.Layer(32, 32, (1, 24), stride=(1, 7))
    gw = (xw - 24) // 7 + 1
    wt = synth.hash_uniform((32, 32, 1, 24), synth.key_salt("sfw"), -0.1, 0.1).to(dev)
    bias = synth.hash_uniform((32,), synth.key_salt("sfb"), -0.5, 0.5).to(dev)
    x = synth.hash_uniform((b, h, xw, 32), synth.key_salt("sfx"), -1.0, 1.0).to(dev)
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=(1, 7))).permute(0, 2, 3, 1)
    outs = []
    old = gconv.SSCONV_FWD
    try:
        for on in (True, False):
            gconv.SSCONV_FWD = on
            y = torch.full((b, h, gw, 32), float("nan"), device=dev)
            layer.forward(wt, bias, gconv.View(x), gconv.View(y), gconv.EPI_BIAS_RELU)
            outs.append(y)
    finally:
        gconv.SSCONV_FWD = old
    scale = ref.abs().max().item()
    for y in outs:
        assert (y.double() - ref).abs().max().item() / scale < 2e-6
