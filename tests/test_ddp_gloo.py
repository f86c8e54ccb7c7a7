"""world_size-2 gloo test of the data-parallel gradient path (runs on CPU; the N>1 GPU path is the same code
over RCCL).  Checks that GradSync + a 1/world scale reproduces single-process full-batch gradients."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port  # noqa: E402


def _net():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 2048), torch.nn.ReLU(), torch.nn.Linear(2048, 3))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from driving_dirty_amd.ddp import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _net()
    sync = GradSync(net, big_numel=4096, chunk_numel=10000)   # Linear(12,2048).weight goes the "big tensor" way, in 3 pieces
    torch.manual_seed(100)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    xs, ys = x[rank::world], y[rank::world]
    for _ in range(2):                             # two steps: hooks must re-arm
        net.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(net(xs), ys, reduction="sum") / x.shape[0] * world
        loss.backward()
        pieces = sync.pieces(net[0].weight)
        assert [(o, n) for _, o, n in pieces] == [(0, 10000), (10000, 10000), (20000, 4576)] and sync.pieces(net[0].bias) is None
        sync.finish()
    grads = [p.grad * sync.grad_scale for p in net.parameters()]
    if rank == 0:
        torch.save(grads, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_matches_full_batch(tmp_path):
    out = str(tmp_path / "g.pt")
    port = free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    net = _net()
    torch.manual_seed(100)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    (torch.nn.functional.mse_loss(net(x), y, reduction="sum") / x.shape[0]).backward()
    for g, p in zip(got, net.parameters()):
        torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-6)


def test_gradsync_single_process_is_noop():
    sys.path.insert(0, ROOT)
    from driving_dirty_amd.ddp import GradSync
    net = _net()
    sync = GradSync(net)
    assert sync.world == 1 and sync.grad_scale == 1.0
    net(torch.randn(2, 12)).sum().backward()
    sync.finish()


def test_gradsync_broadcasts_rank0_weights(tmp_path):
    """Ranks seeded differently must leave GradSync's constructor with rank 0's parameters AND buffers."""
    out = str(tmp_path / "b.pt")
    port = free_port()
    mp.spawn(_bcast_worker, args=(2, port, out), nprocs=2, join=True)
    sd0, sd1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert sorted(sd0) == sorted(sd1)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def _bcast_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from driving_dirty_amd.ddp import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)                       # different weights on every rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    net[1].running_mean.fill_(float(rank) + 1.0)
    GradSync(net)
    torch.save(net.state_dict(), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


# ---- frozen / later-unfrozen parameters (the reference's freeze -> unfreeze flow: roadmap_bce_v2.py:45-47,127-129) ----------------
class _FineTune(torch.nn.Module):
    """A frozen feature extractor under a trainable head, as RoadMapBCE / BBSpatialRoadMap build them."""

    def __init__(self):
        super().__init__()
        from driving_dirty_amd.lightning import LightningModule

        class _AE(LightningModule):
            def __init__(self):
                super().__init__()
                self.body = torch.nn.Sequential(torch.nn.Linear(12, 512), torch.nn.Tanh(), torch.nn.Linear(512, 16), torch.nn.Tanh())

            def forward(self, x):
                return self.body(x)
        torch.manual_seed(11)
        self.ae = _AE()
        self.ae.freeze()
        self.head = torch.nn.Linear(16, 3)

    def forward(self, x):
        return self.head(self.ae(x))


def _ft_batch(step, rank):
    g = torch.Generator().manual_seed(500 + 10 * step + rank)
    return torch.randn(4, 12, generator=g), torch.randn(4, 3, generator=g)


def _ft_loss(net, step, rank):
    x, y = _ft_batch(step, rank)
    return torch.nn.functional.mse_loss(net(x), y)


def _unfreeze(net, how):
    if how == "lightning":
        net.ae.unfreeze()                      # LightningModule.unfreeze(): notifies the live GradSync (lightning.on_unfreeze)
    else:
        for p in net.ae.parameters():          # by hand: no notification, GradSync.finish() has to notice
            p.requires_grad_(True)
        net.ae.train()


def _frozen_worker(rank, world, port, out, how):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from driving_dirty_amd.ddp import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _FineTune()
    sync = GradSync(net, big_numel=4096, chunk_numel=5000)      # constructing it over frozen parameters must not raise
    assert len(sync._hooks) == 2                                  # only the head is hooked
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for step in range(4):
        if step == 1:
            _unfreeze(net, how)
        net.zero_grad(set_to_none=True)
        _ft_loss(net, step, rank).backward()
        if step == 0:
            assert all(p.grad is None for p in net.ae.parameters())
        if step >= 1 and how == "lightning":
            assert sync.pieces(net.ae.body[0].weight) is not None          # already on the asynchronous big-tensor path
        if step >= 2:
            assert sync.pieces(net.ae.body[0].weight) is not None          # hooked by finish() of step 1 at the latest
        sync.finish()
        for p in net.parameters():
            if p.grad is not None:
                p.grad.mul_(sync.grad_scale)
        opt.step()
    torch.save(net.state_dict(), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("how", ["lightning", "by_hand"])
def test_gradsync_on_a_frozen_model_and_unfreeze_later(tmp_path, how):
    """GradSync over a model whose feature extractor is frozen (torch refuses gradient hooks there), unfrozen after step 0 --
    through LightningModule.unfreeze() or by plain requires_grad_(True): the replicas stay bit-identical through step 3 and
    equal a single process fed the summed gradients."""
    sys.path.insert(0, ROOT)
    out = str(tmp_path / "f.pt")
    port = free_port()
    mp.spawn(_frozen_worker, args=(2, port, out, how), nprocs=2, join=True)
    sd0, sd1 = torch.load(out + ".0"), torch.load(out + ".1")
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                   # as on the ranks: one summation order
    try:
        net = _FineTune()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        for step in range(4):
            if step == 1:
                _unfreeze(net, "by_hand")
            sums = None
            for rank in range(2):
                net.zero_grad(set_to_none=True)
                _ft_loss(net, step, rank).backward()
                g = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
                sums = g if sums is None else [None if a is None else a + b for a, b in zip(sums, g)]
            for p, g in zip(net.parameters(), sums):
                p.grad = None if g is None else g * 0.5
            opt.step()
    finally:
        torch.set_num_threads(threads)
    ref = net.state_dict()
    for k in ref:
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged: {k}"
        assert torch.equal(sd0[k], ref[k]), f"differs from the single process fed the summed gradients: {k}"
    moved = (ref["ae.body.0.weight"] - _FineTune().state_dict()["ae.body.0.weight"]).abs().max()
    assert float(moved) > 0                                                  # the unfrozen extractor did train


def test_bench_parent_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher: the parent starts N children with RANK / WORLD_SIZE / MASTER_* set and
    returns their status without importing torch or touching a GPU itself (here the children stop at the GPU check)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""          # no GPU for the children either way
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0                                               # the ranks refuse to run without a GPU ...
    assert "needs an MI355X" in r.stderr and "WORLD_SIZE" not in r.stderr  # ... but they WERE started as ranks of a 2-process job


# ---- on the GPU box: the product optimizer under data parallelism, two ranks on one card over gloo -------------------------------
def _tiny_model(dev):
    from argparse import Namespace
    from driving_dirty_amd import synth
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    ae = BasicAE(Namespace(hidden_dim=16, latent_dim=8, input_height=16, input_width=132))
    model = RoadMapBCE(Namespace(pretrained_ae=ae, unfreeze_epoch_no=0, learning_rate=1e-2, output_img_freq=500))
    synth.fill_module(model, seed=77)
    model = model.to(dev)
    model.ae.encoder.fc1.drop_p = model.ae.encoder.fc2.drop_p = 0.0
    model.frozen = False
    model.ae.unfreeze()
    return model


def _tiny_batch(dev, step, rank):
    from driving_dirty_amd import synth
    views = synth.camera_batch(3, 16, 22, seed=100 + 10 * step + rank).to(dev)
    road = synth.road_maps(3, seed=100 + 10 * step + rank).to(dev)
    return (tuple(views), None, tuple(road))


def _adam_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from driving_dirty_amd.ddp import GradSync
    from driving_dirty_amd.optim import HipAdam
    dev = torch.device("cuda:0")
    model = _tiny_model(dev)
    opt = HipAdam(model.parameters(), lr=1e-2)
    sync = GradSync(model, big_numel=1000, chunk_numel=1 << 18)      # the 5.12 M-element head weight: 20 pieces (4096-element pieces were 1250
    opt.overlap_with_backward(big_numel=1000, grad_scale=sync.grad_scale, grad_sync=sync)      # gloo round trips per step: 12-24 s of this test)
    for step in range(3):
        model.zero_grad(set_to_none=True)
        model.training_step(_tiny_batch(dev, step, rank), step)["loss"].backward()
        assert sync.pieces(model.fc1.weight) is not None and len(sync.pieces(model.fc1.weight)) > 1      # travels in pieces
        if step == 0:      # this rank's own gradients of the small tensors (they are reduced in finish())
            torch.cuda.synchronize()
            torch.save({k: p.grad.cpu().clone() for k, p in model.named_parameters() if p.numel() < 1000}, f"{out}.local{rank}")
        sync.finish()
        opt.step(grad_scale=sync.grad_scale)
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_hipadam_overlapped_under_data_parallel_matches_single_process(tmp_path):
    """HipAdam.overlap_with_backward(grad_sync=GradSync) -- per-piece waits on the side stream, 1/world folded into the Adam
    pass -- on two ranks (gloo, both on cuda:0) leaves the parameters a single process gets from the summed gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    from driving_dirty_amd.optim import HipAdam
    out = str(tmp_path / "a.pt")
    port = free_port()
    mp.spawn(_adam_worker, args=(2, port, out), nprocs=2, join=True)
    got0, got1 = torch.load(out + ".0"), torch.load(out + ".1")
    dev = torch.device("cuda:0")
    model = _tiny_model(dev)
    opt = HipAdam(model.parameters(), lr=1e-2)
    opt.SMALL_NUMEL = 999            # the same tensors through the same kernels as on the ranks (>= 1000 elements: one launch each)
    params = dict(model.named_parameters())
    local = [torch.load(f"{out}.local{r}") for r in range(2)]
    for step in range(3):
        sums = None
        for rank in range(2):
            model.zero_grad(set_to_none=True)
            model.training_step(_tiny_batch(dev, step, rank), step)["loss"].backward()
            g = {k: p.grad.clone() for k, p in params.items()}
            if step == 0:
                ld = {k: float((v - g[k].cpu()).abs().max() / g[k].abs().max().clamp_min(1e-30)) for k, v in local[rank].items()}
                assert all(d == 0.0 for d in ld.values()), f"rank {rank}: local gradients differ from the same backward run alone: {ld}"
            sums = g if sums is None else {k: sums[k] + g[k] for k in g}
        for k, p in params.items():
            p.grad = sums[k]
        opt.step(grad_scale=0.5)
    torch.cuda.synchronize()
    diffs = {}
    for k, p in params.items():
        assert torch.equal(got0[k], got1[k]), k                           # the replicas stay identical, bit for bit
        ref = p.detach().cpu()
        diffs[k] = float((got0[k] - ref).abs().max() / ref.abs().max())
    # ... and equal the single-process result (the sum of two gradients is the same number whoever adds them; what may differ
    # in the last bit is the kernels' own run-to-run summation order inside one backward)
    bad = {k: d for k, d in diffs.items() if d > 1e-6}
    assert not bad, bad


@pytest.mark.gpu
def test_bench_step_over_a_one_rank_rccl_communicator():
    """The N > 1 call pattern of the bench step -- RCCL initialised on the device, rank 0's weights broadcast, each gradient
    piece all-reduced asynchronously from its autograd hook, HipAdam waiting piece by piece on its side stream, compute units
    handed to RCCL for the backward, the barrier and the MAX-over-ranks timing -- on the real backend with the one GPU of
    this box (a 1-rank all-reduce is the identity, so the step must also still produce a finite loss)."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DD_REHEARSE_RCCL="1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-others", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_ranks_seen"] == 1 and "rehearsal" in line and line["value"] > 0
    loss = line["config"]["final_loss"]
    assert loss == loss and abs(loss) < 1e3                                  # finite


# ---- bench.py --gpus N: the failure paths of its first real multi-rank run, rehearsed on the control flow alone (no GPU here) ------
def _bench_control(tmp_env, *args, timeout=120):
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(DD_BENCH_CONTROL_ONLY="1", **tmp_env)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.monotonic() - t0


def test_bench_control_flow_rehearsal_two_ranks():
    """Bounded rendezvous, preflight count, barriers, a collective per step, MAX-over-ranks timing: two gloo ranks, no kernels."""
    import json
    r, _ = _bench_control({}, "--gpus", "2", "--steps", "4", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec == {"control_flow_rehearsal": True, "n_ranks_seen": 2, "steps": 4, "warmup": 1}
    pre = [ln for ln in r.stderr.splitlines() if "bench.py preflight:" in ln]
    assert len(pre) == 1 and pre[0].startswith("[rank 0] ") and '"n_ranks_seen": 2' in pre[0]      # rank-tagged stderr, one preflight line


def test_bench_rank_killed_mid_step_fails_fast_and_names_the_rank():
    """Rank 1 dies inside step 2: `bench.py --gpus 2` returns non-zero within 30 s and says which rank failed."""
    r, dt = _bench_control({"DD_BENCH_FAULT": "1:2"}, "--gpus", "2", "--steps", "6", "--warmup", "1")
    assert r.returncode != 0 and dt < 30.0, (r.returncode, dt, r.stderr[-1500:])
    assert "rank 1 exited with status 17" in r.stderr and "[rank 1] bench.py: injected fault on rank 1 at step 2" in r.stderr
    assert "control_flow_rehearsal" not in r.stdout                       # no record from a failed job


def test_bench_rank_hung_mid_step_is_ended_by_the_watchdog():
    """Rank 1 stops making progress inside step 2 (a collective whose partner never arrives looks the same): the per-rank watchdog
    prints the phase and a traceback and EXITS (status 3), the launcher tears the job down: non-zero within 30 s."""
    r, dt = _bench_control({"DD_BENCH_FAULT": "1:2:hang", "DD_WATCHDOG_S": "4", "DD_DIST_TIMEOUT_S": "60"}, "--gpus", "2", "--steps", "6", "--warmup", "1")
    assert r.returncode != 0 and dt < 30.0, (r.returncode, dt, r.stderr[-1500:])
    assert "bench.py watchdog: rank" in r.stderr and "made no progress" in r.stderr and "in phase 'step 2'" in r.stderr
    assert "exited with status 3" in r.stderr


def test_bench_rendezvous_with_a_missing_rank_times_out():
    """WORLD_SIZE says 2 but only one process shows up: the bounded rendezvous (or the watchdog) ends it -- no silent wait for the
    driver's limit."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DD_BENCH_CONTROL_ONLY="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(free_port()), DD_DIST_TIMEOUT_S="5", DD_WATCHDOG_S="8")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.monotonic() - t0 < 40.0, (r.returncode, r.stderr[-1500:])
