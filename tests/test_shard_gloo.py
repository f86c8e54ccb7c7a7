"""Sharded optimizer (ddp.GradSync(shard_optimizer=True) + optim.HipAdam shard mode) on gloo, world sizes 2 and 4, on CPU.

The bookkeeping under test is the product's own -- GradSync's reduce-scatter pieces / in-place all-gathers / waits and HipAdam's
per-shard state, step counts and re-arming after unfreeze(); only the elementwise kernel (``HipAdam._launch``, a HIP launch in the
product) is replaced by the same arithmetic in torch ops, because this container has no GPU.  The GPU twin (the real kernel, two
ranks on one card) is tests/test_gpu_round4.py.

Claim: through four steps of a model whose feature extractor is frozen, then unfrozen after step 0 (roadmap_bce_v2.py:45-47,
127-129), every replica of the SHARDED run holds bit for bit the parameters of the ALL-REDUCE run.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port  # noqa: E402
sys.path.insert(0, ROOT)


def _torch_adam_cls():
    from driving_dirty_amd.optim import HipAdam

    class TorchAdam(HipAdam):
        """HipAdam with its elementwise kernel written in torch ops (each one IEEE-rounded by itself, so the result of an element
        does not depend on the length or alignment of the tensor it sits in)."""
        SMALL_NUMEL = 0                                    # no multi-tensor launch: every tensor through _launch

        def _launch(self, p, g, m, v, group, step, grad_scale):
            b1, b2 = group["betas"]
            g = g * grad_scale
            m.mul_(b1).add_(g * (1.0 - b1))
            v.mul_(b2).add_((g * g) * (1.0 - b2))
            denom = (v / (1.0 - b2 ** step)).sqrt() + group["eps"]
            p.sub_((m / denom) * (group["lr"] / (1.0 - b1 ** step)))
    return TorchAdam


class _FineTune(torch.nn.Module):
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
        self.head = torch.nn.Linear(16, 4100)               # 65,600 weights: a big tensor that is trainable from step 0
        self.out = torch.nn.Linear(4100, 3)                 # 12,300 weights: big, but 12300 % (4 * world) != 0 at world 4 -> all-reduce

    def forward(self, x):
        return self.out(torch.tanh(self.head(self.ae(x))))


def _loss(net, step, rank):
    g = torch.Generator().manual_seed(900 + 10 * step + rank)
    x, y = torch.randn(4, 12, generator=g), torch.randn(4, 3, generator=g)
    return torch.nn.functional.mse_loss(net(x), y)


def _worker(rank, world, port, out, shard, how):
    torch.set_num_threads(1)
    from driving_dirty_amd import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _FineTune()
    sync = ddp.GradSync(net, big_numel=4096, chunk_numel=4992, shard_optimizer=shard)      # 4992 = 16 * 312: the same piece boundaries in both modes
    opt = _torch_adam_cls()(net.parameters(), lr=1e-2)
    opt.attach(sync)
    seen = {}
    for step in range(4):
        if step == 1:
            if how == "lightning":
                net.ae.unfreeze()
            else:
                for p in net.ae.parameters():
                    p.requires_grad_(True)
                net.ae.train()
        net.zero_grad(set_to_none=True)
        _loss(net, step, rank).backward()
        sync.finish()
        if shard:
            sh = sync.shards(net.head.weight)
            assert sh is not None and len(sh) == 14 and sync.pieces(net.head.weight) is None      # 65,600 / 4,992 -> 14 pieces
            assert all((s.hi - s.lo) * world == s.piece_hi - s.piece_lo and s.lo == s.piece_lo + rank * (s.hi - s.lo) for s in sh)
            seen["out_sharded"] = sync.shards(net.out.weight) is not None
            if step >= 2:
                assert sync.shards(net.ae.body[0].weight) is not None
        opt.step(grad_scale=sync.grad_scale)
    sync.wait_gathers()
    assert not ddp.PARAM_WAITS
    if shard:
        st = opt.state[net.head.weight]
        assert "exp_avg" not in st and sum(m.numel() for m, _ in st["shards"].values()) * world == net.head.weight.numel()
        assert seen["out_sharded"] == (12300 % (4 * world) == 0)
    torch.save(net.state_dict(), f"{out}.{int(shard)}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,how", [(2, "lightning"), (2, "by_hand"), (4, "lightning"), (8, "lightning")])
def test_sharded_optimizer_is_bit_identical_to_the_all_reduce_path(tmp_path, world, how):
    out = str(tmp_path / "s.pt")
    for shard in (False, True):
        mp.spawn(_worker, args=(world, free_port(), out, shard, how), nprocs=world, join=True)
    ref = torch.load(f"{out}.0.0")
    moved = (ref["ae.body.0.weight"] - _FineTune().state_dict()["ae.body.0.weight"]).abs().max()
    assert float(moved) > 0                                      # the unfrozen extractor did train
    for shard in (0, 1):
        for rank in range(world):
            sd = torch.load(f"{out}.{shard}.{rank}")
            for k in ref:
                assert torch.equal(sd[k], ref[k]), f"shard={shard} rank={rank}: {k} differs from the all-reduce path on rank 0"


def test_simulated_shard_world_updates_only_rank0s_slices():
    """simulate_world=N (one process): the optimizer touches exactly the slices rank 0 of an N-rank job would own."""
    from driving_dirty_amd import ddp
    torch.manual_seed(3)
    net = torch.nn.Linear(64, 256)                           # 16,384 weights
    before = net.weight.detach().clone()
    sync = ddp.GradSync(net, big_numel=4096, chunk_numel=6000, shard_optimizer=True, simulate_world=8)
    assert sync.shard and not sync.active and sync.chunk_numel == 5984
    opt = _torch_adam_cls()(net.parameters(), lr=1e-2)
    opt.attach(sync)
    net(torch.randn(5, 64)).square().mean().backward()
    sync.finish()
    opt.step()
    changed = (net.weight.detach() != before).view(-1)
    want = torch.zeros_like(changed)
    for a in range(0, 16384, 5984):
        b = min(a + 5984, 16384)
        want[a:a + (b - a) // 8] = True
    assert torch.equal(changed, want)
    assert not ddp.PARAM_WAITS


def _ckpt_worker(rank, world, port, out):
    """Two steps sharded -> consolidated_state_dict() -> (a) a fresh SHARDED optimizer loads it and takes two more steps, (b) a fresh
    ALL-REDUCE optimizer loads it and takes the same two steps, (c) the uninterrupted sharded run takes them: all three must agree bit for
    bit, and the consolidated moments must equal the all-reduce run's whole moments."""
    torch.set_num_threads(1)
    from driving_dirty_amd import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Adam = _torch_adam_cls()

    def make(shard):
        net = _FineTune()
        for p in net.ae.parameters():
            p.requires_grad_(True)
        net.ae.train()
        sync = ddp.GradSync(net, big_numel=4096, chunk_numel=4992, shard_optimizer=shard)
        opt = Adam(net.parameters(), lr=1e-2)
        opt.attach(sync)
        return net, sync, opt

    def steps(net, sync, opt, which):
        for step in which:
            net.zero_grad(set_to_none=True)
            _loss(net, step, rank).backward()
            sync.finish()
            opt.step(grad_scale=sync.grad_scale)
        sync.wait_gathers()

    ref_net, ref_sync, ref_opt = make(False)                 # all-reduce throughout
    steps(ref_net, ref_sync, ref_opt, range(4))
    net, sync, opt = make(True)                              # sharded throughout
    steps(net, sync, opt, range(2))
    import copy
    sd = copy.deepcopy(opt.consolidated_state_dict())        # collective: every rank.  (Like torch's state_dict() it REFERENCES the live
                                                             # moments of the tensors that were never sharded: copied before the run goes on)
    weights = {k: v.clone() for k, v in net.state_dict().items()}      # through the state_dict pre-hook (waits for the gathers)
    half_net, half_sync, half_opt = make(False)
    steps(half_net, half_sync, half_opt, range(2))
    want = half_opt.state_dict()
    for idx, st in want["state"].items():
        got = sd["state"][idx]
        assert "shards" not in got and got["step"] == st["step"] == 2
        assert torch.equal(got["exp_avg"], st["exp_avg"]) and torch.equal(got["exp_avg_sq"], st["exp_avg_sq"]), idx
    steps(net, sync, opt, range(2, 4))
    resumed = {}
    for name, shard in (("sharded", True), ("all_reduce", False)):
        n2, s2, o2 = make(shard)
        n2.load_state_dict(weights)
        o2.load_state_dict(copy.deepcopy(sd))                # (load_state_dict keeps the tensors it is given: each run gets its own)
        steps(n2, s2, o2, range(2, 4))
        resumed[name] = n2.state_dict()
        if shard:
            st = o2.state[n2.head.weight]
            assert "exp_avg" not in st and st["step"] == 4 and len(st["shards"]) == 14
    # torch.optim.Adam takes the consolidated state as it is (the reference's optimizer, roadmap_bce_v2.py:154-157)
    n3, _, _ = make(False)
    torch.optim.Adam(n3.parameters(), lr=1e-2).load_state_dict(copy.deepcopy(sd))
    final = net.state_dict()
    for k, v in ref_net.state_dict().items():
        assert torch.equal(final[k], v), f"uninterrupted sharded run: {k}"
        for name, got in resumed.items():
            assert torch.equal(got[k], v), f"resumed {name} run: {k} (max diff {float((got[k] - v).abs().max())})"
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_consolidated_optimizer_state_resumes_in_either_mode(tmp_path):
    out = str(tmp_path / "ok.pt")
    mp.spawn(_ckpt_worker, args=(2, free_port(), out), nprocs=2, join=True)
    assert torch.load(out)["ok"]


def _leave_shard_mode_worker(rank, world, port, out):
    """Steps in shard mode, then the same optimizer goes on under an all-reduce GradSync... that needs the consolidation collective while
    the sharded GradSync is still attached: HipAdam._update does it on the first whole update."""
    torch.set_num_threads(1)
    from driving_dirty_amd import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = torch.nn.Linear(64, 128)
    torch.manual_seed(5)
    with torch.no_grad():
        net.weight.copy_(torch.randn(128, 64) * 0.1)
    sync = ddp.GradSync(net, big_numel=4096, chunk_numel=4096, shard_optimizer=True)
    opt = _torch_adam_cls()(net.parameters(), lr=1e-2)
    opt.attach(sync)
    g = torch.Generator().manual_seed(70 + rank)
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        net(torch.randn(4, 64, generator=g)).square().mean().backward()
        sync.finish()
        opt.step(grad_scale=sync.grad_scale)
    assert ddp.PARAM_WAITS                                   # the last step's all-gathers are still writing into the weight ...
    net.state_dict()                                         # ... and reading the parameters as a whole waits for them (state_dict pre-hook)
    assert not ddp.PARAM_WAITS and not sync._gathers
    assert "shards" in opt.state[net.weight]
    m_whole, v_whole = opt._whole_moments(net.weight, opt.state[net.weight])
    net.zero_grad(set_to_none=True)
    net(torch.randn(4, 64, generator=g)).square().mean().backward()
    dist.all_reduce(net.weight.grad)                         # a whole gradient, outside the hooks' reach: the whole-tensor update
    sync._shards.pop(net.weight, None)
    before = {k: v.clone() for k, v in (("m", m_whole), ("v", v_whole))}
    opt._update(net.weight, opt.param_groups[0], 0.5)
    st = opt.state[net.weight]
    assert "shards" not in st and st["step"] == 3
    gsum = net.weight.grad * 0.5
    assert torch.allclose(st["exp_avg"], before["m"] * 0.9 + gsum * (1.0 - 0.9), rtol=0, atol=1e-7)
    if rank == 0:
        torch.save({"ok": True}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_whole_update_after_shard_mode_consolidates_the_moments(tmp_path):
    out = str(tmp_path / "ok.pt")
    mp.spawn(_leave_shard_mode_worker, args=(2, free_port(), out), nprocs=2, join=True)
    assert torch.load(out)["ok"]
