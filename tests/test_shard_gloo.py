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
