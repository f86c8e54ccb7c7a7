"""world_size-2 gloo test of the data-parallel gradient path (runs on CPU; the N>1 GPU path is the same code
over RCCL).  Checks that GradSync + a 1/world scale reproduces single-process full-batch gradients."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    port = 29500 + os.getpid() % 2000
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
