"""Does gloo serve reduce_scatter_tensor / all_gather_into_tensor (in place) on CUDA tensors?  Two ranks on cuda:0."""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    x = torch.arange(8 * world, dtype=torch.float32, device=dev) * (rank + 1)
    out = torch.empty(8, device=dev)
    try:
        dist.reduce_scatter_tensor(out, x, async_op=True).wait()
        torch.cuda.synchronize()
        print(rank, "rs ok", out.tolist(), flush=True)
    except Exception as e:      # noqa: BLE001
        print(rank, "rs FAIL", type(e).__name__, str(e)[:300], flush=True)
    full = torch.zeros(8 * world, device=dev)
    mine = full[rank * 8:(rank + 1) * 8]
    mine.fill_(rank + 1)
    try:
        dist.all_gather_into_tensor(full, mine, async_op=True).wait()
        torch.cuda.synchronize()
        print(rank, "ag ok", full[::8].tolist(), flush=True)
    except Exception as e:      # noqa: BLE001
        print(rank, "ag FAIL", type(e).__name__, str(e)[:300], flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(w, args=(2, 29871), nprocs=2, join=True)
