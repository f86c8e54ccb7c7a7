import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from driving_dirty_amd.optim import HipAdam
from driving_dirty_amd.ddp import GradSync
dev = torch.device("cuda:0")
model = bench.build_model(dev)
model.training_step(bench.synthetic_batch(dev, 2, 0), 0)["loss"].backward()
model.zero_grad(set_to_none=True)
opt = HipAdam(model.parameters(), lr=1e-3)
sync = GradSync(model)
opt.overlap_with_backward(grad_scale=1.0)
batch = bench.synthetic_batch(dev, 32, 0)
for i in range(120):
    model.zero_grad(set_to_none=True)
    out = model.training_step(batch, i)
    out["loss"].backward()
    sync.finish()
    opt.step()
    if i in (5, 20, 60, 119):
        torch.cuda.synchronize()
        print(i, "alloc GB", round(torch.cuda.memory_allocated() / 2**30, 3), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 3), "peak GB", round(torch.cuda.max_memory_allocated() / 2**30, 3), "loss", round(float(out["loss"]), 5), flush=True)
