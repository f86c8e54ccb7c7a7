import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench, torch
from argparse import Namespace
from driving_dirty_amd.autoencoder import BasicAE
from driving_dirty_amd.optim import HipAdam
from driving_dirty_amd.roadmap import RoadMapBCE
dev = torch.device("cuda:0")
H, W = bench.H, bench.W
h2, w2, b5 = 2 * H, 2 * W, 16
torch.manual_seed(1)
ae = BasicAE(Namespace(hidden_dim=bench.HIDDEN, latent_dim=bench.LATENT, input_height=h2, input_width=6 * w2, output_height=h2, output_width=w2))
m = RoadMapBCE(Namespace(pretrained_ae=ae, precision="bf16", unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=10 ** 9)).to(dev)
batch = (tuple(torch.rand(b5, 6, 3, h2, w2, device=dev)), None, tuple(torch.rand(b5, 800, 800, device=dev) < 0.3))
m.training_step(batch, 0)["loss"].backward()
m.zero_grad(set_to_none=True)
opt = HipAdam(m.parameters(), lr=1e-3)
opt.overlap_with_backward()
def step(i):
    m.zero_grad(set_to_none=True)
    m.training_step(batch, i)["loss"].backward()
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6): step(i)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 6 * 1e3)
