import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from driving_dirty_amd import synth
from driving_dirty_amd.spatial import RoadMapBoxesMergingCNN, SpatialMappingCNN
g = np.load("tests/golden/spatial_heads.npz")
dev = torch.device("cuda:0")
def rel(got, ref): 
    got, ref = got.detach().double().cpu(), ref.double()
    return float((got-ref).abs().max()/ref.abs().max().clamp_min(1e-30))
def samp(t, idx): return t.detach().reshape(-1)[torch.from_numpy(idx).to(t.device)]
sm = synth.fill_module(SpatialMappingCNN(), seed=5).to(dev)
rb = synth.fill_module(RoadMapBoxesMergingCNN(), seed=6).to(dev)
views = synth.camera_batch(1, seed=5).to(dev)
rm = synth.road_maps(1, seed=5).float().unsqueeze(1).to(dev)
ssr = synth.hash_uniform((1, 32, 128, 918), synth.key_salt("ssr"), 0.0, 1.0).to(dev).requires_grad_(True)
space = sm(views); pred = rb(ssr, space, rm)
wy = synth.hash_uniform(tuple(pred.shape), synth.key_salt("sp_wy")).to(dev)
(pred * wy).sum().backward()
for name, m in (("space", sm), ("rboxm", rb)):
    for k, p in m.named_parameters():
        key = f"grad.{name}.{k}" if f"grad.{name}.{k}_f64" in g.files else f"gradsamp.{name}.{k}"
        ref64, ref32 = torch.from_numpy(g[key+"_f64"]), torch.from_numpy(g[key+"_f32"])
        got = p.grad if key.startswith("grad.") else samp(p.grad, g[f"gradidx.{name}.{k}"])
        print(f"{key:40s} ours_vs_f64 {rel(got, ref64):.2e}  ref32_vs_f64 {rel(ref32, ref64):.2e}  peak {float(ref64.abs().max()):.3e}")
print("ssrgrad", rel(samp(ssr.grad, g["ssrgrad_idx"]), torch.from_numpy(g["ssrgrad_samp_f64"])), rel(torch.from_numpy(g["ssrgrad_samp_f32"]), torch.from_numpy(g["ssrgrad_samp_f64"])))
print("pred", rel(samp(pred, g["pred_idx"]), torch.from_numpy(g["pred_samp_f64"])))
