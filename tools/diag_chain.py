import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from driving_dirty_amd import synth, gconv, ops, _lib
from driving_dirty_amd.gconv import View, _p, _stream
from driving_dirty_amd.heads import MergeFn
from oracle import spatial_parts
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double()-b.double()).abs().max()/b.double().abs().max())
ref = synth.fill_module(spatial_parts.RoadBoxMergeNet(), seed=6).double().to(dev)
cat = torch.relu(synth.hash_uniform((1,96,256,256), 77, -0.5, 1.0)).double().to(dev)
# oracle chain from the concat buffer on
acts = [cat.clone().requires_grad_(True)]
for l in (ref.up_conv_1, ref.up_conv_2, ref.up_conv_3, ref.up_conv_4):
    a = F.relu(l(acts[-1])); a.retain_grad(); acts.append(a)
pred = torch.sigmoid(ref.up_conv_5(acts[-1]))
wy = synth.hash_uniform(tuple(pred.shape), synth.key_salt("sp_wy")).double().to(dev)
(pred*wy).sum().backward()
# ours
nh = lambda t: t.detach().float().permute(0,2,3,1).contiguous()
mine = [nh(cat)]
ups = MergeFn.UPS_RM
ws = [getattr(ref, f"up_conv_{i}").weight.detach().float() for i in range(1,6)]
bs = [getattr(ref, f"up_conv_{i}").bias.detach().float() for i in range(1,6)]
for L, w, b in zip(ups, ws, bs):
    s = mine[-1]; oh, ow = L.out_hw(s.shape[1], s.shape[2]); d = torch.empty(1, oh, ow, L.cout, device=dev)
    L.forward(w, b, View(s), View(d), gconv.EPI_BIAS_RELU); mine.append(d)
for i in range(5): print("act", i, rel(mine[i].permute(0,3,1,2), acts[i]), "frac zero", float((mine[i]==0).float().mean()))
u = mine[-1]
probs = torch.empty(1, 800, 800, device=dev)
_lib.check(_lib.lib().dd_deconv2x2_c1_fwd(_p(u), _p(ws[4]), _p(bs[4]), _p(probs), 1, 400, 400, 8, _stream()), "f")
print("probs", rel(probs, pred[:,0]))
gu = torch.empty_like(u); dwl = torch.empty_like(ws[4]); dbl = torch.empty(1, device=dev)
wsp = torch.empty(_lib.lib().dd_deconv2x2_c1_workspace_bytes(8), device=dev, dtype=torch.uint8)
_lib.check(_lib.lib().dd_deconv2x2_c1_bwd(_p(u), _p(ws[4]), _p(probs), _p(wy[:,0].float().contiguous()), _p(gu), _p(dwl), _p(dbl), 1, 400, 400, 8, _p(wsp), _stream()), "b")
# oracle grads are wrt post-ReLU activations; ours are masked by (act > 0)
def masked(i): return acts[i].grad * (acts[i] > 0)
print("g_u4", rel(gu.permute(0,3,1,2), masked(4)))
g = gu
for i in range(3, -1, -1):
    L, src = ups[i], mine[i]
    gs = torch.empty_like(src)
    L.backward_data(ws[i], View(g), View(gs), relu_src=src)
    e = (gs.permute(0,3,1,2).double() - masked(i)).abs()
    bad = (e > 1e-4 * float(masked(i).abs().max()))
    print(f"g_act{i}", rel(gs.permute(0,3,1,2), masked(i)), "bad elems", int(bad.sum()), "of", bad.numel())
    if int(bad.sum()):
        idx = bad.nonzero()[:5]
        for t in idx.tolist():
            print("    ", t, "ours", float(gs[t[0], t[2], t[3], t[1]]), "ref", float(masked(i)[tuple(t)]), "act ours", float(src[t[0], t[2], t[3], t[1]]), "act ref", float(acts[i][tuple(t)]))
    g = gs
