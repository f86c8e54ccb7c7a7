import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from driving_dirty_amd import gconv, synth
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double()-b.double()).abs().max()/b.double().abs().max())
def nhwc(t, cs):
    b,c,h,w = t.shape; o = torch.zeros(b,h,w,cs, device=t.device); o[...,:c] = t.permute(0,2,3,1); return o
for name, mod, shape in [("up4", nn.ConvTranspose2d(16, 8, 7, dilation=3), (1,16,382,382)),
                         ("up3", nn.ConvTranspose2d(32, 16, 7, dilation=7), (1,32,340,340)),
                         ("up4s", nn.ConvTranspose2d(16, 8, 7, dilation=3), (1,16,40,382)),
                         ("up4t", nn.ConvTranspose2d(16, 8, 7, dilation=3), (1,16,382,40))]:
    mod = synth.fill_module(mod, seed=3).double().to(dev)
    x = torch.rand(shape, device=dev, dtype=torch.float64, requires_grad=True)
    y = mod(x); gy = torch.randn_like(y); y.backward(gy)
    L = gconv.Layer(mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, mod.dilation, mod.padding, transposed=True)
    b,cin,h,w = shape; oh, ow = L.out_hw(h,w)
    xb = nhwc(x.detach().float(), cin); gb = nhwc(gy.float(), mod.out_channels)
    wd = mod.weight.detach().float()
    yb = torch.zeros(b,oh,ow,mod.out_channels, device=dev)
    L.forward(wd, mod.bias.detach().float(), gconv.View(xb), gconv.View(yb), gconv.EPI_BIAS)
    dxb = torch.zeros(b,h,w,cin, device=dev)
    L.backward_data(wd, gconv.View(gb), gconv.View(dxb))
    dw, db = L.backward_weight(gconv.View(xb), gconv.View(gb))
    e = (dxb.permute(0,3,1,2).double()-x.grad).abs()
    print(name, "fwd", rel(yb.permute(0,3,1,2), y), "dgrad", rel(dxb.permute(0,3,1,2), x.grad), "wgrad", rel(dw, mod.weight.grad), "bias", rel(db, mod.bias.grad))
    idx = (e > 1e-4*float(x.grad.abs().max())).nonzero()
    print("   bad dgrad elems:", idx.shape[0], idx[:6].tolist(), idx[-3:].tolist() if idx.shape[0] else "")
