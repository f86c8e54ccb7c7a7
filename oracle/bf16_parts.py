"""Oracle for the bf16 mixed-precision conv stack (BASELINE config 5), CPU, plain torch.

PARITY UNPINNED: the reference has no mixed-precision mode (it trains in fp32, src/autoencoder/components.py:41-47), so
there is no reference run to capture fixtures from.  This file states the contract the kernels are held to -- the
reference's conv stack evaluated "as torch autocast would": every conv sees bf16-rounded inputs and bf16-rounded
weights, accumulates in (at least) fp32 and its output is rounded to bf16 once; gradients flow the same way (the
gradient w.r.t. a conv's pre-activation is stored in bf16, weight/bias gradients are fp32 sums of bf16 products).
The arithmetic between the rounding points runs in fp64 here, so the only freedom left to an implementation is the
fp32 summation order (1 bf16 ulp on a value that lands on a rounding boundary).

Test infrastructure only -- see ``oracle/__init__.py``.
"""
import torch
from torch.nn import functional as F
from torch.nn import grad as nngrad


def bf16r(t):
    """Round to the nearest bf16 (ties to even), keep the container dtype."""
    return t.to(torch.bfloat16).to(t.dtype)


class _ConvReluBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride):
        xr, wr = bf16r(x.float()).double(), bf16r(w.float()).double()
        y = bf16r(F.relu(F.conv2d(xr, wr, b.double(), stride=stride, padding=1)).float()).double()
        ctx.save_for_backward(xr, wr, y)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, gy):
        xr, wr, y = ctx.saved_tensors
        gz = bf16r((gy * (y > 0)).float()).double()          # gradient w.r.t. the pre-activation, stored in bf16
        dw = nngrad.conv2d_weight(xr, wr.shape, gz, stride=ctx.stride, padding=1)
        dx = nngrad.conv2d_input(xr.shape, wr, gz, stride=ctx.stride, padding=1)
        return dx, dw, gz.sum(dim=(0, 2, 3)), None


def conv_stack_pooled(x, c1, c2, c3):
    """x [B,3,H,W] -> pooled [B, 32*Ho*Wo/4] (fp64 container), modules c1..c3 hold fp32 master weights."""
    a1 = _ConvReluBf16.apply(x.double(), c1.weight.double(), c1.bias, 1)
    a2 = _ConvReluBf16.apply(a1, c2.weight.double(), c2.bias, 1)
    a3 = _ConvReluBf16.apply(a2, c3.weight.double(), c3.bias, 2)
    return F.max_pool1d(a3.reshape(a3.size(0), 1, -1), 4).squeeze(1), (a1, a2, a3)


def encoder_latent(enc, wide, masks=(None, None)):
    """``oracle.ae_parts.EncoderNet`` with its conv stack evaluated in the mixed-precision contract above and the FC
    tail in the container dtype of ``enc`` (double it for an fp64 tail)."""
    pooled, _ = conv_stack_pooled(wide, enc.c1, enc.c2, enc.c3)
    pooled = pooled.to(enc.fc_z_out.weight.dtype)
    return enc.fc_z_out(enc.fc2(enc.fc1(pooled, masks[0]), masks[1]))
