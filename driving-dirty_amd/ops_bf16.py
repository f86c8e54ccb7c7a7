"""bf16 mixed-precision conv stack (BASELINE config 5) over the C ABI's ``dd_*bf16*`` entry points.

Activations / activation gradients: NHWC ``torch.bfloat16`` tensors (the ABI sees their raw uint16 storage);
weights, biases and their gradients: fp32.  Rounding happens exactly once, in the kernel that writes a tensor.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check
from .ops import PACK_DGRAD_S1, PACK_DGRAD_S2, PACK_FWD, _p, _stream, conv_desc, conv_out


def _bf(t, name, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise _lib.HotpathError(f"{name}: expected a contiguous bf16 device tensor, got {getattr(t, 'dtype', type(t))} "
                                f"on {getattr(t, 'device', '?')}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.HotpathError(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
    return t


def stitch6_bf16(views):
    """[B,6,3,H,W] fp32 -> wide NHWC4 bf16 [B,H,6W,4] (view order [0,1,2,5,4,3], channel 3 zero)."""
    b, n, c, h, w = views.shape
    if n != 6 or c != 3:
        raise _lib.HotpathError(f"stitch6_bf16: expected [B,6,3,H,W], got {tuple(views.shape)}")
    ops._dev(views, "views")
    out = torch.empty((b, h, 6 * w, 4), device=views.device, dtype=torch.bfloat16)
    check(_lib.lib().dd_stitch6_bf16(_p(views), _p(out), b, h, w, _stream()), "dd_stitch6_bf16")
    return out


def stitch6_bf16_samples(samples):
    """Tuple of B per-sample [6,3,H,W] fp32 tensors (the reference's collate, helper.py:22-23) -> wide NHWC4 bf16, no stack copy."""
    b = len(samples)
    n, c, h, w = samples[0].shape
    if n != 6 or c != 3:
        raise _lib.HotpathError(f"stitch6_bf16_samples: expected samples of [6,3,H,W], got {tuple(samples[0].shape)}")
    for t in samples:
        ops._dev(t, "sample", (6, 3, h, w))
    table = (C.c_void_p * b)(*[t.data_ptr() for t in samples])
    out = torch.empty((b, h, 6 * w, 4), device=samples[0].device, dtype=torch.bfloat16)
    check(_lib.lib().dd_stitch6_bf16_ptrs(table, _p(out), b, h, w, _stream()), "dd_stitch6_bf16_ptrs")
    return out


def stitch6_bf16_u8(sample):
    """uint8 frames ([B,6,H,W,3] or a tuple of [6,H,W,3]) -> wide NHWC4 bf16: /255 (true division) and the bf16 rounding fused with
    the gather -- the values ``stitch6_bf16(frames.permute(..).float() / 255)`` gives."""
    table, b, h, w, dev, _keep = ops.u8_table(sample, "stitch6_bf16_u8")
    out = torch.empty((b, h, 6 * w, 4), device=dev, dtype=torch.bfloat16)
    check(_lib.lib().dd_stitch6_bf16_u8_ptrs(table, _p(out), b, h, w, _stream()), "dd_stitch6_bf16_u8_ptrs")
    return out


def to_bf16(t):
    ops._dev(t, "t")
    if t.numel() % 4:
        raise _lib.HotpathError("to_bf16: element count must be a multiple of 4")
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    check(_lib.lib().dd_f32_to_bf16(_p(t), _p(out), t.numel(), _stream()), "dd_f32_to_bf16")
    return out


def to_f32(t):
    _bf(t, "t")
    if t.numel() % 4:
        raise _lib.HotpathError("to_f32: element count must be a multiple of 4")
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    check(_lib.lib().dd_bf16_to_f32(_p(t), _p(out), t.numel(), _stream()), "dd_bf16_to_f32")
    return out


def conv_pack(weight, desc, kind):
    ops._dev(weight, "weight", (32, desc.cin_real, 3, 3))
    n = _lib.lib().dd_conv_bf16_packed_elems(C.byref(desc))
    if n <= 0:
        raise _lib.HotpathError(f"conv_bf16_pack: {_lib.lib().dd_last_error().decode()}")
    packed = torch.empty(n, device=weight.device, dtype=torch.bfloat16)
    check(_lib.lib().dd_conv_bf16_pack(_p(weight), C.byref(desc), kind, _p(packed), _stream()), "dd_conv_bf16_pack")
    return packed


def conv_fwd(x, packed, bias, desc, want_bits=True):
    """bf16(relu(conv(x) + bias)) and (optionally) the ReLU signs as one uint32 per pixel."""
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _bf(x, "x", (desc.batch, desc.height, desc.width, desc.cin_store))
    _bf(packed, "packed")
    ops._dev(bias, "bias", (32,))
    y = torch.empty((desc.batch, ho, wo, 32), device=x.device, dtype=torch.bfloat16)
    bits = torch.empty((desc.batch, ho, wo), device=x.device, dtype=torch.int32) if want_bits else None
    check(_lib.lib().dd_conv_bf16_fwd(_p(x), _p(packed), _p(bias), _p(y), _p(bits), C.byref(desc), _stream()), "dd_conv_bf16_fwd")
    return y, bits


def conv_dgrad(dy, packed_dgrad, bits, desc):
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _bf(dy, "dy", (desc.batch, ho, wo, 32))
    _bf(packed_dgrad, "packed")
    if not (bits.is_cuda and bits.dtype == torch.int32 and bits.is_contiguous() and tuple(bits.shape) == (desc.batch, desc.height, desc.width)):
        raise _lib.HotpathError("conv_bf16_dgrad: relu_bits must be a contiguous int32 [B,H,W] device tensor")
    dx = torch.empty((desc.batch, desc.height, desc.width, 32), device=dy.device, dtype=torch.bfloat16)
    check(_lib.lib().dd_conv_bf16_dgrad(_p(dy), _p(packed_dgrad), _p(bits), _p(dx), C.byref(desc), _stream()), "dd_conv_bf16_dgrad")
    return dx


def conv_wgrad(x, dy, desc):
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _bf(x, "x", (desc.batch, desc.height, desc.width, desc.cin_store))
    _bf(dy, "dy", (desc.batch, ho, wo, 32))
    nbytes = _lib.lib().dd_conv_bf16_wgrad_workspace_bytes(C.byref(desc))
    if nbytes <= 0:
        raise _lib.HotpathError(f"conv_bf16_wgrad: {_lib.lib().dd_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    dw = torch.empty((32, desc.cin_real, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty(32, device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_bf16_wgrad(_p(x), _p(dy), _p(dw), _p(db), C.byref(desc), _p(ws), nbytes, _stream()), "dd_conv_bf16_wgrad")
    return dw, db


def pool4_fwd(feat):
    b, h, w, c = feat.shape
    _bf(feat, "feat")
    out = torch.empty((b, (c * h * w) // 4), device=feat.device, dtype=torch.float32)
    check(_lib.lib().dd_pool4_bf16_fwd(_p(feat), _p(out), b, h, w, c, _stream()), "dd_pool4_bf16_fwd")
    return out


def pool4_relu_bwd(dpooled, feat):
    b, h, w, c = feat.shape
    ops._dev(dpooled, "dpooled", (b, (c * h * w) // 4))
    _bf(feat, "feat")
    out = torch.empty_like(feat)
    check(_lib.lib().dd_pool4_relu_bf16_bwd(_p(dpooled), _p(feat), _p(out), b, h, w, c, _stream()), "dd_pool4_relu_bf16_bwd")
    return out


def pool4_has_idx(h, w, c):
    """Whether the tiled pool with routing codes takes this feature map (C == 32, H*W % 4 == 0)."""
    return c == 32 and (h * w) % 4 == 0


def pool4_fwd_idx(feat):
    """pooled, codes: max_pool1d(4) + the backward's routing (dd_pool4_bf16_fwd_idx)."""
    b, h, w, c = feat.shape
    _bf(feat, "feat")
    n = _lib.lib().dd_pool4_bf16_idx_elems(b, h, w, c)
    if n < 0:
        raise _lib.HotpathError(_lib.lib().dd_last_error().decode())
    out = torch.empty((b, (c * h * w) // 4), device=feat.device, dtype=torch.float32)
    idx = torch.empty(n, device=feat.device, dtype=torch.int16)
    check(_lib.lib().dd_pool4_bf16_fwd_idx(_p(feat), _p(out), _p(idx), b, h, w, c, _stream()), "dd_pool4_bf16_fwd_idx")
    return out, idx


def pool4_idx_relu_bwd(dpooled, idx, shape):
    b, h, w, c = shape
    ops._dev(dpooled, "dpooled", (b, (c * h * w) // 4))
    if idx.dtype != torch.int16 or idx.numel() != b * (h * w // 4) * (c // 4) or not idx.is_cuda:
        raise _lib.HotpathError(f"pool4_idx_relu_bwd (bf16): bad routing codes {tuple(idx.shape)} {idx.dtype}")
    out = torch.empty(shape, device=dpooled.device, dtype=torch.bfloat16)
    check(_lib.lib().dd_pool4_idx_relu_bf16_bwd(_p(dpooled), _p(idx), _p(out), b, h, w, c, _stream()), "dd_pool4_idx_relu_bf16_bwd")
    return out


class EncoderConvStackBf16(torch.autograd.Function):
    """c1 -> ReLU -> c2 -> ReLU -> c3 (stride 2) -> ReLU -> NCHW-order max_pool1d(4), bf16 operands / fp32 accumulation.

    Reference arithmetic: Encoder.forward, src/autoencoder/components.py:41-47 (the reference itself is fp32 only; the
    rounding points are those of torch autocast: every conv output is rounded to bf16 once).
    forward(x4 bf16 [B,H,W,4], w1,b1,w2,b2,w3,b3 fp32, pool) -> pooled fp32 [B, 32*Ho*Wo/4], or (pool = False: the ``c3_only``
    exit, components.py:44-45) the bf16-rounded c3 feature as an fp32 NHWC tensor [B,Ho,Wo,32] for the fp32 box heads.
    """

    @staticmethod
    def forward(ctx, x4, w1, b1, w2, b2, w3, b3, pool=True):
        b, h, w, _ = x4.shape
        d1, d2, d3 = conv_desc(b, h, w, 3, 1), conv_desc(b, h, w, 32, 1), conv_desc(b, h, w, 32, 2)
        a1, s1 = conv_fwd(x4, conv_pack(w1, d1, PACK_FWD), b1, d1)
        a2, s2 = conv_fwd(a1, conv_pack(w2, d2, PACK_FWD), b2, d2)
        a3, _ = conv_fwd(a2, conv_pack(w3, d3, PACK_FWD), b3, d3, want_bits=False)
        # the backward's operand images are packed here (see ops.EncoderConvStack.forward)
        need = ctx.needs_input_grad
        p2d = p3d = torch.empty(0, device=x4.device, dtype=torch.bfloat16)
        if need[1] or need[2] or need[3] or need[4]:
            p3d = conv_pack(w3, d3, PACK_DGRAD_S2)
        if need[1] or need[2]:
            p2d = conv_pack(w2, d2, PACK_DGRAD_S1)
        ctx.pool = bool(pool)
        ctx.a3_shape = tuple(a3.shape)
        ctx.pool_idx = ctx.pool and pool4_has_idx(*a3.shape[1:])
        if ctx.pool_idx:      # the pool's backward runs from 30 MB of routing codes: the 0.48 GB feature is neither kept nor read again
            out, codes = pool4_fwd_idx(a3)
            ctx.save_for_backward(x4, a1, a2, codes, p2d, p3d, s1, s2)
            return out
        ctx.save_for_backward(x4, a1, a2, a3, p2d, p3d, s1, s2)
        return pool4_fwd(a3) if pool else to_f32(a3)

    @staticmethod
    def backward(ctx, grad_out):
        x4, a1, a2, a3, p2d, p3d, s1, s2 = ctx.saved_tensors
        b, h, w, _ = x4.shape
        d1, d2, d3 = conv_desc(b, h, w, 3, 1), conv_desc(b, h, w, 32, 1), conv_desc(b, h, w, 32, 2)
        if ctx.pool_idx:
            g3 = pool4_idx_relu_bwd(grad_out.contiguous(), a3, ctx.a3_shape)      # a3 holds the routing codes here
        elif ctx.pool:
            g3 = pool4_relu_bwd(grad_out.contiguous(), a3)
        else:      # feature exit: ReLU mask in fp32, then the one rounding to bf16 the contract prescribes for a pre-activation gradient
            g3 = to_bf16(ops.relu_bwd(grad_out.contiguous(), to_f32(a3)))
        need = ctx.needs_input_grad
        dw3, db3 = conv_wgrad(a2, g3, d3) if (need[5] or need[6]) else (None, None)
        dw2 = db2 = dw1 = db1 = None
        if need[1] or need[2] or need[3] or need[4]:
            g2 = conv_dgrad(g3, p3d, s2, d3)
            del g3
            for hook in ops.MFMA_PHASE_HOOKS:
                hook()
            if need[3] or need[4]:
                dw2, db2 = conv_wgrad(a1, g2, d2)
            if need[1] or need[2]:
                g1 = conv_dgrad(g2, p2d, s1, d2)
                del g2
                dw1, db1 = conv_wgrad(x4, g1, d1)
        return None, dw1, db1, dw2, db2, dw3, db3, None


def encoder_conv_stack(x4, c1, c2, c3, pool=True):
    return EncoderConvStackBf16.apply(x4, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias, pool)
