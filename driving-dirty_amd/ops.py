"""torch.autograd.Function shims over the C ABI (``include/dd_hotpath.h``).

The shims own every tensor (the kernels never allocate), pass raw device pointers plus the
current PyTorch HIP stream, and raise on any non-zero return code.  Activations between conv
layers are NHWC fp32 tensors ``[B,H,W,C]``; the module layer (components.py) presents them to
callers as NCHW-shaped channels_last views, which is what the reference returns logically.
"""
import ctypes as C
import os

import torch

from . import _lib
from . import ddp as _ddp
from ._lib import ConvDesc, check

EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_RELU_MASK = 0, 1, 2, 3
PACK_FWD, PACK_DGRAD_S1, PACK_DGRAD_S2 = 0, 1, 2


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device pointer of a kernel operand.  A parameter whose all-gather (sharded optimizer, ddp.GradSync) is still in flight
    is waited for -- on the current stream -- the first time it is handed to a kernel."""
    if t is None:
        return None
    ptr = t.data_ptr()
    if _ddp.PARAM_WAITS:
        wait = _ddp.PARAM_WAITS.pop(ptr, None)
        if wait is not None:
            wait()
    return C.c_void_p(ptr)


def _dev(t, name, shape=None):
    """Validate a kernel operand on the HOST before any launch (a faulting kernel can reset the GPU)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.HotpathError(f"{name}: expected a contiguous fp32 device tensor, got "
                                f"{getattr(t, 'dtype', type(t))} on {getattr(t, 'device', '?')} "
                                f"contiguous={getattr(t, 'is_contiguous', lambda: '?')()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.HotpathError(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
    return t


def conv_out(n, stride):
    return (n + 2 - 3) // stride + 1


def conv_desc(batch, h, w, cin_real, stride, rows_per_task=0):
    return ConvDesc(batch, h, w, cin_real, 4 if cin_real == 3 else cin_real, 32, 3, stride, 1, rows_per_task)


# ------------------------------------------------------------------------------------------------ layout
def stitch6(views, mask_slot=-1, want_nhwc4=True, want_nchw=False, want_target=False):
    """[B,6,3,H,W] -> wide image (view order [0,1,2,5,4,3]); see dd_stitch6 in dd_hotpath.h."""
    b, n, c, h, w = views.shape
    if n != 6 or c != 3:
        raise _lib.HotpathError(f"stitch6: expected [B,6,3,H,W], got {tuple(views.shape)}")
    _dev(views, "views")
    wide4 = torch.empty((b, h, 6 * w, 4), device=views.device, dtype=torch.float32) if want_nhwc4 else None
    wide = torch.empty((b, 3, h, 6 * w), device=views.device, dtype=torch.float32) if want_nchw else None
    tgt = torch.empty((b, 3, h, w), device=views.device, dtype=torch.float32) if want_target else None
    check(_lib.lib().dd_stitch6(_p(views), _p(wide4), _p(wide), _p(tgt), b, h, w, int(mask_slot), _stream()), "dd_stitch6")
    return wide4, wide, tgt


def stitch6_samples(samples):
    """Tuple of B per-sample [6,3,H,W] tensors (the reference's collate, helper.py:22-23) -> wide NHWC4, no stack copy."""
    import ctypes
    b = len(samples)
    n, c, h, w = samples[0].shape
    for t in samples:
        _dev(t, "sample", (6, 3, h, w))
    if n != 6 or c != 3:
        raise _lib.HotpathError(f"stitch6_samples: expected samples of [6,3,H,W], got {tuple(samples[0].shape)}")
    table = (ctypes.c_void_p * b)(*[t.data_ptr() for t in samples])
    wide4 = torch.empty((b, h, 6 * w, 4), device=samples[0].device, dtype=torch.float32)
    check(_lib.lib().dd_stitch6_ptrs(table, _p(wide4), b, h, w, _stream()), "dd_stitch6_ptrs")
    return wide4


def boxes_to_binary_map(box_sets, device=None):
    """List of per-sample [n,2,4] corner tensors (f64 as the dataset holds them, or f32) -> [B,800,800] fp32 0/1 maps:
    boxes_to_binary_map (bb_to_img.py:5-20) for the whole batch in one launch, bit-identical to the Pillow fill."""
    import ctypes
    b = len(box_sets)
    if b == 0:
        raise _lib.HotpathError("boxes_to_binary_map: empty batch")
    dtype = box_sets[0].dtype
    if dtype not in (torch.float64, torch.float32) or any(t.dtype != dtype for t in box_sets):
        raise _lib.HotpathError("boxes_to_binary_map: boxes must all be float64 or all float32")
    for t in box_sets:
        if t.dim() != 3 or tuple(t.shape[1:]) != (2, 4):
            raise _lib.HotpathError(f"boxes_to_binary_map: expected [n,2,4] boxes, got {tuple(t.shape)}")
    if device is None:
        device = next((t.device for t in box_sets if t.is_cuda), None)
    if device is None or torch.device(device).type != "cuda":
        raise _lib.HotpathError("boxes_to_binary_map: no GPU device given (the rasteriser has no CPU fallback)")
    counts = [int(t.shape[0]) for t in box_sets]
    offsets = (ctypes.c_int32 * (b + 1))(0, *[sum(counts[:i + 1]) for i in range(b)])
    flat = torch.cat([t.reshape(-1, 8) for t in box_sets], dim=0).to(device).contiguous()
    maps = torch.empty((b, 800, 800), device=device, dtype=torch.float32)
    check(_lib.lib().dd_boxes_to_binary_map(_p(flat) if flat.numel() else None, 0 if dtype == torch.float64 else 1, offsets,
                                            _p(maps), b, _stream()), "dd_boxes_to_binary_map")
    return maps


def stitch6_u8(frames):
    """frames [B,6,H,W,3] uint8 -> wide NHWC4 fp32 in [0,1] (ToTensor's /255 fused with the 6-view gather)."""
    b, n, h, w, c = frames.shape
    if n != 6 or c != 3 or frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous():
        raise _lib.HotpathError(f"stitch6_u8: expected contiguous uint8 [B,6,H,W,3] on the GPU, got {tuple(frames.shape)} {frames.dtype}")
    out = torch.empty((b, h, 6 * w, 4), device=frames.device, dtype=torch.float32)
    check(_lib.lib().dd_stitch6_u8(_p(frames), _p(out), b, h, w, _stream()), "dd_stitch6_u8")
    return out


def is_u8_frames(sample):
    """uint8 camera frames as a decoder emits them: a [B,6,H,W,3] tensor or the collate's tuple / list of B [6,H,W,3] tensors."""
    if isinstance(sample, torch.Tensor):
        return sample.dtype == torch.uint8 and sample.dim() == 5
    return isinstance(sample, (tuple, list)) and len(sample) > 0 and all(
        isinstance(t, torch.Tensor) and t.dtype == torch.uint8 and t.dim() == 4 for t in sample)


def u8_table(sample, who):
    """-> (ctypes pointer table, B, H, W, device, keepalive) for uint8 frames, validated on the host (a faulting kernel can reset
    the GPU).  A [B,6,H,W,3] tensor is a table of its B slices: one code path for both forms."""
    import ctypes
    items = [sample[i] for i in range(sample.shape[0])] if isinstance(sample, torch.Tensor) else list(sample)
    if not items:
        raise _lib.HotpathError(f"{who}: empty batch")
    n, h, w, c = items[0].shape
    if n != 6 or c != 3:
        raise _lib.HotpathError(f"{who}: expected uint8 frames of [6,H,W,3] per sample, got {tuple(items[0].shape)}")
    keep = []
    for t in items:
        if not (t.is_cuda and t.dtype == torch.uint8 and tuple(t.shape) == (6, h, w, 3)):
            raise _lib.HotpathError(f"{who}: expected uint8 [6,{h},{w},3] frames on the GPU, got {tuple(t.shape)} {t.dtype} on {t.device}")
        keep.append(t if t.is_contiguous() else t.contiguous())
    table = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
    return table, len(keep), h, w, keep[0].device, keep


def stitch6_u8_samples(sample, mask_slot=-1, want_target=False):
    """uint8 frames ([B,6,H,W,3] or a tuple of [6,H,W,3]) -> wide NHWC4 fp32 in [0,1] (ToTensor's /255, data_helper.py:63-68, fused
    with the 6-view gather), optionally with BasicAE's masked-view task (-> (wide4, target [B,3,H,W]))."""
    table, b, h, w, dev, _keep = u8_table(sample, "stitch6_u8_samples")
    wide4 = torch.empty((b, h, 6 * w, 4), device=dev, dtype=torch.float32)
    tgt = torch.empty((b, 3, h, w), device=dev, dtype=torch.float32) if want_target else None
    check(_lib.lib().dd_stitch6_u8_ptrs(table, _p(wide4), _p(tgt), b, h, w, int(mask_slot), _stream()), "dd_stitch6_u8_ptrs")
    return (wide4, tgt) if want_target else wide4


def wide_image(sample, precision="fp32", mask_slot=-1, want_target=False):
    """Whatever the data pipeline hands over -> the wide NHWC4 image the conv stack reads (view order [0,1,2,5,4,3],
    roadmap_bce_v2.py:53-64), fp32 or bf16, in ONE pass:
      * fp32 [B,6,3,H,W] (autoencoder.py:53-57) or the collate's tuple of B fp32 [6,3,H,W] tensors (helper.py:22-23);
      * uint8 [B,6,H,W,3] or a tuple of B uint8 [6,H,W,3] decoded frames: ToTensor's /255 (data_helper.py:63-68) fused in.
    ``mask_slot`` / ``want_target``: the masked-view task of BasicAE.six_to_one_task (fp32 image only) -> (wide4, target)."""
    per_sample = isinstance(sample, (tuple, list))
    if precision == "bf16":
        from . import ops_bf16
        if mask_slot >= 0 or want_target:
            raise _lib.HotpathError("wide_image: the masked-view task is built for the fp32 image")
        if is_u8_frames(sample):
            return ops_bf16.stitch6_bf16_u8(sample)
        return ops_bf16.stitch6_bf16_samples([t.contiguous() for t in sample]) if per_sample else ops_bf16.stitch6_bf16(sample.contiguous())
    if is_u8_frames(sample):
        return stitch6_u8_samples(sample, mask_slot, want_target)
    if per_sample and mask_slot < 0 and not want_target:
        return stitch6_samples([t.contiguous() for t in sample])      # gather straight from the samples: no stack copy
    x = torch.stack(tuple(sample), dim=0) if per_sample else sample
    wide4, _, tgt = stitch6(x.contiguous(), mask_slot=mask_slot, want_target=want_target)
    return (wide4, tgt) if want_target else wide4


def threat_score(a, b, round_b=False):
    """compute_ts_road_map (helper.py:74-77) in one pass on the device."""
    _dev(a, "a")
    _dev(b, "b", a.shape)
    out = torch.empty((), device=a.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().dd_threat_score_workspace_bytes(), device=a.device, dtype=torch.uint8)
    check(_lib.lib().dd_threat_score(_p(a), _p(b), _p(out), a.numel(), int(round_b), _p(ws), _stream()), "dd_threat_score")
    return out


def nchw_to_nhwc(x, c_store):
    b, c, h, w = x.shape
    _dev(x, "x")
    out = torch.empty((b, h, w, c_store), device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_nchw_to_nhwc(_p(x), _p(out), b, c, h, w, c_store, _stream()), "dd_nchw_to_nhwc")
    return out


def subsample_nhwc4(x, stride, offset, oh, ow):
    """x [B,1,H,W] or [B,H,W] -> [B,oh,ow,4] with channel 0 = x[stride*u + offset, stride*v + offset] (zero outside)."""
    _dev(x, "x")
    b, h, w = x.shape[0], x.shape[-2], x.shape[-1]
    assert x.numel() == b * h * w, "subsample_nhwc4: one channel"
    out = torch.empty((b, oh, ow, 4), device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_subsample_nhwc4(_p(x.contiguous()), _p(out), b, h, w, oh, ow, stride, offset, _stream()), "dd_subsample_nhwc4")
    return out


def subsample_masks_nhwc4(masks, stride, offset, oh, ow):
    """Tuple / list of B per-sample bool / uint8 [H,W] masks (the collate's ``road_image`` tuple) -> [B,oh,ow,4] fp32 with
    channel 0 = float(mask[stride*u + offset, stride*v + offset]) (zero outside): ``subsample_nhwc4`` of
    ``torch.stack(masks).float()`` without the stack and the cast."""
    import ctypes
    b = len(masks)
    h, w = masks[0].shape[-2], masks[0].shape[-1]
    for t in masks:
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.bool, torch.uint8) and t.numel() == h * w):
            raise _lib.HotpathError("subsample_masks_nhwc4: expected contiguous bool / uint8 device masks of one size")
    out = torch.empty((b, oh, ow, 4), device=masks[0].device, dtype=torch.float32)
    table = (ctypes.c_void_p * b)(*[t.data_ptr() for t in masks])
    check(_lib.lib().dd_subsample_nhwc4_u8_ptrs(table, _p(out), b, h, w, oh, ow, stride, offset, _stream()), "dd_subsample_nhwc4_u8_ptrs")
    return out


def deconv2x2_c32_fwd(x, wt, bias, relu=True):
    """x [B,h,w,32] NHWC, wt [32,32,2,2] -> (relu)(ConvTranspose2d k2 s2) [B,2h,2w,32] NHWC, one launch."""
    b, h, w, c = x.shape
    assert c == 32 and tuple(wt.shape) == (32, 32, 2, 2) and x.is_contiguous() and wt.is_contiguous()
    out = torch.empty((b, 2 * h, 2 * w, 32), device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_deconv2x2_c32_fwd(_p(x), _p(wt), _p(bias), _p(out), b, h, w, int(relu), _stream()), "dd_deconv2x2_c32_fwd")
    return out


def deconv2x2_c32_fwd_into(x, wt, bias, out, coff, relu=True):
    """The same into channels [coff, coff+32) of ``out`` [B,2h,2w,C] NHWC (a concat buffer's slice)."""
    b, h, w, c = x.shape
    assert c == 32 and tuple(wt.shape) == (32, 32, 2, 2) and x.is_contiguous() and wt.is_contiguous()
    assert out.is_contiguous() and tuple(out.shape[:3]) == (b, 2 * h, 2 * w) and 0 <= coff and coff + 32 <= out.shape[3]
    check(_lib.lib().dd_deconv2x2_c32_fwd_slice(_p(x), _p(wt), _p(bias), _p(out), b, h, w, int(relu), out.shape[3], coff, _stream()),
          "dd_deconv2x2_c32_fwd_slice")


def ssconv_dgrad_ok(g, dx):
    """ss_conv's data gradient in one launch serves these tensors: dense 32-channel NHWC, gw = (xw - 24) / 7 + 1 <= 128."""
    return (g.dim() == 4 and dx.dim() == 4 and g.shape[3] == 32 and dx.shape[3] == 32 and g.is_contiguous() and dx.is_contiguous()
            and g.shape[:2] == dx.shape[:2] and bool(_lib.lib().dd_ssconv_dgrad_supported(g.shape[1], g.shape[2], dx.shape[2])))


def ssconv_dgrad(g, wt, dx):
    """g [B,h,gw,32], wt [32,32,1,24] (Conv2d weight) -> dx [B,h,xw,32] = dL/d(input) of Conv2d(32,32,(1,24),stride (1,7))."""
    assert tuple(wt.shape) == (32, 32, 1, 24) and wt.is_contiguous() and ssconv_dgrad_ok(g, dx)
    b, h, gw, _ = g.shape
    check(_lib.lib().dd_ssconv_dgrad(_p(g), _p(wt), _p(dx), b, h, gw, dx.shape[2], _stream()), "dd_ssconv_dgrad")


def ssconv_fwd(x, wt, bias, y, relu=True):
    """x [B,h,xw,32], wt [32,32,1,24], bias [32] or None -> y [B,h,gw,32] = (relu)(Conv2d(32,32,(1,24),stride (1,7))(x)), one launch."""
    assert tuple(wt.shape) == (32, 32, 1, 24) and wt.is_contiguous() and ssconv_dgrad_ok(y, x)
    b, h, xw, _ = x.shape
    check(_lib.lib().dd_ssconv_fwd(_p(x), _p(wt), _p(bias), _p(y), b, h, xw, y.shape[2], int(relu), _stream()), "dd_ssconv_fwd")


def conv1x1_c32_c3_nchw(x, wt, bias):
    """x [B,h,w,32] NHWC, wt [32,3,1,1] -> ConvTranspose2d k1 [B,3,h,w] NCHW."""
    b, h, w, c = x.shape
    assert c == 32 and tuple(wt.shape) == (32, 3, 1, 1) and x.is_contiguous() and wt.is_contiguous()
    out = torch.empty((b, 3, h, w), device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_conv1x1_c32_c3_nchw(_p(x), _p(wt), _p(bias), _p(out), b, h, w, _stream()), "dd_conv1x1_c32_c3_nchw")
    return out


def conv1ch_fwd(taps4, w, bias, relu=True):
    """taps4 [B,sh,sw,4] (channel 0), w [32,1,7,7] -> relu(conv + bias) [B,sh-6,sw-6,32] (NHWC)."""
    b, sh, sw, _ = taps4.shape
    assert tuple(w.shape) == (32, 1, 7, 7) and w.is_contiguous()
    y = torch.empty((b, sh - 6, sw - 6, 32), device=taps4.device, dtype=torch.float32)
    check(_lib.lib().dd_conv1ch_fwd(_p(taps4), _p(w), _p(bias), _p(y), b, sh, sw, int(relu), _stream()), "dd_conv1ch_fwd")
    return y


def conv1ch_wgrad(taps4, g):
    """-> (dw [32,1,7,7], db [32]) of conv1ch_fwd from g = dL/dy [B,sh-6,sw-6,32] (ReLU mask already applied)."""
    b, sh, sw, _ = taps4.shape
    assert tuple(g.shape) == (b, sh - 6, sw - 6, 32) and g.is_contiguous()
    dw = torch.empty((32, 1, 7, 7), device=g.device, dtype=torch.float32)
    db = torch.empty(32, device=g.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().dd_conv1ch_wgrad_workspace_bytes(), device=g.device, dtype=torch.uint8)
    check(_lib.lib().dd_conv1ch_wgrad(_p(taps4), _p(g), _p(dw), _p(db), b, sh, sw, _p(ws), _stream()), "dd_conv1ch_wgrad")
    return dw, db


def conv1ch_fwd_phase3(taps4, w, bias, relu=True):
    """conv1ch_fwd with the output in the phase-major layout of a dilation-3 consumer: -> (y [9B, ph, pw, 32], its sign words int32
    [9B, ph, pw]), ph = ceil((sh-6) / 3), pw likewise; image b * 9 + (i % 3) * 3 + j % 3 holds the pixels (i, j) of that residue class."""
    b, sh, sw, _ = taps4.shape
    assert tuple(w.shape) == (32, 1, 7, 7) and w.is_contiguous()
    ph, pw = (sh - 6 + 2) // 3, (sw - 6 + 2) // 3
    y = torch.empty((9 * b, ph, pw, 32), device=taps4.device, dtype=torch.float32)
    bits = torch.empty((9 * b, ph, pw), device=taps4.device, dtype=torch.int32)
    check(_lib.lib().dd_conv1ch_fwd_phase3(_p(taps4), _p(w), _p(bias), _p(y), _p(bits), b, sh, sw, int(relu), _stream()), "dd_conv1ch_fwd_phase3")
    return y, bits


def conv1ch_wgrad_phase3(taps4, g_phase):
    """conv1ch_wgrad from dL/dy in the phase-major layout [9B, ph, pw, 32]."""
    b, sh, sw, _ = taps4.shape
    ph, pw = (sh - 6 + 2) // 3, (sw - 6 + 2) // 3
    assert tuple(g_phase.shape) == (9 * b, ph, pw, 32) and g_phase.is_contiguous()
    dw = torch.empty((32, 1, 7, 7), device=g_phase.device, dtype=torch.float32)
    db = torch.empty(32, device=g_phase.device, dtype=torch.float32)
    ws = torch.empty(_lib.lib().dd_conv1ch_wgrad_workspace_bytes(), device=g_phase.device, dtype=torch.uint8)
    check(_lib.lib().dd_conv1ch_wgrad_phase3(_p(taps4), _p(g_phase), _p(dw), _p(db), b, sh, sw, _p(ws), _stream()), "dd_conv1ch_wgrad_phase3")
    return dw, db


def phase3_scatter(src_phase, dst, coff, off):
    """dst [B,oh,ow,cs] channels [coff, +32) <- the phase images src_phase [9B, ph, pw, 32], read ``off`` cells in from their corner."""
    _dev(src_phase, "src_phase")
    _dev(dst, "dst")
    b, oh, ow, cs = dst.shape
    assert src_phase.shape[0] == 9 * b and src_phase.shape[3] == 32
    check(_lib.lib().dd_phase3_scatter(_p(src_phase), _p(dst), b, oh, ow, src_phase.shape[1], src_phase.shape[2], off, cs, coff, _stream()),
          "dd_phase3_scatter")


def phase3_gather(src, coff, ph, pw, off):
    """-> [9B, ph, pw, 32]: the 32-channel slice [coff, +32) of the dense NHWC buffer src in phase images, ``off`` cells in; zero elsewhere."""
    _dev(src, "src")
    b, oh, ow, cs = src.shape
    out = torch.empty((9 * b, ph, pw, 32), device=src.device, dtype=torch.float32)
    check(_lib.lib().dd_phase3_gather(_p(src), _p(out), b, oh, ow, ph, pw, off, cs, coff, _stream()), "dd_phase3_gather")
    return out


def nhwc_to_nchw(x, c):
    b, h, w, cs = x.shape
    _dev(x, "x")
    out = torch.empty((b, c, h, w), device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_nhwc_to_nchw(_p(x), _p(out), b, c, h, w, cs, _stream()), "dd_nhwc_to_nchw")
    return out


class ToNHWC(torch.autograd.Function):
    """Differentiable NCHW -> NHWC(c_store) re-layout (API edges: callers may hand in ordinary NCHW tensors)."""

    @staticmethod
    def forward(ctx, x, c_store):
        ctx.c = x.shape[1]
        return nchw_to_nhwc(x.contiguous(), c_store)

    @staticmethod
    def backward(ctx, g):
        return nhwc_to_nchw(g.contiguous(), ctx.c), None


# ------------------------------------------------------------------------------------------------ conv primitives
def conv_pack(weight, desc, kind):
    _dev(weight, "weight", (32, desc.cin_real, 3, 3))
    n = _lib.lib().dd_conv_packed_floats(C.byref(desc), kind)
    if n <= 0:
        raise _lib.HotpathError(f"conv_pack: {_lib.lib().dd_last_error().decode()}")
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_pack(_p(weight), _p(packed), C.byref(desc), kind, _stream()), "dd_conv_pack")
    return packed


def conv_fwd(x, packed, bias, desc, epilogue=EPI_BIAS_RELU, mask=None):
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _dev(x, "x", (desc.batch, desc.height, desc.width, desc.cin_store))
    if bias is not None:
        _dev(bias, "bias", (32,))
    if mask is not None:
        _dev(mask, "mask", (desc.batch, ho, wo, 32))
    y = torch.empty((desc.batch, ho, wo, 32), device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_fwd(_p(x), _p(packed), _p(bias), _p(mask), _p(y), C.byref(desc), epilogue, _stream()), "dd_conv_fwd")
    return y


def conv_fwd_bits(x, packed, bias, desc):
    """relu(conv(x) + bias) plus the ReLU signs as one uint32 per pixel (bit c = channel c is positive)."""
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _dev(x, "x", (desc.batch, desc.height, desc.width, desc.cin_store))
    _dev(bias, "bias", (32,))
    y = torch.empty((desc.batch, ho, wo, 32), device=x.device, dtype=torch.float32)
    bits = torch.empty((desc.batch, ho, wo), device=x.device, dtype=torch.int32)
    check(_lib.lib().dd_conv_fwd_relu_bits(_p(x), _p(packed), _p(bias), _p(y), _p(bits), C.byref(desc), _stream()), "dd_conv_fwd_relu_bits")
    return y, bits


# Winograd F(2,3) along x for the 32 -> 32 stride-1 layer: 2/3 of the matrix-core work of the direct form, exact fp32
# arithmetic.  On by default for c2's forward and data gradient inside EncoderConvStack; the direct kernels stay
# available (set ops.WINOGRAD = False) and are what the kernel-level entry points above call.
WINOGRAD = True
WINOGRAD_2D = True      # forward / data gradient of c2 by F(2x2,3x3) instead of F(2,3) along x (16 instead of 24 multiplies per tile)


def conv_wino_pack(weight, desc, kind):
    _dev(weight, "weight", (32, 32, 3, 3))
    n = _lib.lib().dd_conv_wino_packed_floats(C.byref(desc))
    if n <= 0:
        raise _lib.HotpathError(f"conv_wino_pack: {_lib.lib().dd_last_error().decode()}")
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wino_pack(_p(weight), _p(packed), C.byref(desc), kind, _stream()), "dd_conv_wino_pack")
    return packed


def conv_wino_fwd_bits(x, packed, bias, desc):
    _dev(x, "x", (desc.batch, desc.height, desc.width, 32))
    _dev(bias, "bias", (32,))
    y = torch.empty((desc.batch, desc.height, desc.width, 32), device=x.device, dtype=torch.float32)
    bits = torch.empty((desc.batch, desc.height, desc.width), device=x.device, dtype=torch.int32)
    check(_lib.lib().dd_conv_wino_fwd_relu_bits(_p(x), _p(packed), _p(bias), _p(y), _p(bits), C.byref(desc), _stream()),
          "dd_conv_wino_fwd_relu_bits")
    return y, bits


def conv_wino_dgrad_bits(dy, packed, bits, desc):
    _dev(dy, "dy", (desc.batch, desc.height, desc.width, 32))
    if not (bits.is_cuda and bits.dtype == torch.int32 and bits.is_contiguous() and tuple(bits.shape) == (desc.batch, desc.height, desc.width)):
        raise _lib.HotpathError("conv_wino_dgrad_bits: relu_bits must be a contiguous int32 [B,H,W] device tensor")
    dx = torch.empty((desc.batch, desc.height, desc.width, 32), device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wino_dgrad_relu_bits(_p(dy), _p(packed), _p(bits), _p(dx), C.byref(desc), _stream()),
          "dd_conv_wino_dgrad_relu_bits")
    return dx


def conv_wino2_pack(weight, desc, kind):
    _dev(weight, "weight", (32, 32, 3, 3))
    n = _lib.lib().dd_conv_wino2_packed_floats(C.byref(desc))
    if n <= 0:
        raise _lib.HotpathError(f"conv_wino2_pack: {_lib.lib().dd_last_error().decode()}")
    packed = torch.empty(n, device=weight.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wino2_pack(_p(weight), _p(packed), C.byref(desc), kind, _stream()), "dd_conv_wino2_pack")
    return packed


def conv_wino2_fwd_bits(x, packed, bias, desc):
    _dev(x, "x", (desc.batch, desc.height, desc.width, 32))
    _dev(bias, "bias", (32,))
    y = torch.empty((desc.batch, desc.height, desc.width, 32), device=x.device, dtype=torch.float32)
    bits = torch.empty((desc.batch, desc.height, desc.width), device=x.device, dtype=torch.int32)
    check(_lib.lib().dd_conv_wino2_fwd_relu_bits(_p(x), _p(packed), _p(bias), _p(y), _p(bits), C.byref(desc), _stream()),
          "dd_conv_wino2_fwd_relu_bits")
    return y, bits


def conv_wino2_dgrad_bits(dy, packed, bits, desc):
    _dev(dy, "dy", (desc.batch, desc.height, desc.width, 32))
    if not (bits.is_cuda and bits.dtype == torch.int32 and bits.is_contiguous() and tuple(bits.shape) == (desc.batch, desc.height, desc.width)):
        raise _lib.HotpathError("conv_wino2_dgrad_bits: relu_bits must be a contiguous int32 [B,H,W] device tensor")
    dx = torch.empty((desc.batch, desc.height, desc.width, 32), device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wino2_dgrad_relu_bits(_p(dy), _p(packed), _p(bits), _p(dx), C.byref(desc), _stream()),
          "dd_conv_wino2_dgrad_relu_bits")
    return dx


def conv_wino2_dgrad_w1(dy, packed, bits, x4, desc):
    """c2's data gradient consumed in place: returns the 3 -> 32 layer's (dW1 [32,3,3,3], db1 [32]) instead of g1
    (dd_conv_wino2_dgrad_w1: g1 is never written to or re-read from HBM)."""
    _dev(dy, "dy", (desc.batch, desc.height, desc.width, 32))
    _dev(x4, "x4", (desc.batch, desc.height, desc.width, 4))
    if not (bits.is_cuda and bits.dtype == torch.int32 and bits.is_contiguous() and tuple(bits.shape) == (desc.batch, desc.height, desc.width)):
        raise _lib.HotpathError("conv_wino2_dgrad_w1: relu_bits must be a contiguous int32 [B,H,W] device tensor")
    n = _lib.lib().dd_conv_wino2_dgrad_w1_workspace_bytes(C.byref(desc))
    ws = torch.empty(n, device=dy.device, dtype=torch.uint8)
    dw = torch.empty((32, 3, 3, 3), device=dy.device, dtype=torch.float32)
    db = torch.empty((32,), device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wino2_dgrad_w1(_p(dy), _p(packed), _p(bits), _p(x4), _p(dw), _p(db), _p(ws), n, C.byref(desc),
                                            _stream()), "dd_conv_wino2_dgrad_w1")
    return dw, db


def conv_wino2_wgrad(x, dy, desc, finish_stream=None):
    _dev(x, "x", (desc.batch, desc.height, desc.width, 32))
    _dev(dy, "dy", (desc.batch, desc.height, desc.width, 32))
    nbytes = _lib.lib().dd_conv_wino2_wgrad_workspace_bytes(C.byref(desc))
    if nbytes <= 0:
        raise _lib.HotpathError(f"conv_wino2_wgrad: {_lib.lib().dd_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    dw = torch.empty((32, 32, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty(32, device=x.device, dtype=torch.float32)
    if finish_stream is None:
        check(_lib.lib().dd_conv_wino2_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes, C.byref(desc), _stream()), "dd_conv_wino2_wgrad")
        return dw, db
    # the two reduce kernels go to `finish_stream` (beside whatever the caller launches next); the caller's stream has to
    # wait for the returned event before it reads dw / db
    main = torch.cuda.current_stream()
    check(_lib.lib().dd_conv_wino2_wgrad_partials(_p(x), _p(dy), _p(ws), nbytes, C.byref(desc), _stream()), "dd_conv_wino2_wgrad_partials")
    finish_stream.wait_event(main.record_event())
    with torch.cuda.stream(finish_stream):
        for t in (ws, dw, db):
            t.record_stream(finish_stream)
        check(_lib.lib().dd_conv_wino2_wgrad_finish(_p(ws), nbytes, _p(dw), _p(db), C.byref(desc), _stream()), "dd_conv_wino2_wgrad_finish")
        done = finish_stream.record_event()
    return dw, db, done


def conv_wino_wgrad(x, dy, desc):
    _dev(x, "x", (desc.batch, desc.height, desc.width, 32))
    _dev(dy, "dy", (desc.batch, desc.height, desc.width, 32))
    nbytes = _lib.lib().dd_conv_wino_wgrad_workspace_bytes(C.byref(desc))
    if nbytes <= 0:
        raise _lib.HotpathError(f"conv_wino_wgrad: {_lib.lib().dd_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    dw = torch.empty((32, 32, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty(32, device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wino_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes, C.byref(desc), _stream()), "dd_conv_wino_wgrad")
    return dw, db


def conv_dgrad_bits(dy, packed_dgrad, bits, desc):
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _dev(dy, "dy", (desc.batch, ho, wo, 32))
    if not (bits.is_cuda and bits.dtype == torch.int32 and bits.is_contiguous() and tuple(bits.shape) == (desc.batch, desc.height, desc.width)):
        raise _lib.HotpathError("conv_dgrad_bits: relu_bits must be a contiguous int32 [B,H,W] device tensor")
    dx = torch.empty((desc.batch, desc.height, desc.width, 32), device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_dgrad_relu_bits(_p(dy), _p(packed_dgrad), _p(bits), _p(dx), C.byref(desc), _stream()), "dd_conv_dgrad_relu_bits")
    return dx


def conv_dgrad(dy, packed_dgrad, relu_src, desc):
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _dev(dy, "dy", (desc.batch, ho, wo, 32))
    if relu_src is not None:
        _dev(relu_src, "relu_src", (desc.batch, desc.height, desc.width, 32))
    dx = torch.empty((desc.batch, desc.height, desc.width, 32), device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_dgrad(_p(dy), _p(packed_dgrad), _p(relu_src), _p(dx), C.byref(desc), _stream()), "dd_conv_dgrad")
    return dx


def conv_wgrad(x, dy, desc):
    ho, wo = conv_out(desc.height, desc.stride), conv_out(desc.width, desc.stride)
    _dev(x, "x", (desc.batch, desc.height, desc.width, desc.cin_store))
    _dev(dy, "dy", (desc.batch, ho, wo, 32))
    nbytes = _lib.lib().dd_conv_wgrad_workspace_bytes(C.byref(desc))
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    dw = torch.empty((32, desc.cin_real, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.empty(32, device=x.device, dtype=torch.float32)
    check(_lib.lib().dd_conv_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes, C.byref(desc), _stream()), "dd_conv_wgrad")
    return dw, db


def add(a, b):
    _dev(a, "a")
    _dev(b, "b", a.shape)
    out = torch.empty_like(a)
    check(_lib.lib().dd_add(_p(a), _p(b), _p(out), a.numel(), _stream()), "dd_add")
    return out


def relu_bwd(dy, y):
    _dev(dy, "dy", y.shape)
    _dev(y, "y")
    out = torch.empty_like(y)
    check(_lib.lib().dd_relu_bwd(_p(dy), _p(y), _p(out), y.numel(), _stream()), "dd_relu_bwd")
    return out


def relu_sign_bits(x):
    """int32 [B,H,W]: bit c = x[b,y,x,c] > 0 of an NHWC activation with 32 channels (dd_relu_sign_bits)."""
    _dev(x, "x")
    if x.dim() != 4 or x.shape[3] != 32:
        raise _lib.HotpathError(f"relu_sign_bits: expected [B,H,W,32], got {tuple(x.shape)}")
    bits = torch.empty(x.shape[:3], device=x.device, dtype=torch.int32)
    check(_lib.lib().dd_relu_sign_bits(_p(x), _p(bits), bits.numel(), _stream()), "dd_relu_sign_bits")
    return bits


def channel_slice(v):
    """``v`` [B,H,W,C]: (dense NHWC buffer [B,H,W,cs] it is a channel slice of, channel offset), or None (dense tensors: (v, 0))."""
    if v.dim() != 4:
        return None
    if v.is_contiguous():
        return v, 0
    b, h, w, c = v.shape
    sb, sh, sw, sc = v.stride()
    if sc != 1 or sw < c or sh != w * sw or (b > 1 and sb != h * sh):
        return None
    coff = v.storage_offset() % sw
    if coff + c > sw or v.storage_offset() != coff or v.untyped_storage().nbytes() // v.element_size() < b * h * w * sw:
        return None
    return v.as_strided((b, h, w, sw), (h * w * sw, w * sw, sw, 1), 0), coff


def relu_bwd_pad_bits(dy, bits_pad):
    """dy [B,H,W,32] (dense, or a channel slice of a dense NHWC buffer: read where it lies), sign words [B,H+2,W+2] ->
    [B,H+2,W+2,32]: dy behind the ReLU in the interior, zero on the border ring."""
    src = channel_slice(dy)
    if src is None:
        src = (dy.contiguous(), 0)
    buf, coff = src
    _dev(buf, "dy")
    b, h, w, c = dy.shape
    if c != 32 or coff % 4 or buf.shape[3] % 4 or not (bits_pad.is_cuda and bits_pad.dtype == torch.int32 and bits_pad.is_contiguous() and
                                                        tuple(bits_pad.shape) == (b, h + 2, w + 2)):
        raise _lib.HotpathError(f"relu_bwd_pad_bits: dy {tuple(dy.shape)} needs 32 channels (a 4-aligned slice) and contiguous int32 sign words [B,H+2,W+2], got {tuple(bits_pad.shape)}")
    out = torch.empty((b, h + 2, w + 2, 32), device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_relu_bwd_pad_bits(_p(buf), _p(bits_pad), _p(out), b, h, w, buf.shape[3], coff, _stream()), "dd_relu_bwd_pad_bits")
    return out


def pool4_fwd(feat):
    b, h, w, c = feat.shape
    _dev(feat, "feat")
    out = torch.empty((b, (c * h * w) // 4), device=feat.device, dtype=torch.float32)
    check(_lib.lib().dd_pool4_fwd(_p(feat), _p(out), b, h, w, c, _stream()), "dd_pool4_fwd")
    return out


def pool4_has_idx(h, w, c):
    """The routing-code form needs windows that stay inside one channel plane."""
    return (h * w) % 4 == 0 and c % 4 == 0


def pool4_fwd_idx(feat):
    """pooled, codes: max_pool1d(4) + the backward's routing (dd_pool4_fwd_idx)."""
    b, h, w, c = feat.shape
    _dev(feat, "feat")
    n = _lib.lib().dd_pool4_idx_elems(b, h, w, c)
    if n < 0:
        raise _lib.HotpathError(_lib.lib().dd_last_error().decode())
    out = torch.empty((b, (c * h * w) // 4), device=feat.device, dtype=torch.float32)
    idx = torch.empty((n,), device=feat.device, dtype=torch.int16)
    check(_lib.lib().dd_pool4_fwd_idx(_p(feat), _p(out), _p(idx), b, h, w, c, _stream()), "dd_pool4_fwd_idx")
    return out, idx


def pool4_idx_relu_bwd(dpooled, idx, shape):
    b, h, w, c = shape
    _dev(dpooled, "dpooled", (b, (c * h * w) // 4))
    if idx.dtype != torch.int16 or not idx.is_cuda or idx.numel() != b * (h * w // 4) * (c // 4):
        raise _lib.HotpathError(f"pool4_idx_relu_bwd: bad routing codes {tuple(idx.shape)} {idx.dtype}")
    out = torch.empty(shape, device=dpooled.device, dtype=torch.float32)
    check(_lib.lib().dd_pool4_idx_relu_bwd(_p(dpooled), _p(idx), _p(out), b, h, w, c, _stream()), "dd_pool4_idx_relu_bwd")
    return out


def pool4_relu_bwd_add(dpooled, feat, gfeat):
    """(feat > 0) * (gfeat + routed dpooled): the c3 feature's gradient when both the pool and the box heads consume it, one pass."""
    b, h, w, c = feat.shape
    _dev(dpooled, "dpooled", (b, (c * h * w) // 4))
    _dev(feat, "feat")
    _dev(gfeat, "gfeat", feat.shape)
    out = torch.empty_like(feat)
    check(_lib.lib().dd_pool4_relu_bwd_add(_p(dpooled), _p(feat), _p(gfeat), _p(out), b, h, w, c, _stream()), "dd_pool4_relu_bwd_add")
    return out


def pool4_relu_bwd(dpooled, feat):
    b, h, w, c = feat.shape
    _dev(dpooled, "dpooled", (b, (c * h * w) // 4))
    _dev(feat, "feat")
    out = torch.empty_like(feat)
    check(_lib.lib().dd_pool4_relu_bwd(_p(dpooled), _p(feat), _p(out), b, h, w, c, _stream()), "dd_pool4_relu_bwd")
    return out


# Callbacks fired when backward enters its long MFMA-bound stretch (the c2 weight / data gradient kernels, ~4.4 ms
# at bs = 32 that use a third of the HBM bandwidth): the place to start bandwidth-bound side work such as the
# optimizer pass of already-finished gradients (optim.HipAdam.overlap_with_backward).
MFMA_PHASE_HOOKS = []
# ... and fired again between c2's weight gradient and its data gradient, ~1.2 ms later: the place for work that has to WAIT for
# something started at the top of the backward (factor mode of ddp.GradSync: the gathered factors of the two big Linear layers).
MFMA_PHASE2_HOOKS = []
C2_DGRAD_FIRST = False      # c2's data gradient before its weight gradient (same results; see EncoderConvStack.backward)

# c1's weight gradient taken from c2's data gradient inside conv_wino2_fwd<EPI_RELU_BITS_W1> (2-D Winograd path only)
FUSE_C1_WGRAD = True      # c1's weight gradient inside c2's data gradient (the 2-D Winograd form); tests may switch it off in process


# Test hook: when set to a dict, EncoderConvStack.forward leaves its three ReLU outputs (NHWC) in it, so that a checker can
# replay the product's ReLU / max-pool decisions (oracle.branch); None in production.
TRACE = None

_PACK_STREAMS = {}


def _pack_stream(device):
    """Side stream for the weight-image packs of EncoderConvStack (one per device)."""
    key = torch.device(device).index
    if key not in _PACK_STREAMS:
        _PACK_STREAMS[key] = torch.cuda.Stream(device=device)
    return _PACK_STREAMS[key]


# ------------------------------------------------------------------------------------------------ encoder conv stack
class EncoderConvStack(torch.autograd.Function):
    """c1 -> ReLU -> c2 -> ReLU -> c3 (stride 2) -> ReLU [-> NCHW-order max_pool1d(4)] as one autograd node.

    Reference: Encoder.forward, src/autoencoder/components.py:41-47.  Running the three layers as
    one node lets the backward fuse each ReLU's gradient into the neighbouring kernel (the pool
    backward masks with c3's output, each data-gradient kernel masks with its layer's input) and
    keeps every intermediate in NHWC.

    forward(x4 [B,H,W,4], w1,b1,w2,b2,w3,b3, pool) -> feat [B,Ho,Wo,32] (NHWC)   or  pooled [B, 32*Ho*Wo/4]
    """

    @staticmethod
    def forward(ctx, x4, w1, b1, w2, b2, w3, b3, pool, rows_per_task):
        b, h, w, _ = x4.shape
        d1 = conv_desc(b, h, w, 3, 1, rows_per_task)
        d2 = conv_desc(b, h, w, 32, 1, rows_per_task)
        d3 = conv_desc(b, h, w, 32, 2, rows_per_task)
        # Operand images.  Only c1's is needed at once; the other four (c2 / c3 forward, and the backward's two: the weights do
        # not change before the backward, where these tiny kernels would sit on the critical path behind the optimizer pass
        # that overlaps it -- 4 us alone, up to 390 us squeezed between Adam's workgroups) are packed on a side stream
        # while c1's forward runs: five 5-us launches less on the critical path.
        need = ctx.needs_input_grad               # (x4, w1, b1, w2, b2, w3, b3, ...)
        p1 = conv_pack(w1, d1, PACK_FWD)
        main = torch.cuda.current_stream()
        side = _pack_stream(x4.device)
        side.wait_event(main.record_event())
        with torch.cuda.stream(side):
            if WINOGRAD and WINOGRAD_2D:
                p2 = conv_wino2_pack(w2, d2, 0)
            elif WINOGRAD:
                p2 = conv_wino_pack(w2, d2, 0)
            else:
                p2 = conv_pack(w2, d2, PACK_FWD)
            p3 = conv_pack(w3, d3, PACK_FWD)
            p2d = p3d = torch.empty(0, device=x4.device)
            if need[1] or need[2] or need[3] or need[4]:
                p3d = conv_pack(w3, d3, PACK_DGRAD_S2)
            if need[1] or need[2]:
                if WINOGRAD and WINOGRAD_2D:
                    p2d = conv_wino2_pack(w2, d2, 1)
                elif WINOGRAD:
                    p2d = conv_wino_pack(w2, d2, 1)
                else:
                    p2d = conv_pack(w2, d2, PACK_DGRAD_S1)
            for t in (p2, p3, p2d, p3d):
                t.record_stream(main)             # allocated under the side stream, consumed on the main one
            packed_ready = side.record_event()
        # c1 / c2 also emit their ReLU signs as bit planes (60 MB instead of 1.9 GB to re-read in the backward)
        a1, s1 = conv_fwd_bits(x4, p1, b1, d1)
        main.wait_event(packed_ready)
        if WINOGRAD and WINOGRAD_2D:
            a2, s2 = conv_wino2_fwd_bits(a1, p2, b2, d2)
        elif WINOGRAD:
            a2, s2 = conv_wino_fwd_bits(a1, p2, b2, d2)
        else:
            a2, s2 = conv_fwd_bits(a1, p2, b2, d2)
        a3 = conv_fwd(a2, p3, b3, d3)
        if TRACE is not None:
            TRACE.update(a1=a1, a2=a2, a3=a3)
        ctx.wino = (bool(WINOGRAD), bool(WINOGRAD and WINOGRAD_2D))
        ctx.pool = int(pool)                  # 0: conv feature, 1: pooled vector, 2: both (joint roadmap + box model)
        ctx.rows_per_task = rows_per_task
        ctx.a3_shape = tuple(a3.shape)
        if ctx.pool == 1 and pool4_has_idx(*a3.shape[1:]):
            # the pool decides the backward's routing now (4 bits per window): the feature itself is not kept
            pooled, codes = pool4_fwd_idx(a3)
            ctx.save_for_backward(x4, a1, a2, codes, p2d, p3d, s1, s2)
            ctx.codes = True
            return pooled
        ctx.codes = False
        ctx.save_for_backward(x4, a1, a2, a3, p2d, p3d, s1, s2)
        if ctx.pool == 2:
            return a3, pool4_fwd(a3)
        if ctx.pool:
            return pool4_fwd(a3)
        return a3

    @staticmethod
    def backward(ctx, grad, grad_pooled=None):
        x4, a1, a2, a3, p2d, p3d, s1, s2 = ctx.saved_tensors
        wino, wino2 = ctx.wino
        b, h, w, _ = x4.shape
        rpt = ctx.rows_per_task
        d1, d2, d3 = conv_desc(b, h, w, 3, 1, rpt), conv_desc(b, h, w, 32, 1, rpt), conv_desc(b, h, w, 32, 2, rpt)
        if ctx.pool == 2:                     # two consumers of the c3 feature: their gradients add
            if grad is not None and grad_pooled is not None and a3.shape[3] == 32 and (a3.shape[1] * a3.shape[2]) % 4 == 0:
                g3 = pool4_relu_bwd_add(grad_pooled.contiguous(), a3, grad.contiguous())      # one pass instead of three over the 481 MB feature
            else:
                parts = []
                if grad is not None:
                    parts.append(relu_bwd(grad.contiguous(), a3))
                if grad_pooled is not None:
                    parts.append(pool4_relu_bwd(grad_pooled.contiguous(), a3))
                g3 = parts[0] if len(parts) == 1 else add(parts[0], parts[1])
        else:
            grad = grad.contiguous()
            if ctx.codes:
                g3 = pool4_idx_relu_bwd(grad, a3, ctx.a3_shape)      # a3 holds the routing codes here
            else:
                g3 = pool4_relu_bwd(grad, a3) if ctx.pool else relu_bwd(grad, a3)
        need = ctx.needs_input_grad
        dw3, db3 = conv_wgrad(a2, g3, d3) if (need[5] or need[6]) else (None, None)
        dw2 = db2 = dw1 = db1 = None
        if need[1] or need[2] or need[3] or need[4]:
            g2 = conv_dgrad_bits(g3, p3d, s2, d3)
            del g3
            for hook in MFMA_PHASE_HOOKS:
                hook()
            reduced = None

            def c2_weight_gradient():
                nonlocal dw2, db2, reduced
                if need[3] or need[4]:
                    if wino2:      # the reduce of the partials runs on the side stream, beside c2's data gradient
                        dw2, db2, reduced = conv_wino2_wgrad(a1, g2, d2, finish_stream=_pack_stream(g2.device))
                    elif wino:
                        dw2, db2 = conv_wino_wgrad(a1, g2, d2)
                    else:
                        dw2, db2 = conv_wgrad(a1, g2, d2)

            def c2_data_gradient():
                nonlocal dw1, db1
                if (need[1] or need[2]) and wino2 and FUSE_C1_WGRAD:
                    dw1, db1 = conv_wino2_dgrad_w1(g2, p2d, s1, x4, d2)      # g1 never leaves the registers
                elif need[1] or need[2]:
                    if wino2:
                        g1 = conv_wino2_dgrad_bits(g2, p2d, s1, d2)
                    elif wino:
                        g1 = conv_wino_dgrad_bits(g2, p2d, s1, d2)
                    else:
                        g1 = conv_dgrad_bits(g2, p2d, s1, d2)
                    dw1, db1 = conv_wgrad(x4, g1, d1)

            # C2_DGRAD_FIRST (optim.HipAdam, factor mode of ddp.GradSync): the optimizer passes of the two big Linear layers can only
            # start once their gathered factors have arrived, ~3 ms into the backward, and they can only run BESIDE c2's weight gradient
            # (440 registers per SIMD: one 48-register Adam wave fits; the data-gradient kernel's 475 leave no room) -- so that kernel
            # goes last and the second hook sits in front of it
            if C2_DGRAD_FIRST:
                c2_data_gradient()
                for hook in MFMA_PHASE2_HOOKS:
                    hook()
                c2_weight_gradient()
            else:
                c2_weight_gradient()
                for hook in MFMA_PHASE2_HOOKS:
                    hook()
                c2_data_gradient()
            del g2
            if reduced is not None:
                torch.cuda.current_stream().wait_event(reduced)
        return None, dw1, db1, dw2, db2, dw3, db3, None, None


def encoder_conv_stack(x4, c1, c2, c3, pool, rows_per_task=0):
    return EncoderConvStack.apply(x4, c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias, pool, rows_per_task)


# ------------------------------------------------------------------------------------------------ skinny GEMMs
def _linear_ws(m, n, k, device):
    nbytes = _lib.lib().dd_linear_workspace_bytes(m, n, k)
    return torch.empty(nbytes, device=device, dtype=torch.uint8), nbytes


# data_ptr of an nn.Linear weight -> optim.HipAdam in rank-B mode: Linear.backward hands (x, dy) over instead of forming dW
RANKB = {}


class Linear(torch.autograd.Function):
    """y = x W^T + b with nn.Linear's [out, in] weight, all three passes on the fp32 matrix cores
    (dd_linear_fwd / dgrad / wgrad).  Reference call sites: components.py:105 (DenseBlock.fc1),
    components.py:51 (fc_z_out), roadmap_bce_v2.py:75 (head)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        m, k = x.shape
        n = weight.shape[0]
        _dev(x, "x")
        _dev(weight, "weight", (n, k))
        if bias is not None:
            _dev(bias, "bias", (n,))
        y = torch.empty((m, n), device=x.device, dtype=torch.float32)
        ws, nbytes = _linear_ws(m, n, k, x.device)
        check(_lib.lib().dd_linear_fwd(_p(x), _p(weight), _p(bias), _p(y), m, n, k, _p(ws), nbytes, _stream()), "dd_linear_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        m, k = x.shape
        n = weight.shape[0]
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ws, nbytes = _linear_ws(m, n, k, x.device)
            check(_lib.lib().dd_linear_dgrad(_p(dy), _p(weight), _p(dx), m, n, k, _p(ws), nbytes, _stream()), "dd_linear_dgrad")
        sync = _ddp.FACTOR_SYNC.get(weight.data_ptr()) if _ddp.FACTOR_SYNC else None
        fused = RANKB.get(weight.data_ptr()) if RANKB else None
        taken = 0
        if sync is not None and ctx.needs_input_grad[1] and sync.linear_factors(weight, x, dy):
            # data parallel, factor mode (ddp.GradSync): x and dy travel instead of dW; the optimizer forms the global-batch gradient
            # (rank-B mode of the optimizer: inside its Adam pass, the bias from the gathered dy as well)
            if ctx.has_bias and ctx.needs_input_grad[2] and not (fused is not None and fused.factor_bias(weight, sync.world * m)):
                db = column_sum(dy)
        elif fused is not None and ctx.needs_input_grad[1] and (taken := fused.linear_factors(weight, x, dy)):
            # rank-B mode (optim.HipAdam): no dW at all, dd_adam_step_rankb forms it from (x, dy) inside the optimizer pass
            if taken == 1 and ctx.has_bias and ctx.needs_input_grad[2]:
                db = column_sum(dy)
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight)
            db = torch.empty(n, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            check(_lib.lib().dd_linear_wgrad(_p(dy), _p(x), _p(dw), _p(db), m, n, k, _stream()), "dd_linear_wgrad")
        return dx, dw, db


def linear(x, weight, bias):
    x = x.contiguous()
    # factor mode of ddp.GradSync: a big input (fc1's 120 MB of pooled activations) starts its all-gather now, not in the backward.
    # Decided HERE: inside Function.forward grad mode is always off and ctx.needs_input_grad ignores torch.no_grad()
    if _ddp.FACTOR_SYNC and torch.is_grad_enabled() and weight.requires_grad:
        sync = _ddp.FACTOR_SYNC.get(weight.data_ptr())
        if sync is not None:
            sync.linear_input(weight, x)
    return Linear.apply(x, weight, bias)


def column_sum(dy):
    """db [n] = sum over the rows of dy [m, n] on the hot path's own kernel (dd_column_sum): the bias gradient of a Linear layer whose
    weight gradient is not formed by dd_linear_wgrad."""
    m, n = dy.shape
    _dev(dy, "dy")
    db = torch.empty(n, device=dy.device, dtype=torch.float32)
    check(_lib.lib().dd_column_sum(_p(dy), _p(db), m, n, _stream()), "dd_column_sum")
    return db


def linear_wgrad(dy, x, dw):
    """dw [n, k] = dy^T x for dy [m, n], x [m, k] (dd_linear_wgrad without the bias sum): the optimizer's global-batch weight gradient
    from gathered factors (ddp.GradSync, factor mode)."""
    m, n = dy.shape
    k = x.shape[1]
    _dev(dy, "dy")
    _dev(x, "x", (m, k))
    _dev(dw, "dw", (n, k))
    check(_lib.lib().dd_linear_wgrad(_p(dy), _p(x), _p(dw), None, m, n, k, _stream()), "dd_linear_wgrad")


# ------------------------------------------------------------------------------------------------ dense block tail
class BnReluDrop(torch.autograd.Function):
    """BatchNorm1d -> ReLU -> dropout(keep mask) in one kernel each way (components.py:105-108)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, keep, training, eps, momentum, scale, num_batches_tracked=None):
        rows, feat = x.shape
        if num_batches_tracked is not None and not (num_batches_tracked.is_cuda and num_batches_tracked.dtype == torch.int64):
            raise _lib.HotpathError("bn_relu_drop: num_batches_tracked must be an int64 device tensor")
        _dev(x, "x")
        y = torch.empty_like(x)
        save_mean = torch.empty(feat, device=x.device, dtype=torch.float32)
        save_inv = torch.empty(feat, device=x.device, dtype=torch.float32)
        check(_lib.lib().dd_bn_relu_drop_fwd(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(keep),
                                             _p(y), _p(save_mean), _p(save_inv), rows, feat, eps, momentum, scale,
                                             int(training), _p(num_batches_tracked), _stream()), "dd_bn_relu_drop_fwd")
        if TRACE is not None:
            TRACE.setdefault("dense", []).append(y)
        ctx.save_for_backward(x, y, gamma, keep, save_mean, save_inv, running_mean, running_var)
        ctx.cfg = (bool(training), eps, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, keep, save_mean, save_inv, running_mean, running_var = ctx.saved_tensors
        training, eps, scale = ctx.cfg
        rows, feat = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        check(_lib.lib().dd_bn_relu_drop_bwd(_p(dy), _p(x), _p(y), _p(gamma), _p(keep), _p(save_mean), _p(save_inv),
                                             _p(running_mean), _p(running_var), _p(dx), _p(dgamma), _p(dbeta), rows, feat,
                                             eps, scale, int(training), _stream()), "dd_bn_relu_drop_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ encoder tail
FUSE_MLP_TAIL = True      # the encoder tail as one launch each way where its sizes allow (dd_mlp_tail_supported)


def mlp_tail_supported(m, h1, h2, l):
    return FUSE_MLP_TAIL and bool(_lib.lib().dd_mlp_tail_supported(m, h1, h2, l))


class EncoderTail(torch.autograd.Function):
    """BatchNorm1d -> ReLU -> dropout -> Linear -> BatchNorm1d -> ReLU -> dropout -> Linear as one launch each way
    (dd_mlp_tail_fwd / dd_mlp_tail_bwd; reference components.py:48-51 + DenseBlock.forward :104-109).

    forward(lin1, gamma1, beta1, w2, bias2, gamma2, beta2, wz, bz, keep1, keep2, bn1, bn2, scale1, scale2) -> z
    ``bn1`` / ``bn2`` are the BatchNorm1d modules (running statistics, eps, momentum, mode)."""

    @staticmethod
    def forward(ctx, lin1, gamma1, beta1, w2, bias2, gamma2, beta2, wz, bz, keep1, keep2, bn1, bn2, scale1, scale2):
        m, h1 = lin1.shape
        h2, l = w2.shape[0], wz.shape[0]
        for name, t, shape in (("lin1", lin1, None), ("w2", w2, (h2, h1)), ("wz", wz, (l, h2)), ("bias2", bias2, (h2,)), ("bz", bz, (l,)),
                               ("gamma1", gamma1, (h1,)), ("beta1", beta1, (h1,)), ("gamma2", gamma2, (h2,)), ("beta2", beta2, (h2,))):
            _dev(t, name, shape)
        if keep1 is not None:
            _dev(keep1, "keep1", (m, h1))
        if keep2 is not None:
            _dev(keep2, "keep2", (m, h2))
        training = bool(bn1.training)
        dev = lin1.device
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        y1, lin2, y2, z = new(m, h1), new(m, h2), new(m, h2), new(m, l)
        mean1, inv1, mean2, inv2 = new(h1), new(h1), new(h2), new(h2)
        mom = lambda bn: 0.1 if bn.momentum is None else bn.momentum
        nbt = lambda bn: bn.num_batches_tracked if (training and bn.num_batches_tracked is not None) else None
        check(_lib.lib().dd_mlp_tail_fwd(_p(lin1), _p(gamma1), _p(beta1), _p(bn1.running_mean), _p(bn1.running_var), _p(nbt(bn1)),
                                         _p(keep1), _p(w2), _p(bias2), _p(gamma2), _p(beta2), _p(bn2.running_mean),
                                         _p(bn2.running_var), _p(nbt(bn2)), _p(keep2), _p(wz), _p(bz), _p(y1), _p(lin2), _p(y2),
                                         _p(z), _p(mean1), _p(inv1), _p(mean2), _p(inv2), m, h1, h2, l, bn1.eps, bn2.eps,
                                         mom(bn1), mom(bn2), scale1, scale2, int(training), _stream()), "dd_mlp_tail_fwd")
        if TRACE is not None:
            TRACE.setdefault("dense", []).extend([y1, y2])
        ctx.save_for_backward(lin1, y1, lin2, y2, gamma1, gamma2, keep1, keep2, w2, wz, mean1, inv1, mean2, inv2,
                              bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var)
        ctx.cfg = (training, bn1.eps, bn2.eps, scale1, scale2)
        return z

    @staticmethod
    def backward(ctx, dz):
        (lin1, y1, lin2, y2, gamma1, gamma2, keep1, keep2, w2, wz, mean1, inv1, mean2, inv2, rm1, rv1, rm2, rv2) = ctx.saved_tensors
        training, eps1, eps2, scale1, scale2 = ctx.cfg
        m, h1 = lin1.shape
        h2, l = w2.shape[0], wz.shape[0]
        dz = dz.contiguous()
        dlin1 = torch.empty_like(lin1)
        dg1, db1, dg2, db2 = torch.empty_like(gamma1), torch.empty_like(gamma1), torch.empty_like(gamma2), torch.empty_like(gamma2)
        dw2, dwz = torch.empty_like(w2), torch.empty_like(wz)
        dbias2 = torch.empty(h2, device=dz.device, dtype=torch.float32)
        dbz = torch.empty(l, device=dz.device, dtype=torch.float32)
        check(_lib.lib().dd_mlp_tail_bwd(_p(dz), _p(lin1), _p(y1), _p(lin2), _p(y2), _p(gamma1), _p(gamma2), _p(keep1), _p(keep2),
                                         _p(w2), _p(wz), _p(mean1), _p(inv1), _p(mean2), _p(inv2), _p(rm1), _p(rv1), _p(rm2), _p(rv2),
                                         _p(dlin1), _p(dg1), _p(db1), _p(dw2), _p(dbias2), _p(dg2), _p(db2), _p(dwz), _p(dbz),
                                         m, h1, h2, l, eps1, eps2, scale1, scale2, int(training), _stream()), "dd_mlp_tail_bwd")
        return dlin1, dg1, db1, dw2, dbias2, dg2, db2, dwz, dbz, None, None, None, None, None, None



# ------------------------------------------------------------------------------------------------ losses
def _loss_ws(n, device):
    return torch.empty(_lib.lib().dd_loss_workspace_bytes(n), device=device, dtype=torch.uint8)


def _scaled_loss_grad(ctx, dz, g):
    """dz * g for the 0-dim upstream gradient ``g`` of a scalar loss.  The forward already wrote d(loss)/d(input), so for a
    plain ``loss.backward()`` (g == 1, known only on the device) nothing needs to move: ``dd_scale_by_device_scalar`` scales
    the saved buffer in place and is skipped on the device when g == 1.  The in-place pass is remembered on ``ctx``: a
    repeated backward through a retained graph (``retain_graph=True``) finds the buffer already carrying the previous factor
    and takes the generic out-of-place route with the ratio of the two factors."""
    prev = getattr(ctx, "applied_scale", None)
    if prev is not None:
        if float(prev) == 0.0:
            raise RuntimeError("loss backward: the retained graph was first back-propagated with a zero upstream gradient; "
                               "its saved loss gradient cannot be rescaled (run the forward again)")
        return dz * (g / prev)
    if (g.numel() == 1 and g.is_cuda and g.dtype == torch.float32 and dz.is_contiguous() and dz.dtype == torch.float32
            and dz.data_ptr() % 16 == 0 and not torch.is_grad_enabled()):
        check(_lib.lib().dd_scale_by_device_scalar(_p(dz), _p(g), dz.numel(), _stream()), "dd_scale_by_device_scalar")
        ctx.applied_scale = g.detach()
        return dz
    return dz * g


class BceWithLogits(torch.autograd.Function):
    """mean BCE-with-logits; the gradient is produced by the forward's single pass (roadmap_bce_v2.py:106)."""

    @staticmethod
    def forward(ctx, logits, target):
        _dev(logits, "logits")
        n = logits.numel()
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dz = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        if target.dtype in (torch.bool, torch.uint8):      # the dataset's bool road masks, read as bytes
            if not target.is_cuda or not target.is_contiguous() or target.shape != logits.shape:
                raise _lib.HotpathError(f"bce: target must be a contiguous GPU tensor of shape {tuple(logits.shape)}")
            check(_lib.lib().dd_bce_logits_u8(_p(logits), _p(target), _p(loss), _p(dz), None, n, 1.0,
                                              _p(_loss_ws(n, logits.device)), _stream()), "dd_bce_logits_u8")
        else:
            _dev(target, "target", logits.shape)
            check(_lib.lib().dd_bce_logits(_p(logits), _p(target), _p(loss), _p(dz), None, n, 1.0,
                                           _p(_loss_ws(n, logits.device)), _stream()), "dd_bce_logits")
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return _scaled_loss_grad(ctx, dz, g), None


class BceWithLogitsProbs(torch.autograd.Function):
    """(mean BCE-with-logits, sigmoid(logits)) from ONE pass over the logits: the roadmap step needs both
    (roadmap_bce_v2.py:81 and :106) and the separate sigmoid kernel would read the 82 MB of logits a second time.
    The probabilities carry no gradient (the reference takes the loss from the logits)."""

    @staticmethod
    def forward(ctx, logits, target):
        """``target``: a tensor of the logits' shape, or the collate's TUPLE of per-sample bool / uint8 masks (read through
        a pointer table: no torch.stack copy)."""
        _dev(logits, "logits")
        n = logits.numel()
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        probs = torch.empty_like(logits)
        dz = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        if isinstance(target, (tuple, list)):
            b, per = len(target), n // max(len(target), 1)
            for t in target:
                if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.bool, torch.uint8) and t.numel() == per):
                    raise _lib.HotpathError("bce: per-sample masks must be contiguous bool / uint8 GPU tensors of logits.numel() / batch elements")
            table = (C.c_void_p * b)(*[t.data_ptr() for t in target])
            check(_lib.lib().dd_bce_logits_u8_ptrs(_p(logits), table, b, per, _p(loss), _p(dz), _p(probs), 1.0,
                                                   _p(_loss_ws(n, logits.device)), _stream()), "dd_bce_logits_u8_ptrs")
            ctx.save_for_backward(dz)
            ctx.mark_non_differentiable(probs)
            ctx.set_materialize_grads(False)
            return loss, probs
        if not target.is_cuda or not target.is_contiguous() or target.shape != logits.shape:
            raise _lib.HotpathError(f"bce: target must be a contiguous GPU tensor of shape {tuple(logits.shape)}")
        if target.dtype in (torch.bool, torch.uint8):
            check(_lib.lib().dd_bce_logits_u8(_p(logits), _p(target), _p(loss), _p(dz), _p(probs), n, 1.0,
                                              _p(_loss_ws(n, logits.device)), _stream()), "dd_bce_logits_u8")
        else:
            _dev(target, "target", logits.shape)
            check(_lib.lib().dd_bce_logits(_p(logits), _p(target), _p(loss), _p(dz), _p(probs), n, 1.0,
                                           _p(_loss_ws(n, logits.device)), _stream()), "dd_bce_logits")
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)      # no 82 MB of zeros for the probabilities' (unused) gradient slot
        return loss, probs

    @staticmethod
    def backward(ctx, g, _gp):
        (dz,) = ctx.saved_tensors
        if g is None:
            return None, None
        return _scaled_loss_grad(ctx, dz, g), None


class MseLoss(torch.autograd.Function):
    """mean((pred - target)^2) (autoencoder.py:91; symmetric in its arguments)."""

    @staticmethod
    def forward(ctx, pred, target):
        _dev(pred, "pred")
        _dev(target, "target", pred.shape)
        n = pred.numel()
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        da = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        check(_lib.lib().dd_mse(_p(pred), _p(target), _p(loss), _p(da), n, 1.0, _p(_loss_ws(n, pred.device)), _stream()), "dd_mse")
        ctx.save_for_backward(da)
        return loss

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        return _scaled_loss_grad(ctx, da, g), None


class BceProbs(torch.autograd.Function):
    """mean F.binary_cross_entropy on probabilities (spatial_w_rm.py:131), gradient from the same pass."""

    @staticmethod
    def forward(ctx, probs, target):
        _dev(probs, "probs")
        _dev(target, "target", probs.shape)
        n = probs.numel()
        loss = torch.empty((), device=probs.device, dtype=torch.float32)
        dp = torch.empty_like(probs) if ctx.needs_input_grad[0] else None
        check(_lib.lib().dd_bce_probs(_p(probs), _p(target), _p(loss), _p(dp), n, 1.0, _p(_loss_ws(n, probs.device)), _stream()),
              "dd_bce_probs")
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return _scaled_loss_grad(ctx, dp, g), None


def sigmoid(z):
    """sigmoid(logits) outside autograd (the reference's second forward output, roadmap_bce_v2.py:81)."""
    _dev(z, "z")
    if z.numel() % 4:
        raise _lib.HotpathError("sigmoid: element count must be a multiple of 4")
    p = torch.empty_like(z)
    check(_lib.lib().dd_sigmoid(_p(z), _p(p), z.numel(), _stream()), "dd_sigmoid")
    return p


class Sigmoid(torch.autograd.Function):
    """p = sigmoid(z) with its autograd on the device (dd_sigmoid / dd_sigmoid_bwd): the sigmoid INSIDE ``RoadMap.forward``
    of the MSE twin (roadmap_pretrain_ae.py:76), whose loss is taken from the probabilities."""

    @staticmethod
    def forward(ctx, z):
        p = sigmoid(z.contiguous())
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp = dp.contiguous()
        _dev(dp, "dp", p.shape)
        dz = torch.empty_like(p)
        check(_lib.lib().dd_sigmoid_bwd(_p(dp), _p(p), _p(dz), p.numel(), _stream()), "dd_sigmoid_bwd")
        return dz


def sigmoid_and_loss(logits, target):
    """One pass: (loss, probs) without autograd -- used by validation."""
    n = logits.numel()
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    probs = torch.empty_like(logits)
    check(_lib.lib().dd_bce_logits(_p(logits), _p(target), _p(loss), None, _p(probs), n, 1.0, _p(_loss_ws(n, logits.device)),
                                   _stream()), "dd_bce_logits")
    return loss, probs


# ------------------------------------------------------------------------------------------------ optimizer
def adam_step_multi(tensors, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """One launch for a list of small (p, g, m, v) quadruples that share ``step`` (dd_adam_step_multi)."""
    if not tensors:
        return
    table = (_lib.AdamTensor * len(tensors))()
    for i, quad in enumerate(tensors):
        for name, t in zip("pgmv", quad):
            _dev(t, name, quad[0].shape)
        table[i] = _lib.AdamTensor(_p(quad[0]), _p(quad[1]), _p(quad[2]), _p(quad[3]), quad[0].numel())
    check(_lib.lib().dd_adam_step_multi(table, len(tensors), lr, beta1, beta2, eps, int(step), grad_scale, _stream()),
          "dd_adam_step_multi")


def adam_step_rankb(p, m, v, dy, x, bias, bias_m, bias_v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """Adam on the Linear weight ``p`` [n, k] (moments ``m``, ``v``) with its gradient dy^T x formed inside the pass from the layer's
    output gradient ``dy`` [rows, n] and input ``x`` [rows, k]; ``bias`` (optional, with its moments) is updated from dy's column sums
    in the same launch (dd_adam_step_rankb)."""
    rows, n = dy.shape
    k = x.shape[1]
    _dev(dy, "dy")
    _dev(x, "x", (rows, k))
    for name, t in (("p", p), ("m", m), ("v", v)):
        _dev(t, name, (n, k))
    if bias is not None:
        for name, t in (("bias", bias), ("bias_m", bias_m), ("bias_v", bias_v)):
            _dev(t, name, (n,))
    check(_lib.lib().dd_adam_step_rankb(_p(p), _p(m), _p(v), _p(dy), _p(x), rows, n, k, _p(bias), _p(bias_m), _p(bias_v),
                                        lr, beta1, beta2, eps, int(step), grad_scale, _stream()), "dd_adam_step_rankb")


def adam_step_flat(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    for name, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _dev(t, name, p.shape)
    check(_lib.lib().dd_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, int(step), grad_scale,
                                  _stream()), "dd_adam_step")
