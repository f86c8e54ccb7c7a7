"""Host-side planning for the generic NHWC convolution kernels (``dd_gconv_*``, csrc/gconv.hip).

A ``Layer`` describes one ``nn.Conv2d`` / ``nn.ConvTranspose2d`` of the reference in its own terms (kernel,
stride, dilation, padding, output_padding) and knows how to express its forward, its data gradient and its
weight gradient as launches of the ONE implicit-GEMM kernel family:

  Conv2d                      forward: plain;            dgrad: flipped taps, pad' = d(k-1)-p, div = stride
  ConvTranspose2d stride 1    forward: flipped taps, pad' = d(k-1)-p (+output_padding rows/cols);  dgrad: plain conv
  ConvTranspose2d k = s = 2   forward: 4 launches of a 1x1 conv, one per output phase;  dgrad: k2 s2 conv
Buffers are NHWC; a layer may read a channel slice of its input buffer and write a channel slice / a
sub-rectangle of its output buffer (mosaic tiling and channel concat without copies).
"""
import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _lib
from . import ddp as _ddp
from ._lib import GConvDesc, check

EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_RELU_MASK, EPI_BIAS_SIGMOID = 0, 1, 2, 3, 4


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


def _chk(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _lib.HotpathError(f"{name}: expected a contiguous fp32 device tensor")
    return t


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


@dataclass
class View:
    """A rectangular, channel-sliced window of an NHWC buffer [B, mem_h, mem_w, cstore]."""
    buf: torch.Tensor
    coff: int = 0
    chans: int = None
    off_h: int = 0
    off_w: int = 0
    h: int = None
    w: int = None

    def __post_init__(self):
        b, mh, mw, cs = self.buf.shape
        self.chans = cs - self.coff if self.chans is None else self.chans
        self.h = mh - self.off_h if self.h is None else self.h
        self.w = mw - self.off_w if self.w is None else self.w
        if self.off_h or self.off_w:
            assert self.coff % 1 == 0


def _desc(batch, src, dst, cin, cout, k, stride=(1, 1), dil=(1, 1), pad=(0, 0), div=(1, 1), out_hw=None,
          ostride=(1, 1), ooff=(0, 0), mask_pass=(0, 0)):
    """src/dst: View.  src window offsets are folded into the padding (a shifted origin)."""
    _, imh, imw, ics = src.buf.shape
    _, omh, omw, ocs = dst.buf.shape
    assert src.off_h == 0 and src.off_w == 0, "input windows are not needed by this path"
    oh, ow = out_hw if out_hw is not None else (dst.h, dst.w)
    return GConvDesc(batch, imh, imw, ics, src.coff, cin, oh, ow, omh, omw, ocs, dst.coff, cout, k[0], k[1],
                     stride[0], stride[1], dil[0], dil[1], pad[0], pad[1], div[0], div[1], ostride[0], ostride[1],
                     dst.off_h + ooff[0], dst.off_w + ooff[1], mask_pass[0], mask_pass[1])


def _pack(w, d, w_off, sn, sc, flip, n_real, c_real):
    n = _lib.lib().dd_gconv_packed_floats(C.byref(d))
    if n <= 0:
        raise _lib.HotpathError(f"gconv: {_lib.lib().dd_last_error().decode()}")
    packed = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().dd_gconv_pack(_p(w), _p(packed), C.byref(d), w_off, sn, sc, int(flip), n_real, c_real, _stream()), "dd_gconv_pack")
    return packed


def _fwd(x, packed, bias, mask, y, d, epi):
    check(_lib.lib().dd_gconv_fwd(_p(x), _p(packed), _p(bias), _p(mask), _p(y), C.byref(d), epi, _stream()), "dd_gconv_fwd")


# The dilated stride-1 layers (the box heads' up-convs) have their own phase-decomposed, LDS-staged kernel (csrc/dconv.hip);
# DD_DCONV=0 routes them through the generic gather engine again (A/B knob, same results up to summation order).
DCONV = os.environ.get("DD_DCONV", "1") != "0"
# (Strided Conv2d layers take their data gradient by input phase, Layer._backward_data_phased: the gather with a divisibility test per tap
# it replaced lost its A/B in round 2 and is gone.)


# Precision mode "fp32x3" (csrc/dconv_split.hip; off by default): the forward, data gradient and weight gradient of up_conv_1 / up_conv_2
# with every fp32 product taken as six bf16 x bf16 products (exact 3-way operand split, fp32 accumulation) on the bf16 matrix pipe; error
# bound: DESIGN.md 3.3d.  Selected per module -- ``hparams.precision = "fp32x3"`` on BBSpatialRoadMap / JointRoadMapBBox, or
# ``box_merge.precision = "fp32x3"`` -- through ``split_products()`` below; ``gconv.SPLIT_BF16 = True`` is the process-wide default
# (DD_DCONV_SPLIT=1 sets it at import: tools' A/B).  The default everywhere stays the exact-fp32 MFMA kernels.
SPLIT_BF16 = os.environ.get("DD_DCONV_SPLIT", "0") == "1"


class split_products:
    """``with split_products(True):`` -- the layers run inside take the split-product kernels where those serve them (False: the exact
    ones; None: whatever the process-wide default says).  heads.MergeFn records the mode of its forward and re-enters it in its
    backward, so a module's choice holds for the whole step however the backward is started."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        global SPLIT_BF16
        self.prev = SPLIT_BF16
        if self.on is not None:
            SPLIT_BF16 = bool(self.on)
        return self

    def __exit__(self, *exc):
        global SPLIT_BF16
        SPLIT_BF16 = self.prev
        return False
# ... its kernels write the bf16 planes of their OUTPUT from the epilogue for the next layer (round 4 A/B against a split pass per operand:
# 44.75-45.26 -> 44.21-44.47 ms)


def split_rows(view):
    """The three-plane bf16 image (csrc/dconv_split.hip) of the channels of a whole-buffer View: [B][H][chans / 16][W][112 B]."""
    if not (_whole(view) and view.chans % 16 == 0):
        raise _lib.HotpathError("split_rows: a whole-buffer View with a multiple of 16 channels")
    b, h, w, cs = view.buf.shape
    xs = torch.empty(b * h * (view.chans // 16) * w * 112, device=view.buf.device, dtype=torch.uint8)
    check(_lib.lib().dd_dconv_split_rows(_p(_chk(view.buf, "x")), _p(xs), b * h, w, cs, view.coff, view.chans, _stream()), "dd_dconv_split_rows")
    return xs


K2S2_WGRAD = os.environ.get("DD_K2S2_WGRAD", "1") != "0"      # A/B knob: 0 = the k2 s2 32->32 weight gradient by four phase launches
SSCONV_FWD = os.environ.get("DD_SSCONV_FWD", "1") != "0"      # A/B knob: 0 = ss_conv's forward on the generic engine
SSCONV_DGRAD = os.environ.get("DD_SSCONV_DGRAD", "1") != "0"      # A/B knob: 0 = ss_conv's data gradient by seven phase launches


def _ops():
    from . import ops      # ops imports nothing from here; deferred so that either module can be imported first
    return ops


def _p_weight(w):
    return w if w.is_contiguous() else w.contiguous()


def _whole(v):
    """The View covers every pixel of its buffer (the kernels that take whole-buffer dimensions must not be handed a window)."""
    return v.off_h == 0 and v.off_w == 0 and v.h == v.buf.shape[1] and v.w == v.buf.shape[2]


def _dconv_ok(d):
    return DCONV and bool(_lib.lib().dd_dconv_supported(C.byref(d)))


def _conv(x, weight, bias, mask, y, d, epi, w_off, sn, sc, flip, n_real, c_real, xs=None, emit=None, colsum=None):
    """pack + launch on the dilated kernel when the descriptor qualifies, on the generic one otherwise.  Returns the split image of
    x when the split-product path ran (``xs``: one a caller already holds), None otherwise.  ``emit`` (a dict): on the split path the
    kernel also writes the split image of its OUTPUT from its epilogue (``emit['ys']``: the next layer's operand, no split pass)."""
    lib = _lib.lib()
    if SPLIT_BF16 and lib.dd_dconv_split_supported(C.byref(d)) and (
            (d.pad_h > 0 and mask is None and epi in (EPI_NONE, EPI_BIAS, EPI_BIAS_RELU)) or (d.pad_h == 0 and epi in (EPI_NONE, EPI_RELU_MASK))):
        packed = torch.empty(lib.dd_dconv_split_packed_bytes(C.byref(d)), device=x.device, dtype=torch.uint8)
        if xs is None:
            xs = torch.empty(lib.dd_dconv_split_input_bytes(C.byref(d)), device=x.device, dtype=torch.uint8)
            check(lib.dd_dconv_split_input(_p(x), _p(xs), C.byref(d), _stream()), "dd_dconv_split_input")
        check(lib.dd_dconv_split_pack(_p(weight), _p(packed), C.byref(d), w_off, sn, sc, int(flip), n_real, c_real, _stream()), "dd_dconv_split_pack")
        ys = None
        if emit is not None and d.cout % 16 == 0 and d.out_coff == 0 and d.ooff_h == 0 and d.ooff_w == 0 and d.omem_h == d.out_h and d.omem_w == d.out_w:
            ys = torch.empty(d.batch * d.out_h * (d.cout // 16) * d.out_w * 112, device=x.device, dtype=torch.uint8)
            emit["ys"] = ys
        check(lib.dd_dconv_fwd_split(_p(xs), _p(packed), _p(bias), _p(mask), _p(y), _p(ys), C.byref(d), epi, _stream()), "dd_dconv_fwd_split")
        return xs
    if _dconv_ok(d):
        n = lib.dd_dconv_packed_floats(C.byref(d))
        packed = torch.empty(n, device=weight.device, dtype=torch.float32)
        check(lib.dd_dconv_pack(_p(weight), _p(packed), C.byref(d), w_off, sn, sc, int(flip), n_real, c_real, _stream()), "dd_dconv_pack")
        if (colsum is not None and bias is None and epi in (EPI_NONE, EPI_RELU_MASK)
                and lib.dd_dconv_colsum_supported(C.byref(d), epi, int(mask is not None))):
            # the launch also leaves the per-channel sums of its output: the bias gradient of the layer below (csrc/dconv_m.hip)
            nbytes = lib.dd_dconv_colsum_workspace_bytes()
            ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
            check(lib.dd_dconv_fwd_colsum(_p(x), _p(packed), _p(mask), _p(y), _p(colsum), C.byref(d), epi, _p(ws), nbytes, _stream()),
                  "dd_dconv_fwd_colsum")
            return "colsum"
        check(lib.dd_dconv_fwd(_p(x), _p(packed), _p(bias), _p(mask), _p(y), C.byref(d), epi, _stream()), "dd_dconv_fwd")
    else:
        _fwd(x, _pack(weight, d, w_off, sn, sc, flip, n_real, c_real), bias, mask, y, d, epi)


def _wgrad(x, dy, dw, db, d, w_off, sn, sc, flip, n_real, c_real, accumulate):
    nbytes = _lib.lib().dd_gconv_wgrad_workspace_bytes(C.byref(d))
    if nbytes <= 0:
        raise _lib.HotpathError(f"gconv_wgrad: {_lib.lib().dd_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(_lib.lib().dd_gconv_wgrad(_p(x), _p(dy), _p(dw), _p(db), C.byref(d), w_off, sn, sc, int(flip), n_real, c_real,
                                    int(accumulate), _p(ws), nbytes, _stream()), "dd_gconv_wgrad")


class Layer:
    """One Conv2d (``transposed=False``, weight OIHW) or ConvTranspose2d (``transposed=True``, weight IOHW)."""

    def __init__(self, cin, cout, k, stride=1, dil=1, pad=0, transposed=False, output_padding=0):
        self.cin, self.cout = cin, cout
        self.k, self.stride, self.dil, self.pad = _pair(k), _pair(stride), _pair(dil), _pair(pad)
        self.opad = _pair(output_padding)
        self.transposed = transposed
        self.T = self.k[0] * self.k[1]
        self.cin_store = (cin + 3) // 4 * 4
        self.k2s2 = transposed and self.k == (2, 2) and self.stride == (2, 2)
        if transposed and not self.k2s2 and self.stride != (1, 1):
            raise _lib.HotpathError("ConvTranspose2d: only stride 1 (dilated) and k2 s2 are built")

    # ---- shapes
    def out_hw(self, h, w):
        k, s, d, p = self.k, self.stride, self.dil, self.pad
        if self.transposed:
            return tuple((n - 1) * s[i] - 2 * p[i] + d[i] * (k[i] - 1) + self.opad[i] + 1 for i, n in enumerate((h, w)))
        return tuple((n + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i, n in enumerate((h, w)))

    def _flip_pad(self):
        return tuple(self.dil[i] * (self.k[i] - 1) - self.pad[i] for i in range(2))

    # ---- forward: writes dst (View) from src (View)
    def forward(self, weight, bias, src, dst, epilogue, mask=None, keep=None, xs=None, emit=None):
        """``keep`` (a dict): receives ``keep['xs']``, the split image of the input, when the split-product experiment ran -- the
        weight gradient of the same layer takes it back (``backward_weight(xs=...)``) instead of splitting x again.  ``xs``: that image
        when the caller already holds it (the previous layer's ``emit['ys']``); ``emit`` (a dict): also write the output's."""
        b = src.buf.shape[0]
        cs = src.chans if src.chans % 4 == 0 else self.cin_store
        _chk(weight, "weight")
        if (not self.transposed and self.k == (1, 24) and self.stride == (1, 7) and self.dil == (1, 1) and self.pad == (0, 0)
                and self.cin == 32 and self.cout == 32 and mask is None and epilogue in (EPI_BIAS, EPI_BIAS_RELU) and SSCONV_FWD
                and _whole(src) and _whole(dst) and src.coff == 0 and dst.coff == 0 and _ops().ssconv_dgrad_ok(dst.buf, src.buf)):
            _ops().ssconv_fwd(src.buf, _p_weight(weight), bias, dst.buf, relu=epilogue == EPI_BIAS_RELU)      # ss_conv (csrc/ssconv.hip)
        elif not self.transposed:
            d = _desc(b, src, dst, cs, self.cout, self.k, self.stride, self.dil, self.pad)
            _conv(src.buf, weight, bias, mask, dst.buf, d, epilogue, 0, self.cin * self.T, self.T, False, self.cout, self.cin)
        elif self.k2s2:
            ih, iw = src.buf.shape[1:3]
            for ph in range(4):
                d = _desc(b, src, dst, cs, self.cout, (1, 1), out_hw=(ih, iw), ostride=(2, 2), ooff=(ph // 2, ph % 2))
                pk = _pack(weight, d, ph, 4, self.cout * 4, False, self.cout, self.cin)
                _fwd(src.buf, pk, bias, mask, dst.buf, d, epilogue)
        else:
            d = _desc(b, src, dst, cs, self.cout, self.k, (1, 1), self.dil, self._flip_pad())
            xs = _conv(src.buf, weight, bias, mask, dst.buf, d, epilogue, 0, self.T, self.cout * self.T, True, self.cout, self.cin, xs=xs, emit=emit)
            if keep is not None and xs is not None:
                keep["xs"] = xs

    # ---- data gradient: dsrc (View with the input's geometry, >= 4-aligned channels) from ddst (View of dy)
    def backward_data(self, weight, ddst, dsrc, relu_src=None, mask_pass=(0, 0), gs=None, emit=None, colsum=None):
        """dsrc.buf[..., dsrc.coff : +cin] = dL/dx (x masked by ``relu_src > 0`` when given; channels
        [mask_pass[0], mask_pass[1]) of the dsrc buffer are exempt: a concat slice that is not a ReLU output).
        ``colsum`` (a [cin] float tensor): where the kernel can, it also receives the per-channel sums of what was written -- the bias
        gradient of the layer that produced x; returns True when it was filled."""
        b = ddst.buf.shape[0]
        epi = EPI_RELU_MASK if relu_src is not None else EPI_NONE
        cin_out = self.cin
        cos = self.cout if self.cout % 4 == 0 else (self.cout + 3) // 4 * 4
        if self.transposed and not self.k2s2 and cin_out <= 96:
            # the dilated kernel takes up to 96 output channels in one launch (three column tiles per wave)
            d = _desc(b, ddst, dsrc, cos, cin_out, self.k, (1, 1), self.dil, self.pad, mask_pass=mask_pass)
            if _dconv_ok(d):
                r = _conv(ddst.buf, weight, None, relu_src, dsrc.buf, d, epi, 0, self.cout * self.T, self.T, False, cin_out, self.cout, xs=gs,
                          emit=emit, colsum=colsum)
                return isinstance(r, str) and r == "colsum"
        if (not self.transposed and self.k == (1, 24) and self.stride == (1, 7) and self.dil == (1, 1) and self.pad == (0, 0)
                and self.cin == 32 and self.cout == 32 and relu_src is None and SSCONV_DGRAD and _whole(ddst) and _whole(dsrc)
                and ddst.coff == 0 and dsrc.coff == 0 and _ops().ssconv_dgrad_ok(ddst.buf, dsrc.buf)):
            _ops().ssconv_dgrad(ddst.buf, _p_weight(weight), dsrc.buf)      # ss_conv: all seven phases in one launch (csrc/ssconv.hip)
            return
        if not self.transposed and self.stride != (1, 1) and self.dil == (1, 1) and self.pad == (0, 0) and cin_out <= 64:
            return self._backward_data_phased(weight, ddst, dsrc, relu_src, mask_pass, epi, cos)
        for n0 in range(0, cin_out, 64):                    # the generic kernel writes at most 64 channels per launch
            nn = min(64, cin_out - n0)
            out = View(dsrc.buf, dsrc.coff + n0, nn, dsrc.off_h, dsrc.off_w, dsrc.h, dsrc.w)
            msk = relu_src
            if not self.transposed:
                d = _desc(b, ddst, out, cos, nn, self.k, (1, 1), self.dil, self._flip_pad(), div=self.stride, mask_pass=mask_pass)
                if _dconv_ok(d):      # stride-1 3x3 layers (out_conv, rm_conv_2): the LDS-tiled kernel
                    _conv(ddst.buf, weight, None, msk, out.buf, d, epi, n0 * self.T, self.T, self.cin * self.T, True, nn, self.cout)
                    continue
                pk = _pack(weight, d, n0 * self.T, self.T, self.cin * self.T, True, nn, self.cout)
            elif self.k2s2:
                d = _desc(b, ddst, out, cos, nn, (2, 2), (2, 2), mask_pass=mask_pass)
                pk = _pack(weight, d, n0 * self.cout * 4, self.cout * 4, 4, False, nn, self.cout)
            else:
                d = _desc(b, ddst, out, cos, nn, self.k, (1, 1), self.dil, self.pad, mask_pass=mask_pass)
                pk = _pack(weight, d, n0 * self.cout * self.T, self.cout * self.T, self.T, False, nn, self.cout)
            _fwd(ddst.buf, pk, None, msk, out.buf, d, epi)

    def _backward_data_phased(self, weight, ddst, dsrc, relu_src, mask_pass, epi, cos):
        """Data gradient of a strided Conv2d (no padding, no dilation) by input phase: input pixels of equal residue
        (iy mod sh, ix mod sw) = (ry, rx) only ever meet the taps ky = ry + sh*jy, kx = rx + sw*jx, so
            dx[sh*my + ry, sw*mx + rx] = sum_j g[my - jy, mx - jx] * w[ry + sh*jy, rx + sw*jx]
        is a stride-1 transposed convolution of g with the (ry, rx) sub-kernel, written at output stride (sh, sw), offset
        (ry, rx).  The gather with a divisibility test visits all kh*kw taps for every pixel (ss_conv, k 1x24 stride 7: 24 taps
        for the 3.4 that contribute -- 3.6 ms at bs 32); the sh*sw phase launches do exactly the useful work (0.5 ms)."""
        b = ddst.buf.shape[0]
        sh, sw = self.stride
        for ry in range(min(sh, self.k[0])):
            for rx in range(min(sw, self.k[1])):
                sub = weight[:, :, ry::sh, rx::sw].contiguous()                  # [cout, cin, njy, njx]
                njy, njx = sub.shape[2:]
                t = njy * njx
                mh, mw = (dsrc.h - ry + sh - 1) // sh, (dsrc.w - rx + sw - 1) // sw      # pixels of this phase
                out = View(dsrc.buf, dsrc.coff, self.cin, dsrc.off_h, dsrc.off_w, dsrc.h, dsrc.w)
                d = _desc(b, ddst, out, cos, self.cin, (njy, njx), (1, 1), (1, 1), (njy - 1, njx - 1), out_hw=(mh, mw),
                          ostride=(sh, sw), ooff=(ry, rx), mask_pass=mask_pass)
                pk = _pack(sub, d, 0, t, self.cin * t, True, self.cin, self.cout)
                _fwd(ddst.buf, pk, None, relu_src, out.buf, d, epi)
        # phases past the kernel extent (stride > kernel) receive no gradient
        for ry in range(sh):
            for rx in range(sw):
                if ry >= self.k[0] or rx >= self.k[1]:
                    dsrc.buf[:, dsrc.off_h + ry:dsrc.off_h + dsrc.h:sh, dsrc.off_w + rx:dsrc.off_w + dsrc.w:sw, dsrc.coff:dsrc.coff + self.cin] = 0

    # ---- weight (+bias) gradient
    def split_wgrad_ok(self, src, ddst):
        """The split-product weight gradient (csrc/dconv_split.hip) serves this call: the experiment is on, a k7 d7 layer it is built
        for, whole-buffer Views."""
        return (SPLIT_BF16 and self.transposed and not self.k2s2 and self.k[0] == self.k[1] and self.dil[0] == self.dil[1] and self.pad == (0, 0)
                and src.chans == self.cin and ddst.chans == self.cout and _whole(src) and _whole(ddst)
                and bool(_lib.lib().dd_dconv_wgrad_split_supported(self.k[0], self.dil[0], self.cin, self.cout)))

    def backward_weight(self, src, ddst, want_bias=True, xs=None, gs=None):
        b = src.buf.shape[0]
        dev = src.buf.device
        cs = src.chans if src.chans % 4 == 0 else self.cin_store
        if self.transposed:
            dw = torch.empty((self.cin, self.cout) + self.k, device=dev, dtype=torch.float32)
        else:
            dw = torch.empty((self.cout, self.cin) + self.k, device=dev, dtype=torch.float32)
        db = torch.empty(self.cout, device=dev, dtype=torch.float32) if want_bias else None
        if not self.transposed:
            d = _desc(b, src, ddst, cs, self.cout, self.k, self.stride, self.dil, self.pad)
            _wgrad(src.buf, ddst.buf, dw, db, d, 0, self.cin * self.T, self.T, False, self.cout, self.cin, 0)
        elif (self.k2s2 and K2S2_WGRAD and self.cin == 32 and self.cout == 32 and _whole(src) and _whole(ddst) and src.coff == 0
              and src.buf.shape[3] == 32 and src.buf.is_contiguous() and ddst.buf.is_contiguous() and src.buf.shape[2] >= 2
              and ddst.buf.shape[1] == 2 * src.buf.shape[1] and ddst.buf.shape[2] == 2 * src.buf.shape[2]):
            # ss_deconv / the decoder's dc3: the four phases in one launch (csrc/gconv.hip, deconv2x2_c32_wgrad_kernel)
            lib = _lib.lib()
            nbytes = lib.dd_deconv2x2_c32_wgrad_workspace_bytes()
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            _, ih, iw, _ = src.buf.shape
            check(lib.dd_deconv2x2_c32_wgrad(_p(src.buf), _p(ddst.buf), _p(dw), _p(db) if db is not None else None, b, ih, iw,
                                             ddst.buf.shape[3], ddst.coff, _p(ws), nbytes, _stream()), "dd_deconv2x2_c32_wgrad")
        elif self.k2s2:
            ih, iw = src.buf.shape[1:3]
            for ph in range(4):
                d = _desc(b, src, ddst, cs, self.cout, (1, 1), out_hw=(ih, iw), ostride=(2, 2), ooff=(ph // 2, ph % 2))
                _wgrad(src.buf, ddst.buf, dw, db, d, ph, 4, self.cout * 4, False, self.cout, self.cin, 2 if ph > 0 else 0)
        elif self.split_wgrad_ok(src, ddst):
            # EXPERIMENT: both operands as three bf16 planes, six bf16 x bf16 products per fp32 product (csrc/dconv_split.hip); the
            # split images of x (from the forward) and of dL/dy (from the data gradient) are taken over when the caller holds them
            lib = _lib.lib()
            xs = split_rows(src) if xs is None else xs
            gs = split_rows(ddst) if gs is None else gs
            nbytes = lib.dd_dconv_wgrad_split_workspace_bytes(self.cin, self.cout)
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            _, ih, iw, _ = src.buf.shape
            _, gh, gw, _ = ddst.buf.shape
            check(lib.dd_dconv_wgrad_split(_p(xs), _p(gs), _p(dw), b, ih, iw, self.cin, gh, gw, self.cout, 0, _p(ws), nbytes, _stream()),
                  "dd_dconv_wgrad_split")
            if want_bias:
                channel_sum(ddst, db)
        elif (DCONV and self.k[0] == self.k[1] and self.dil[0] == self.dil[1] and self.pad == (0, 0) and src.chans == self.cin
              and ddst.chans == self.cout and _whole(src) and _whole(ddst)
              and _lib.lib().dd_dconv_wgrad_supported(self.k[0], self.dil[0], self.cin, self.cout)):
            # the box heads' dilated up-convs: LDS-staged weight-gradient kernel (csrc/dconv.hip)
            lib = _lib.lib()
            nbytes = lib.dd_dconv_wgrad_workspace_bytes(self.k[0], self.dil[0], self.cin, self.cout)
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            _, ih, iw, ics = src.buf.shape
            _, gh, gw, gcs = ddst.buf.shape
            check(lib.dd_dconv_wgrad(_p(src.buf), _p(ddst.buf), _p(dw), b, ih, iw, ics, src.coff, self.cin, gh, gw, gcs, ddst.coff, self.cout,
                                     self.k[0], self.dil[0], 0, _p(ws), nbytes, _stream()), "dd_dconv_wgrad")
            if want_bias:
                channel_sum(ddst, db)
        elif self.cin <= 96 and self.cout % 4 == 0 and ddst.off_h == 0 and ddst.off_w == 0:
            # stride-1 transposed conv, role-swapped: dWt[c][o][t] = sum over INPUT pixels p of x[p][c] * dy[p + t*d - pad][o]
            # is the weight gradient of the plain conv  dx = conv(dy, Wt)  with x in the role of its output gradient.
            # Every tap of every pixel is inside dy there, whereas the flipped-tap form spends (out/in)^2 - 1 of its
            # MFMAs on the zero border.  The bias gradient (sum of dy) then needs its own per-channel sum.
            d = _desc(b, ddst, src, self.cout, self.cin, self.k, (1, 1), self.dil, self.pad)
            _wgrad(ddst.buf, src.buf, dw, None, d, 0, self.cout * self.T, self.T, False, self.cin, self.cout, 0)
            if want_bias:
                channel_sum(ddst, db)
        else:
            d = _desc(b, src, ddst, cs, self.cout, self.k, (1, 1), self.dil, self._flip_pad())
            _wgrad(src.buf, ddst.buf, dw, db, d, 0, self.T, self.cout * self.T, True, self.cout, self.cin, 0)
        return dw, db


def copy_channels(src, dst):
    """dst.buf[..., dst.coff : +chans] = src.buf[..., src.coff : +chans] over the Views' windows (equal h x w; whole buffers of equal
    pixel count take the one-dimensional kernel)."""
    assert src.chans == dst.chans and src.buf.shape[0] == dst.buf.shape[0] and (src.h, src.w) == (dst.h, dst.w)
    whole = all(v.off_h == 0 and v.off_w == 0 and v.h == v.buf.shape[1] and v.w == v.buf.shape[2] for v in (src, dst))
    if not whole:
        check(_lib.lib().dd_copy_channels_window(_p(_chk(src.buf, "src")), _p(_chk(dst.buf, "dst")), src.buf.shape[0], src.h, src.w, src.chans,
                                                 src.buf.shape[1], src.buf.shape[2], src.off_h, src.off_w, src.buf.shape[3], src.coff,
                                                 dst.buf.shape[1], dst.buf.shape[2], dst.off_h, dst.off_w, dst.buf.shape[3], dst.coff,
                                                 _stream()), "dd_copy_channels_window")
        return
    npix = src.buf.shape[0] * src.buf.shape[1] * src.buf.shape[2]
    check(_lib.lib().dd_copy_channels(_p(_chk(src.buf, "src")), _p(_chk(dst.buf, "dst")), npix, src.chans, src.buf.shape[3], src.coff,
                                      dst.buf.shape[3], dst.coff, _stream()), "dd_copy_channels")


def channel_sum(view, out, accumulate=False):
    """out[c] = sum over all pixels of view.buf[..., view.coff + c] (dense buffers only)."""
    b, mh, mw, cs = view.buf.shape
    assert view.off_h == 0 and view.off_w == 0 and view.h == mh and view.w == mw
    ws = torch.empty(_lib.lib().dd_channel_sum_workspace_bytes(), device=view.buf.device, dtype=torch.uint8)
    check(_lib.lib().dd_channel_sum(_p(view.buf), _p(out), b * mh * mw, cs, view.coff, view.chans, int(accumulate), _p(ws),
                                    _stream()), "dd_channel_sum")


def view_to_nhwc4(views, view, transform):
    """views [B,6,3,H,W] -- or the collate's tuple / list of B per-sample [6,3,H,W] tensors, read through a pointer table
    (no torch.stack) -- -> one view as NHWC4 with SpatialMappingCNN's rot90/flip applied (see dd_view_to_nhwc4).  uint8 frames
    ([B,6,H,W,3] or a tuple of [6,H,W,3]) are read as they are, ToTensor's /255 fused (dd_view_to_nhwc4_u8_ptrs)."""
    from . import ops
    if ops.is_u8_frames(views):
        table, b, h, w, dev, _keep = ops.u8_table(views, "view_to_nhwc4")
        oh, ow = (w, h) if transform in (1, 2) else (h, w)
        out = torch.empty((b, oh, ow, 4), device=dev, dtype=torch.float32)
        check(_lib.lib().dd_view_to_nhwc4_u8_ptrs(table, _p(out), b, h, w, view, transform, _stream()), "dd_view_to_nhwc4_u8_ptrs")
        return out
    if isinstance(views, (tuple, list)):
        b = len(views)
        _, _, h, w = views[0].shape
        for t in views:
            if not (_chk(t, "sample").dim() == 4 and tuple(t.shape) == (6, 3, h, w)):
                raise _lib.HotpathError(f"view_to_nhwc4: expected samples of [6,3,{h},{w}], got {tuple(t.shape)}")
        oh, ow = (w, h) if transform in (1, 2) else (h, w)
        out = torch.empty((b, oh, ow, 4), device=views[0].device, dtype=torch.float32)
        table = (C.c_void_p * b)(*[t.data_ptr() for t in views])
        check(_lib.lib().dd_view_to_nhwc4_ptrs(table, _p(out), b, h, w, view, transform, _stream()), "dd_view_to_nhwc4_ptrs")
        return out
    b, _, _, h, w = views.shape
    oh, ow = (w, h) if transform in (1, 2) else (h, w)
    out = torch.empty((b, oh, ow, 4), device=views.device, dtype=torch.float32)
    check(_lib.lib().dd_view_to_nhwc4(_p(_chk(views, "views")), _p(out), b, h, w, view, transform, _stream()), "dd_view_to_nhwc4")
    return out


def _sample_table(views, who):
    """-> (ctypes table of per-sample device pointers, B, H, W, device, u8 flag, keepalive) for every form the data pipeline hands over:
    fp32 [B,6,3,H,W], the collate's tuple of fp32 [6,3,H,W], uint8 [B,6,H,W,3] frames or a tuple of uint8 [6,H,W,3]."""
    from . import ops
    if ops.is_u8_frames(views):
        table, b, h, w, dev, keep = ops.u8_table(views, who)
        return table, b, h, w, dev, 1, keep
    items = [views[i] for i in range(views.shape[0])] if isinstance(views, torch.Tensor) else list(views)
    if not items:
        raise _lib.HotpathError(f"{who}: empty batch")
    _, _, h, w = items[0].shape
    keep = []
    for t in items:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and tuple(t.shape) == (6, 3, h, w)):
            raise _lib.HotpathError(f"{who}: expected fp32 [6,3,{h},{w}] samples on the GPU, got {tuple(getattr(t, 'shape', ()))}")
        keep.append(t if t.is_contiguous() else t.contiguous())
    table = (C.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
    return table, len(keep), h, w, keep[0].device, 0, keep


STRIP6 = os.environ.get("DD_STRIP6", "1") != "0"      # A/B knob: 0 = the six strip convs of SpatialMappingCNN on the generic engine (round 4)


def strip6_supported(h, w):
    return STRIP6 and bool(_lib.lib().dd_strip6_supported(int(h), int(w)))


def _ptr6(tensors, who, shapes=None):
    for i, t in enumerate(tensors):
        _chk(t, f"{who}[{i}]")
        if shapes is not None and tuple(t.shape) != tuple(shapes[i]):
            raise _lib.HotpathError(f"{who}[{i}]: shape {tuple(t.shape)} != {tuple(shapes[i])}")
    return (C.c_void_p * 6)(*[t.data_ptr() for t in tensors])


_STRIP_SHAPES = [(32, 3, 1, 50), (32, 3, 1, 50), (32, 3, 52, 1), (32, 3, 52, 1), (32, 3, 1, 50), (32, 3, 1, 50)]      # bl, fl, b, f, br, fr


def strip6_fwd(views, weights, biases, want_bits=False):
    """The six strip convs of SpatialMappingCNN (+ bias + ReLU) in one launch, written into their tiles of the 3 x 2 mosaic
    (dd_strip6_fwd; spatial_bb/components.py:34-73).  ``weights`` / ``biases``: bl, fl, b, f, br, fr.  -> mosaic [B, 3 th, 2 tw, 32];
    with ``want_bits`` also its sign words, int32 [B, 3 th, 2 tw] (the ReLU mask of out_conv's data gradient on the Winograd kernels)."""
    table, b, h, w, dev, u8, _keep = _sample_table(views, "strip6_fwd")
    th, tw = (h - 1) // 3 + 1, (w - 50) // 2 + 1
    weights = [x.contiguous() for x in weights]
    mosaic = torch.empty((b, 3 * th, 2 * tw, 32), device=dev, dtype=torch.float32)
    bits = torch.empty((b, 3 * th, 2 * tw), device=dev, dtype=torch.int32) if want_bits else None
    check(_lib.lib().dd_strip6_fwd(table, u8, _ptr6(weights, "weight", _STRIP_SHAPES), _ptr6(biases, "bias", [(32,)] * 6), _p(mosaic),
                                   _p(bits) if want_bits else None, b, h, w, _stream()), "dd_strip6_fwd")
    return (mosaic, bits) if want_bits else mosaic


def strip6_wgrad(views, g):
    """Weight and bias gradients of the six strip convs from dL/d(mosaic) (ReLU-masked) in one launch + a fixed-order reduce
    (dd_strip6_wgrad).  -> ([dw] * 6, [db] * 6) in the order bl, fl, b, f, br, fr."""
    table, b, h, w, dev, u8, _keep = _sample_table(views, "strip6_wgrad")
    th, tw = (h - 1) // 3 + 1, (w - 50) // 2 + 1
    _chk(g, "g")
    if tuple(g.shape) != (b, 3 * th, 2 * tw, 32):
        raise _lib.HotpathError(f"strip6_wgrad: g {tuple(g.shape)} != {(b, 3 * th, 2 * tw, 32)}")
    dws = [torch.empty(s, device=dev, dtype=torch.float32) for s in _STRIP_SHAPES]
    dbs = [torch.empty(32, device=dev, dtype=torch.float32) for _ in range(6)]
    nbytes = _lib.lib().dd_strip6_wgrad_workspace_bytes()
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    check(_lib.lib().dd_strip6_wgrad(table, u8, _p(g), _ptr6(dws, "dw"), _ptr6(dbs, "db"), b, h, w, _p(ws), nbytes, _stream()), "dd_strip6_wgrad")
    return dws, dbs


def add(a, b):
    out = torch.empty_like(a)
    check(_lib.lib().dd_add(_p(_chk(a, "a")), _p(_chk(b, "b")), _p(out), a.numel(), _stream()), "dd_add")
    return out
