"""Autograd nodes for the decoder conv stack and the spatial bounding-box heads, built on the generic NHWC
convolution kernels (gconv.py).  Each node runs a whole reference module as one hand-ordered chain, so every
ReLU backward is fused into a neighbouring kernel's epilogue, tiling / concat are done by addressing, and all
intermediates stay NHWC.

  DecoderConvStack   reference Decoder.forward conv part          src/autoencoder/components.py:88-92
  SpatialMapFn       reference SpatialMappingCNN.forward          spatial_bb/components.py:28-77
  MergeFn            reference RoadMapBoxesMergingCNN / BoxesMergingCNN.forward   spatial_bb/components.py:141-170, 95-119
"""
import ctypes as C

import os

import torch

from . import _lib, ops
from ._lib import check
from .gconv import EPI_BIAS, EPI_BIAS_RELU, Layer, View, _p, _stream, add, copy_channels, view_to_nhwc4
from . import gconv as gconv_mod
from .gconv import split_rows as gconv_split_rows


# Test hook: when set to a dict, the nodes below leave their ReLU outputs (NHWC) in it so that a checker can replay the
# product's ReLU decisions (oracle.branch); None in production.
TRACE = None


def padded_nhwc(v):
    """``v`` [B,H,W,C]: if it is a rectangular window of a dense NHWC buffer [B,Hm,Wm,C] that starts its storage (the interior of a padded
    activation: SpatialMapFn's Winograd out_conv), return (that buffer, off_h, off_w); None otherwise."""
    if v.dim() != 4 or v.is_contiguous():
        return None
    b, h, w, c = v.shape
    sb, sh, sw, sc = v.stride()
    if sc != 1 or sw != c or sh % c or sh < w * c:
        return None
    wm = sh // c
    total = v.untyped_storage().nbytes() // v.element_size()
    if b > 1:
        if sb % sh:
            return None
        hm = sb // sh
    else:
        hm = total // sh
    off = v.storage_offset()
    if off % c or hm < h or total < b * hm * wm * c:
        return None
    off_h, off_w = (off // c) // wm, (off // c) % wm
    if off_h + h > hm or off_w + w > wm:
        return None
    return v.as_strided((b, hm, wm, c), (hm * wm * c, wm * c, c, 1), 0), off_h, off_w


def as_nhwc(t, cstore):
    """NCHW-shaped tensor -> NHWC buffer [B,H,W,cstore]; free when ``t`` already is a channels-last view (dense, or a window of a dense
    NHWC buffer: ``padded_nhwc``)."""
    b, c, h, w = t.shape
    v = t.permute(0, 2, 3, 1)
    if c == cstore and (v.is_contiguous() or padded_nhwc(v) is not None):
        return v
    return ops.ToNHWC.apply(t, cstore)


def _zeros(shape, dev):
    return torch.zeros(shape, device=dev, dtype=torch.float32)


def _empty(shape, dev):
    return torch.empty(shape, device=dev, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ decoder
class DecoderConvStack(torch.autograd.Function):
    """[B, 64*dh*dw] (NCHW-flat, as fc2 emits it) -> [B,3,2dh,2dw]:  dc1 k3 p1 +ReLU, dc2 k3 p1 +ReLU, dc3 k2 s2 +ReLU, dc4 k1."""

    L1 = Layer(64, 32, 3, pad=1, transposed=True)
    L2 = Layer(32, 32, 3, pad=1, transposed=True)
    L3 = Layer(32, 32, 2, stride=2, transposed=True)
    L4 = Layer(32, 3, 1, transposed=True)

    @staticmethod
    def _wino_desc(a1):
        b, h, w, c = a1.shape
        if c != 32 or h < 4 or w < 4:
            return None
        d = ops.conv_desc(b, h, w, 32, 1)
        return d if ops._lib.lib().dd_conv_wino2_packed_floats(ops.C.byref(d)) > 0 else None

    @staticmethod
    def _as_conv(w):
        """ConvTranspose2d weight [Cin,Cout,3,3] <-> the Conv2d weight [Cout,Cin,3,3] of the same map (stride 1): transposed and flipped."""
        return w.permute(1, 0, 2, 3).flip(2, 3).contiguous()

    @staticmethod
    def forward(ctx, h, dh, dw, w1, b1, w2, b2, w3, b3, w4, b4):
        cls = DecoderConvStack
        b, dev = h.shape[0], h.device
        x0 = ops.nchw_to_nhwc(h.contiguous().view(b, 64, dh, dw), 64)
        a1 = _empty((b, dh, dw, 32), dev)
        cls.L1.forward(w1, b1, View(x0), View(a1), EPI_BIAS_RELU)
        ctx.wino = WINO_DC2 and cls._wino_desc(a1) is not None
        if ctx.wino:
            # dc2 = ConvTranspose2d(32 -> 32, k3, padding 1): a padding-1 convolution with the weights transposed and flipped -- exactly the
            # encoder's c2 layer, so it runs on that layer's Winograd F(2x2,3x3) kernels (4/9 of the multiplies)
            d = cls._wino_desc(a1)
            a2, _ = ops.conv_wino2_fwd_bits(a1, ops.conv_wino2_pack(cls._as_conv(w2), d, ops.PACK_FWD), b2, d)
        else:
            a2 = _empty((b, dh, dw, 32), dev)
            cls.L2.forward(w2, b2, View(a1), View(a2), EPI_BIAS_RELU)
        if dw >= 32:      # dc3 in one launch (four phases = four column tiles), dc4 straight to the NCHW output
            a3 = ops.deconv2x2_c32_fwd(a2, w3.contiguous(), b3, relu=True)
            y = ops.conv1x1_c32_c3_nchw(a3, w4.contiguous(), b4)
        else:
            a3, y4 = _empty((b, 2 * dh, 2 * dw, 32), dev), _zeros((b, 2 * dh, 2 * dw, 4), dev)
            cls.L3.forward(w3, b3, View(a2), View(a3), EPI_BIAS_RELU)
            cls.L4.forward(w4, b4, View(a3), View(y4, 0, 3), EPI_BIAS)
            y = ops.nhwc_to_nchw(y4, 3)
        if TRACE is not None:
            TRACE.update(dc1=a1, dc2=a2, dc3=a3)
        ctx.save_for_backward(x0, a1, a2, a3, w1, w2, w3, w4)
        ctx.dims = (dh, dw)
        return y

    @staticmethod
    def backward(ctx, gy):
        cls = DecoderConvStack
        x0, a1, a2, a3, w1, w2, w3, w4 = ctx.saved_tensors
        b, dev = x0.shape[0], x0.device
        g4 = ops.nchw_to_nhwc(gy.contiguous(), 4)
        dw4, db4 = cls.L4.backward_weight(View(a3), View(g4, 0, 3))
        g3 = _empty(a3.shape, dev)
        cls.L4.backward_data(w4, View(g4), View(g3), relu_src=a3)
        dw3, db3 = cls.L3.backward_weight(View(a2), View(g3))
        g2 = _empty(a2.shape, dev)
        cls.L3.backward_data(w3, View(g3), View(g2), relu_src=a2)
        if ctx.wino:
            d = cls._wino_desc(a1)
            dwc, db2 = ops.conv_wino2_wgrad(a1, g2, d)
            dw2 = cls._as_conv(dwc)                          # the map is its own inverse
            g1 = ops.conv_wino2_dgrad_bits(g2, ops.conv_wino2_pack(cls._as_conv(w2), d, ops.PACK_DGRAD_S1), ops.relu_sign_bits(a1), d)
        else:
            dw2, db2 = cls.L2.backward_weight(View(a1), View(g2))
            g1 = _empty(a1.shape, dev)
            cls.L2.backward_data(w2, View(g2), View(g1), relu_src=a1)
        dw1, db1 = cls.L1.backward_weight(View(x0), View(g1))
        gh = None
        if ctx.needs_input_grad[0]:
            g0 = _empty(x0.shape, dev)
            cls.L1.backward_data(w1, View(g1), View(g0))
            gh = ops.nhwc_to_nchw(g0, 64).view(b, -1)
        return gh, None, None, dw1, db1, dw2, db2, dw3, db3, dw4, db4


# ------------------------------------------------------------------------------------------------ spatial map
# (attribute, view index, transform, tile row, tile col): mosaic  BL FL / B F / BR FR  (components.py:9-13,70-73)
_TILES = (("bl_conv", 3, 0, 0, 0), ("fl_conv", 0, 0, 0, 1), ("b_conv", 4, 1, 1, 0), ("f_conv", 1, 2, 1, 1),
          ("br_conv", 5, 3, 2, 0), ("fr_conv", 2, 3, 2, 1))
_ORDER = ("f_conv", "fl_conv", "fr_conv", "b_conv", "bl_conv", "br_conv", "out_conv")   # parameter order of the module




WINO_OUT = True      # SpatialMapFn: out_conv on the Winograd kernels of the encoder's c2 layer (tests run both positions)
WINO_DC2 = True      # DecoderConvStack: dc2 likewise
WINO_RM2 = True      # MergeFn: rm_conv_2 (dilation 3) as nine plain 3x3 convolutions on phase images, on the same kernels


class SpatialMapFn(torch.autograd.Function):
    """views [B,6,3,H,W] -> spatial map [B,256,256,32] (NHWC).  The six strip convs write their tile of the
    258x258 mosaic directly; rot90 / flip happen in the one pass that lays a view out as NHWC4."""

    SIDE = Layer(3, 32, (1, 50), stride=(3, 2))
    FRONT = Layer(3, 32, (52, 1), stride=(3, 2), pad=1)
    OUT = Layer(32, 32, 3)

    @staticmethod
    def _strip(name):
        return SpatialMapFn.FRONT if name in ("f_conv", "b_conv") else SpatialMapFn.SIDE

    @staticmethod
    def _wino_desc(mosaic):
        """The c2-layer descriptor under which the Winograd kernels take out_conv on this mosaic, or None."""
        b, mh, mw, c = mosaic.shape
        if c != 32 or not mosaic.is_contiguous() or mh < 4 or mw < 4:
            return None
        d = ops.conv_desc(b, mh, mw, 32, 1)
        return d if ops._lib.lib().dd_conv_wino2_packed_floats(ops.C.byref(d)) > 0 else None

    @staticmethod
    def forward(ctx, views, *params):
        cls = SpatialMapFn
        p = {n: (params[2 * i], params[2 * i + 1]) for i, n in enumerate(_ORDER)}
        per_sample = isinstance(views, (tuple, list))      # the collate's tuple of [6,3,H,W] tensors: read through a pointer table
        b, dev = (len(views), views[0].device) if per_sample else (views.shape[0], views.device)
        h_in, w_in = (views[0].shape[1:3] if views[0].dtype == torch.uint8 else views[0].shape[2:4]) if per_sample else \
            (views.shape[2:4] if views.dtype == torch.uint8 else views.shape[3:5])
        fused = gconv_mod.strip6_supported(h_in, w_in)
        th = tw = None
        mosaic = mosaic_bits = None
        laid = []                                           # generic path: the six NHWC4 layouts, kept for the weight gradients
        if fused:
            # all six strip convs in ONE launch reading the views where they lie: rot90 / flip / mosaic placement are index arithmetic
            # in the kernel (csrc/strip6.hip), no re-laid copies at all
            mosaic = gconv_mod.strip6_fwd(views, [p[n][0] for n, *_ in _TILES], [p[n][1] for n, *_ in _TILES], want_bits=WINO_OUT)
            if WINO_OUT:      # the mosaic's sign words from the strip kernels' own epilogue: the ReLU mask of out_conv's data gradient below
                mosaic, mosaic_bits = mosaic
            th, tw = mosaic.shape[1] // 3, mosaic.shape[2] // 2
        for name, vi, tf, tr, tc in (() if fused else _TILES):
            xv = view_to_nhwc4(views, vi, tf)
            laid.append(xv)
            oh, ow = cls._strip(name).out_hw(xv.shape[1], xv.shape[2])
            if mosaic is None:
                th, tw = oh, ow
                mosaic = _empty((b, 3 * th, 2 * tw, 32), dev)
            assert (oh, ow) == (th, tw), "the six strip convs must produce equal tiles"
            cls._strip(name).forward(p[name][0], p[name][1], View(xv), View(mosaic, 0, 32, tr * th, tc * tw, th, tw), EPI_BIAS_RELU)
        oh, ow = cls.OUT.out_hw(3 * th, 2 * tw)
        ctx.wino = WINO_OUT and cls._wino_desc(mosaic) is not None
        if ctx.wino:
            # out_conv (32 -> 32, k3, padding 0) on the c2 layer's Winograd F(2x2,3x3) kernels: they compute the padding-1 convolution of the
            # mosaic, whose interior is out_conv's output -- returned as a view, no crop pass; 4/9 of the multiplies (bs 32: forward 0.45 ->
            # 0.22 ms, data gradient 0.51 -> 0.24, weight gradient 0.55 -> 0.26; tools/bench_outconv.py).  The backward runs from the sign
            # words of this output; the 268 MB output itself is not kept.
            d = cls._wino_desc(mosaic)
            full, keep = ops.conv_wino2_fwd_bits(mosaic, ops.conv_wino2_pack(p["out_conv"][0], d, ops.PACK_FWD), p["out_conv"][1], d)
            out = full[:, 1:1 + oh, 1:1 + ow, :]
        else:
            out = _empty((b, oh, ow, 32), dev)
            cls.OUT.forward(p["out_conv"][0], p["out_conv"][1], View(mosaic), View(out), EPI_BIAS_RELU)
            keep = out
        if TRACE is not None:
            TRACE.update(mosaic=mosaic, space_out=out, tile=(th, tw))
        if ctx.wino and mosaic_bits is None:
            mosaic_bits = ops.relu_sign_bits(mosaic)        # the generic strip path does not write them
        ctx.mosaic_bits = mosaic_bits if ctx.wino else None  # no gradient flows to it: a plain reference
        if per_sample:
            ctx.samples = tuple(views)                      # inputs without gradients: plain references keep them alive
            ctx.save_for_backward(mosaic, keep, p["out_conv"][0])
        else:
            ctx.samples = None
            ctx.save_for_backward(views, mosaic, keep, p["out_conv"][0])
        ctx.tile = (th, tw)
        ctx.fused = fused
        ctx.laid = None if fused else laid                   # generic path: the six layouts are kept for the weight gradients (inputs without gradients: plain references)
        return out

    @staticmethod
    def backward(ctx, gout):
        cls = SpatialMapFn
        if ctx.samples is not None:
            views, (mosaic, out, w_out) = ctx.samples, ctx.saved_tensors
        else:
            views, mosaic, out, w_out = ctx.saved_tensors
        th, tw = ctx.tile
        grads = {}
        if ctx.wino:      # `out` holds the sign words of the padded output here
            d = cls._wino_desc(mosaic)
            g = ops.relu_bwd_pad_bits(gout, out)      # dL/d(out_conv output) behind its ReLU, zero on the border ring the layer does not have (gout may be a channel slice of the merging head's concat gradient: read in place)
            grads["out_conv"] = ops.conv_wino2_wgrad(mosaic, g, d)
            gm = ops.conv_wino2_dgrad_bits(g, ops.conv_wino2_pack(w_out, d, ops.PACK_DGRAD_S1), ctx.mosaic_bits, d)
        else:
            g = ops.relu_bwd(gout.contiguous(), out)
            grads["out_conv"] = cls.OUT.backward_weight(View(mosaic), View(g))
            gm = _empty(mosaic.shape, mosaic.device)
            cls.OUT.backward_data(w_out, View(g), View(gm), relu_src=mosaic)
        if ctx.fused:                                       # six weight + bias gradients: one launch and a fixed-order reduce (csrc/strip6.hip)
            dws, dbs = gconv_mod.strip6_wgrad(views, gm)
            for k, (name, *_rest) in enumerate(_TILES):
                grads[name] = (dws[k], dbs[k])
        for k, (name, vi, tf, tr, tc) in enumerate(() if ctx.fused else _TILES):
            xv = ctx.laid[k] if ctx.laid is not None else view_to_nhwc4(views, vi, tf)      # kept from the forward (six launches of 15-55 us otherwise)
            grads[name] = cls._strip(name).backward_weight(View(xv), View(gm, 0, 32, tr * th, tc * tw, th, tw))
        flat = []
        for n in _ORDER:
            flat += list(grads[n])
        return (None, *flat)


def road_map_taps(rm):
    """rm [B,1,H,W] (or the collate's tuple of B bool / uint8 [H,W] masks, read where they lie) -> the NHWC4 image of the pixels
    rm_conv_1 (k7, stride 3, dilation 3, padding 1: components.py:80) reads, (3u - 1, 3v - 1) for u, v in [0, out + 6): on it the
    layer is a dense 7x7 convolution with the same products in the same order, and neither the forward nor the weight gradient
    gathers from a 16-byte-per-pixel copy of the full mask."""
    if isinstance(rm, (tuple, list)):
        oh, ow = MergeFn.RM1.out_hw(rm[0].shape[-2], rm[0].shape[-1])
        return ops.subsample_masks_nhwc4(rm, 3, -1, oh + 6, ow + 6)
    oh, ow = MergeFn.RM1.out_hw(rm.shape[-2], rm.shape[-1])
    return ops.subsample_nhwc4(rm, 3, -1, oh + 6, ow + 6)


# ------------------------------------------------------------------------------------------------ merging heads
class MergeFn(torch.autograd.Function):
    """ssr [B,128,918,32], spatial map [B,256,256,32] (both NHWC) and, for the road-map variant, rm4 [B,268,268,4]
    (``road_map_taps``: the pixels (3u - 1, 3v - 1) of the 1-channel mask that rm_conv_1 reads, NHWC4) -> box-mask
    probabilities [B,800,800].

    The channel concat (components.py:109,159) is a 64- or 96-channel NHWC buffer that ss_deconv and rm_conv_2
    write by channel slice; up_conv_N run as flipped-tap dilated gathers; the last ConvTranspose2d(8->1,k2,s2)+sigmoid
    has its own VALU kernel."""

    SS_CONV = Layer(32, 32, (1, 24), stride=(1, 7))
    SS_DECONV = Layer(32, 32, 2, stride=2, transposed=True)
    RM1 = Layer(1, 32, 7, stride=3, dil=3, pad=1)       # rm_conv_1 as the reference states it (shapes) ...
    RM1S = Layer(1, 32, 7)                              # ... and as a dense layer on the road map's pixels (3u - 1, 3v - 1): the generic-engine
                                                        # form of what ops.conv1ch_* run (kept as the cross-check of tests/test_gpu_gconv.py)
    RM2 = Layer(32, 32, 3, dil=3)
    UPS_RM = (Layer(96, 64, 7, dil=7, transposed=True), Layer(64, 32, 7, dil=7, transposed=True),
              Layer(32, 16, 7, dil=7, transposed=True), Layer(16, 8, 7, dil=3, transposed=True))
    UPS_PLAIN = (Layer(64, 32, 8, dil=8, transposed=True), Layer(32, 16, 8, dil=8, transposed=True),
                 Layer(16, 8, 6, dil=6, transposed=True, output_padding=2))

    @staticmethod
    def _rm2_wino_desc(b, rh, rw):
        """The c2-layer descriptor of the 9 b phase images of rm_conv_2's rh x rw input, or None when the Winograd kernels do not take them."""
        ph, pw = (rh + 2) // 3, (rw + 2) // 3
        if ph < 4 or pw < 4:
            return None
        d = ops.conv_desc(9 * b, ph, pw, 32, 1)
        return d if ops._lib.lib().dd_conv_wino2_packed_floats(ops.C.byref(d)) > 0 else None

    @staticmethod
    def _dense_from_phase3(p, oh, ow):
        """[9B, ph, pw, 32] phase-major -> [B, oh, ow, 32] (tests / TRACE only)."""
        b = p.shape[0] // 9
        out = torch.empty((b, oh, ow, 32), device=p.device, dtype=p.dtype)
        pv = p.view(b, 3, 3, p.shape[1], p.shape[2], 32)
        for a in range(3):
            for c in range(3):
                na, nc = (oh - a + 2) // 3, (ow - c + 2) // 3
                out[:, a::3, c::3] = pv[:, a, c, :na, :nc]
        return out

    @staticmethod
    def forward(ctx, ssr, space, rm4, with_rm, *params):
        cls = MergeFn
        ups = cls.UPS_RM if with_rm else cls.UPS_PLAIN
        it = iter(params)
        nxt = lambda: (next(it), next(it))          # noqa: E731
        p_ssc, p_ssd = nxt(), nxt()
        p_rm1, p_rm2 = (nxt(), nxt()) if with_rm else (None, None)
        p_up = [nxt() for _ in ups]
        p_last = nxt()
        b, dev = ssr.shape[0], ssr.device
        sh, sw = cls.SS_CONV.out_hw(ssr.shape[1], ssr.shape[2])
        s1 = _empty((b, sh, sw, 32), dev)
        cls.SS_CONV.forward(p_ssc[0], p_ssc[1], View(ssr), View(s1), EPI_BIAS_RELU)
        ch, cw = 2 * sh, 2 * sw
        assert (ch, cw) == tuple(space.shape[1:3]), "ssr and spatial map must meet at the same size"
        cat = _empty((b, ch, cw, 96 if with_rm else 64), dev)
        if s1.shape[2] >= 32:      # all four phases in one launch, straight into the concat slice
            ops.deconv2x2_c32_fwd_into(s1, p_ssd[0].contiguous(), p_ssd[1], cat, 0, relu=True)
        else:
            cls.SS_DECONV.forward(p_ssd[0], p_ssd[1], View(s1), View(cat, 0, 32), EPI_BIAS_RELU)
        base = padded_nhwc(space)
        if base is None:
            copy_channels(View(space.contiguous()), View(cat, 32, 32))
        else:      # the interior of a padded activation (SpatialMapFn's Winograd out_conv): copied out of its window, no dense copy first
            copy_channels(View(base[0], 0, 32, base[1], base[2], space.shape[1], space.shape[2]), View(cat, 32, 32))
        r1 = r1_bits = None
        ctx.wino_rm2 = False
        if with_rm:
            rh, rw = rm4.shape[1] - 6, rm4.shape[2] - 6
            assert cls.RM2.out_hw(rh, rw) == (ch, cw)
            d9 = cls._rm2_wino_desc(b, rh, rw) if WINO_RM2 else None
            if d9 is not None:
                # rm_conv_2 (32 -> 32, k3, dilation 3) is nine plain 3x3 convolutions on the residue classes of its input: rm_conv_1 writes
                # them as nine images per scene (phase-major, csrc/conv1ch.hip), the c2 layer's Winograd kernels run the padding-1
                # convolution of all 9 B of them, and the interiors are scattered into the concat slice (1.36 -> ~0.95 ms at bs 32)
                ctx.wino_rm2 = True
                r1, r1_bits = ops.conv1ch_fwd_phase3(rm4, p_rm1[0], p_rm1[1], relu=True)
                y9, _ = ops.conv_wino2_fwd_bits(r1, ops.conv_wino2_pack(p_rm2[0], d9, ops.PACK_FWD), p_rm2[1], d9)
                ops.phase3_scatter(y9, cat, 64, 1)
                del y9
            else:
                r1 = ops.conv1ch_fwd(rm4, p_rm1[0], p_rm1[1], relu=True)      # rm_conv_1: taps as the K dimension (csrc/conv1ch.hip)
                cls.RM2.forward(p_rm2[0], p_rm2[1], View(r1), View(cat, 64, 32), EPI_BIAS_RELU)
        acts = [cat]
        split_x = []                                 # split-product experiment (gconv.SPLIT_BF16): each layer input's bf16 planes, kept for its weight gradient
        planes = None                                # ... and the planes of the previous layer's OUTPUT, written by its epilogue (no split pass)
        for li, (layer, (w, bias)) in enumerate(zip(ups, p_up)):
            src = acts[-1]
            oh, ow = layer.out_hw(src.shape[1], src.shape[2])
            dst = _empty((b, oh, ow, layer.cout), dev)
            nxt_up = ups[li + 1] if li + 1 < len(ups) else None
            # the output's planes are worth their stores only if the next layer's forward runs on the split kernels too (k7 d7, Cout > 16)
            keep, emit = {}, ({} if nxt_up is not None and nxt_up.k == (7, 7) and nxt_up.dil == (7, 7) and nxt_up.cout > 16 else None)
            layer.forward(w, bias, View(src), View(dst), EPI_BIAS_RELU, keep=keep, xs=planes, emit=emit)
            split_x.append(keep.get("xs"))
            planes = (emit or {}).get("ys")
            acts.append(dst)
        ctx.split_x = split_x
        ctx.split_mode = gconv_mod.SPLIT_BF16                # the backward runs in the precision mode of its forward (gconv.split_products)
        u = acts[-1]
        probs = _empty((b, 2 * u.shape[1], 2 * u.shape[2]), dev)
        check(_lib.lib().dd_deconv2x2_c1_fwd(_p(u), _p(p_last[0]), _p(p_last[1]), _p(probs), b, u.shape[1], u.shape[2], 8,
                                             _stream()), "dd_deconv2x2_c1_fwd")
        if TRACE is not None:
            TRACE.update(s1=s1, cat=cat, r1=cls._dense_from_phase3(r1, rh, rw) if ctx.wino_rm2 else r1, acts=acts)
        ctx.with_rm = with_rm
        ctx.save_for_backward(ssr, s1, rm4 if with_rm else None, r1, probs, p_ssc[0], p_ssd[0],
                              p_rm2[0] if with_rm else None, p_last[0], *[w for w, _ in p_up], *acts)
        ctx.nup = len(ups)
        ctx.r1_bits = r1_bits      # sign words of the phase-major r1 (no gradient flows to them: a plain reference)
        return probs

    @staticmethod
    def backward(ctx, gprobs):
        with gconv_mod.split_products(ctx.split_mode):
            return MergeFn._backward(ctx, gprobs)

    @staticmethod
    def _backward(ctx, gprobs):
        cls = MergeFn
        with_rm, nup = ctx.with_rm, ctx.nup
        ups = cls.UPS_RM if with_rm else cls.UPS_PLAIN
        sv = ctx.saved_tensors
        ssr, s1, rm4, r1, probs, w_ssc, w_ssd, w_rm2, w_last = sv[:9]
        w_up, acts = sv[9:9 + nup], sv[9 + nup:]
        cat, u = acts[0], acts[-1]
        b, dev = ssr.shape[0], ssr.device
        # last layer: sigmoid' and the ReLU mask of u are applied inside
        gu = _empty(u.shape, dev)
        dw_last, db_last = torch.empty_like(w_last), _empty((1,), dev)
        ws = torch.empty(_lib.lib().dd_deconv2x2_c1_workspace_bytes(8), device=dev, dtype=torch.uint8)
        check(_lib.lib().dd_deconv2x2_c1_bwd(_p(u), _p(w_last), _p(probs), _p(gprobs.contiguous()), _p(gu), _p(dw_last), _p(db_last),
                                             b, u.shape[1], u.shape[2], 8, _p(ws), _stream()), "dd_deconv2x2_c1_bwd")
        g = gu
        g_planes = None                              # split-product experiment: the bf16 planes of g, when the kernel that produced g wrote them
        db_from_above = None                         # this layer's bias gradient, when the data gradient above already summed dL/dy
        g_up = []
        for i in range(nup - 1, -1, -1):
            layer, src = ups[i], acts[i]
            # split-product experiment: dL/dy's bf16 planes exist ONCE for the layer's weight and data gradient -- written by the
            # epilogue of the data gradient above it, or by one split pass here; the input's planes come from the forward
            gs = g_planes
            if gs is None and layer.split_wgrad_ok(View(src), View(g)):
                gs = gconv_split_rows(View(g))
            use_w = gs is not None and layer.split_wgrad_ok(View(src), View(g))
            dw_i, db_i = layer.backward_weight(View(src), View(g), xs=ctx.split_x[i] if use_w else None, gs=gs if use_w else None,
                                               want_bias=db_from_above is None)
            g_up.append((dw_i, db_i if db_from_above is None else db_from_above))
            gsrc = _empty(src.shape, dev)
            emit = {} if i > 0 else None             # the concat buffer's gradient (i == 0) goes to layers outside the experiment
            # masks with the producer's ReLU output; in the concat buffer (i == 0) only the slices this node's own ReLUs
            # wrote (ss_deconv 0:32, rm_conv_2 64:96): the spatial-map slice 32:64 is an external input
            # the per-channel sums of gsrc are the bias gradient of the up-conv below: the kernel that writes gsrc leaves them where it can
            csum = _empty((ups[i - 1].cout,), dev) if i > 0 and gsrc.shape[3] == ups[i - 1].cout else None
            took = layer.backward_data(w_up[i], View(g), View(gsrc), relu_src=src, mask_pass=(32, 64) if i == 0 else (0, 0), gs=gs, emit=emit,
                                       colsum=csum)
            db_from_above = csum if took else None
            g, g_planes = gsrc, (emit or {}).get("ys")
        g_up.reverse()
        gcat = g                                                                 # [B,256,256,64|96]; slices 0:32 / 64:96 ReLU-masked
        g_space = None
        if ctx.needs_input_grad[1]:                           # plain dL/d(spatial_map): no mask
            if WINO_OUT:      # handed on as the slice of the concat gradient it is: SpatialMapFn's Winograd backward reads it where it lies
                g_space = gcat[..., 32:64]
            else:
                g_space = _empty(gcat.shape[:3] + (32,), dev)
                copy_channels(View(gcat, 32, 32), View(g_space))
        g_rm2 = g_rm1 = (None, None)
        if with_rm:
            if ctx.wino_rm2:
                d9 = cls._rm2_wino_desc(gcat.shape[0], rm4.shape[1] - 6, rm4.shape[2] - 6)
                g9 = ops.phase3_gather(gcat, 64, r1.shape[1], r1.shape[2], 1)      # zero border ring and padding cells included
                g_rm2 = ops.conv_wino2_wgrad(r1, g9, d9)
                gr1 = ops.conv_wino2_dgrad_bits(g9, ops.conv_wino2_pack(w_rm2, d9, ops.PACK_DGRAD_S1), ctx.r1_bits, d9)
                g_rm1 = ops.conv1ch_wgrad_phase3(rm4, gr1)
            else:
                g_rm2 = cls.RM2.backward_weight(View(r1), View(gcat, 64, 32))
                gr1 = _empty(r1.shape, dev)
                cls.RM2.backward_data(w_rm2, View(gcat, 64, 32), View(gr1), relu_src=r1)
                g_rm1 = ops.conv1ch_wgrad(rm4, gr1)
        g_ssd = cls.SS_DECONV.backward_weight(View(s1), View(gcat, 0, 32))
        gs1 = _empty(s1.shape, dev)
        cls.SS_DECONV.backward_data(w_ssd, View(gcat, 0, 32), View(gs1), relu_src=s1)
        g_ssc = cls.SS_CONV.backward_weight(View(ssr), View(gs1))
        g_ssr = None
        if ctx.needs_input_grad[0]:
            g_ssr = _empty(ssr.shape, dev)
            cls.SS_CONV.backward_data(w_ssc, View(gs1), View(g_ssr))     # plain dL/d(ssr): the encoder's ReLU backward is the encoder's
        flat = [*g_ssc, *g_ssd]
        if with_rm:
            flat += [*g_rm1, *g_rm2]
        for gw in g_up:
            flat += list(gw)
        flat += [dw_last, db_last]
        return (g_ssr, g_space, None, None, *flat)
