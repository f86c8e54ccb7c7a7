"""Closed-form synthetic tensors: the same numbers on any machine, no RNG state, no files.

Used by the parity tests, the golden-fixture generator and ``bench.py`` so that the GPU box
(which never sees the reference) can rebuild bit-identical inputs and weights.
"""
import zlib

import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """murmur3 finaliser on uint64 arrays holding 32-bit values."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_uniform(shape, salt, lo=-0.5, hi=0.5, dtype=torch.float32):
    """Tensor of ``shape`` with element i = lo + (hi-lo) * mix32(i*0x9E3779B1 + salt) / 2^32."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    u = _mix32(i * np.uint64(0x9E3779B1) + np.uint64(salt & 0xFFFFFFFF))
    v = lo + (hi - lo) * (u.astype(np.float64) / 4294967296.0)
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def key_salt(name, seed=0):
    return (zlib.crc32(name.encode()) + 0x9E3779B9 * seed) & 0xFFFFFFFF


@torch.no_grad()
def fill_module(module, seed=0, gain=1.0):
    """Overwrite every parameter/buffer of ``module`` with hash values scaled like a default init.

    weights (dim >= 2): uniform(+-gain*sqrt(3/fan_in)); biases: uniform(+-0.1);
    BN weight: 1 + uniform(+-0.25); BN bias / running_mean: uniform(+-0.1); running_var: 1 + uniform(0, 0.5).
    ConvTranspose2d weights are [Cin, Cout, kh, kw]: fan_in is taken as Cin*kh*kw.
    """
    transposed = {id(p) for m in module.modules() if isinstance(m, torch.nn.ConvTranspose2d) for p in [m.weight]}
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        s = key_salt(name, seed)
        if name.endswith("num_batches_tracked"):
            continue
        if name.endswith("running_var"):
            v = 1.0 + hash_uniform(t.shape, s, 0.0, 0.5)
        elif name.endswith("running_mean"):
            v = hash_uniform(t.shape, s, -0.1, 0.1)
        elif name.endswith("weight") and len(name.split(".")) > 1 and "bn" in name.split(".")[-2]:
            v = 1.0 + hash_uniform(t.shape, s, -0.25, 0.25)
        elif t.dim() >= 2:
            if id(t) in transposed:
                fan_in = t.shape[0] * int(np.prod(t.shape[2:]))
            else:
                fan_in = int(np.prod(t.shape[1:]))
            b = gain * (3.0 / fan_in) ** 0.5
            v = hash_uniform(t.shape, s, -b, b)
        else:
            v = hash_uniform(t.shape, s, -0.1, 0.1)
        t.copy_(v.to(t.dtype))
    return module


def camera_batch(batch, height=256, width=306, seed=1, views=6, channels=3):
    """[B,6,3,H,W] images in [0,1), as ``ToTensor`` would deliver them (reference autoencoder.py:133)."""
    return hash_uniform((batch, views, channels, height, width), key_salt("camera", seed), 0.0, 1.0)


def road_maps(batch, seed=1, size=800, density=0.3):
    """[B,size,size] bool road masks (reference data_helper.py road_image)."""
    return hash_uniform((batch, size, size), key_salt("road", seed), 0.0, 1.0) < density


def car_boxes(n_boxes, seed=1):
    """[n,2,4] float64 car-like rotated rectangles in metres (corner order fl, fr, bl, br; rows x then y), as the
    dataset's ``target['bounding_box']`` holds them (reference data_helper.py:118-129).  Half are nearly axis-aligned
    (lane traffic), some poke out of the +-40 m map."""
    u = hash_uniform((n_boxes, 6), key_salt("cars", seed), 0.0, 1.0, dtype=torch.float64).numpy()
    cx, cy = u[:, 0] * 84.0 - 42.0, u[:, 1] * 84.0 - 42.0
    length, width = 3.5 + 2.5 * u[:, 2], 1.6 + 0.8 * u[:, 3]
    theta = np.where(u[:, 4] < 0.5, np.round(u[:, 5] * 4.0) * (np.pi / 2) + (u[:, 4] - 0.25) * 0.2, u[:, 5] * 2.0 * np.pi)
    c, s = np.cos(theta), np.sin(theta)
    local = np.array([[0.5, 0.5], [0.5, -0.5], [-0.5, 0.5], [-0.5, -0.5]])              # fl fr bl br
    lx, ly = local[None, :, 0] * length[:, None], local[None, :, 1] * width[:, None]
    xs = cx[:, None] + lx * c[:, None] - ly * s[:, None]
    ys = cy[:, None] + lx * s[:, None] + ly * c[:, None]
    return torch.from_numpy(np.stack([xs, ys], axis=1))


def wild_quads(n_boxes, seed=1):
    """[n,2,4] float64 arbitrary quadrilaterals (self-intersecting, slivers, far outside the map): rasteriser edge cases."""
    u = hash_uniform((n_boxes, 2, 4), key_salt("quads", seed), 0.0, 1.0, dtype=torch.float64).numpy()
    big = u * 96.0 - 48.0
    c = hash_uniform((n_boxes, 2, 1), key_salt("quadc", seed), -41.0, 41.0, dtype=torch.float64).numpy()
    small = c + (u - 0.5) * 3.0
    pick = (np.arange(n_boxes) % 3 == 0)[:, None, None]
    return torch.from_numpy(np.where(pick, big, small))
