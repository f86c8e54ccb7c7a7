// Layout glue of the hot path, fused with the NCHW<->NHWC conversion the MFMA kernels want:
//   dd_stitch6        6-view reorder + wide stitch (+ masked-view task)   reference roadmap_bce_v2.py:53-64, autoencoder.py:53-73
//   dd_nchw_to_nhwc / dd_nhwc_to_nchw                                     API edges of Encoder.forward (components.py:40)
//   dd_pool4_*        max_pool1d(4) over the NCHW-flattened c3 feature    components.py:46-47
// All of it is HBM-bound byte shuffling: every kernel reads and writes each byte once, 16 bytes per lane
// on the NHWC side.
#include "dd_common.h"

namespace {

__constant__ int kViewOrder[6] = {0, 1, 2, 5, 4, 3};

// one thread per wide-image pixel: reads 3 planes (coalesced along x), writes one 16-byte NHWC4 pixel
__global__ __launch_bounds__(256) void stitch6_kernel(const float* __restrict__ views, f32x4* __restrict__ wide4,
                                                      float* __restrict__ wide_nchw, float* __restrict__ target,
                                                      int B, int H, int W, int mask_slot) {
  const long npx = (long)B * H * 6 * W;
  const long plane = (long)H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(p % (6 * W));
    const int yy = (int)((p / (6 * W)) % H);
    const int b = (int)(p / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const float* src = views + (((long)b * 6 + kViewOrder[slot]) * 3) * plane + (long)yy * W + xx;
    float c0 = src[0], c1 = src[plane], c2 = src[2 * plane];
    if (slot == mask_slot) {
      if (target) {
        float* t = target + ((long)b * 3) * plane + (long)yy * W + xx;
        t[0] = c0; t[plane] = c1; t[2 * plane] = c2;
      }
      c0 = c1 = c2 = 0.f;
    }
    if (wide4) wide4[p] = f32x4{c0, c1, c2, 0.f};
    if (wide_nchw) {
      float* o = wide_nchw + ((long)b * 3) * H * 6 * W + (long)yy * 6 * W + xw;
      o[0] = c0; o[(long)H * 6 * W] = c1; o[2L * H * 6 * W] = c2;
    }
  }
}

// The reference hands the batch over as a TUPLE of per-sample tensors (collate_fn = tuple(zip(*batch)), helper.py:22-23)
// and stacks them first (roadmap_bce_v2.py:55).  Reading through a table of per-sample base pointers skips that copy.
struct SamplePtrs {
  const float* p[64];      // passed by value in the kernel arguments: no device-side table, no extra copy
};

__global__ __launch_bounds__(256) void stitch6_ptrs_kernel(const SamplePtrs samples, f32x4* __restrict__ wide4,
                                                           int B, int H, int W) {
  const long npx = (long)B * H * 6 * W;
  const long plane = (long)H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(p % (6 * W));
    const int yy = (int)((p / (6 * W)) % H);
    const int b = (int)(p / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const float* src = samples.p[b] + ((long)kViewOrder[slot] * 3) * plane + (long)yy * W + xx;
    wide4[p] = f32x4{src[0], src[plane], src[2 * plane], 0.f};
  }
}

// uint8 HWC camera frames (what a JPEG decoder emits) -> the same wide NHWC4 fp32 image: ToTensor's /255
// (reference autoencoder.py:133 torchvision.transforms.ToTensor) fused with the gather; 3 bytes in, 16 bytes out.
__global__ __launch_bounds__(256) void stitch6_u8_kernel(const unsigned char* __restrict__ frames, f32x4* __restrict__ wide4,
                                                         int B, int H, int W) {
  const long npx = (long)B * H * 6 * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(p % (6 * W));
    const int yy = (int)((p / (6 * W)) % H);
    const int b = (int)(p / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const unsigned char* src = frames + ((((long)b * 6 + kViewOrder[slot]) * H + yy) * W + xx) * 3;
    // a true division, correctly rounded: bit for bit ToTensor's img.float().div(255) (x * (1/255.f) differs in the last place for
    // some of the 256 values)
    wide4[p] = f32x4{(float)src[0] / 255.0f, (float)src[1] / 255.0f, (float)src[2] / 255.0f, 0.f};
  }
}

// The same from a table of per-sample base pointers (each [6,H,W,3] uint8: the collate's tuple of decoded frames), with the
// masked-view task of BasicAE.six_to_one_task (autoencoder.py:59-73) optional: view slot `mask_slot` of the wide image is
// blanked and written, as fp32 NCHW, to `target`.
struct SamplePtrsU8 {
  const unsigned char* p[64];
};

__global__ __launch_bounds__(256) void stitch6_u8_ptrs_kernel(const SamplePtrsU8 samples, f32x4* __restrict__ wide4,
                                                              float* __restrict__ target, int B, int H, int W, int mask_slot) {
  const long npx = (long)B * H * 6 * W;
  const long plane = (long)H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(p % (6 * W));
    const int yy = (int)((p / (6 * W)) % H);
    const int b = (int)(p / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const unsigned char* src = samples.p[b] + (((long)kViewOrder[slot] * H + yy) * W + xx) * 3;
    float c0 = (float)src[0] / 255.0f, c1 = (float)src[1] / 255.0f, c2 = (float)src[2] / 255.0f;
    if (slot == mask_slot) {
      if (target) {
        float* t = target + ((long)b * 3) * plane + (long)yy * W + xx;
        t[0] = c0; t[plane] = c1; t[2 * plane] = c2;
      }
      c0 = c1 = c2 = 0.f;
    }
    wide4[p] = f32x4{c0, c1, c2, 0.f};
  }
}

// NCHW -> NHWC(Cs): one thread per (pixel, 4-channel group)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, f32x4* __restrict__ dst,
                                                           int B, int C, int H, int W, int Cs) {
  const int groups = Cs / 4;
  const long plane = (long)H * W;
  const long total = (long)B * plane * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % plane;
    const int g = (int)((i / plane) % groups);
    const int b = (int)(i / (plane * groups));
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * g + k;
      v[k] = (c < C) ? src[((long)b * C + c) * plane + p] : 0.f;
    }
    dst[((long)b * plane + p) * groups + g] = v;
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int B, int C, int H, int W, int Cs) {
  const long plane = (long)H * W;
  const long total = (long)B * C * plane;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % plane;
    const int c = (int)((i / plane) % C);
    const int b = (int)(i / (plane * C));
    dst[i] = src[((long)b * plane + p) * Cs + c];
  }
}

// Tiled form for 8 <= Cs <= 64 (the decoder's 64-channel feature: 160 MB each way per step): a 64-pixel x Cs tile goes
// through LDS so that BOTH sides are read / written as contiguous runs (the one-thread-per-element kernels above leave
// one side at a Cs*4-byte stride: 0.37 ms for what is 0.08 ms of HBM traffic).
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void layout_tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long plane,
                                                          int Cs) {
  __shared__ float t[64][65];
  const int tid = threadIdx.x;
  const long b = blockIdx.y, p0 = (long)blockIdx.x * 64;
  const int npx = (int)min(64L, plane - p0);
  if (TO_NHWC) {
    for (int c = tid >> 6; c < Cs; c += 4) {
      const int px = tid & 63;
      t[c][px] = (c < C && px < npx) ? src[(b * C + c) * plane + p0 + px] : 0.f;
    }
    __syncthreads();
    float* out = dst + (b * plane + p0) * Cs;
    for (int j = tid; j < npx * Cs; j += 256) out[j] = t[j % Cs][j / Cs];
  } else {
    const float* in = src + (b * plane + p0) * Cs;
    for (int j = tid; j < npx * Cs; j += 256) t[j % Cs][j / Cs] = in[j];
    __syncthreads();
    for (int c = tid >> 6; c < C; c += 4) {
      const int px = tid & 63;
      if (px < npx) dst[(b * C + c) * plane + p0 + px] = t[c][px];
    }
  }
}

// ---- pool, fast path: H*W % 4 == 0 and C % 4 == 0, so a window of 4 never leaves its channel plane.
// thread = (quad of 4 consecutive flat pixels, group of 4 channels): 4 x 16-byte loads, 4 outputs.
template <bool IDX>
__global__ __launch_bounds__(256) void pool4_fwd_quad(const f32x4* __restrict__ feat, float* __restrict__ pooled,
                                                      unsigned short* __restrict__ idx, int B, long HW, int C) {
  const int groups = C / 4;
  const long quads = HW / 4;
  const long total = (long)B * quads * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long q = (i / groups) % quads;
    const long b = i / (groups * quads);
    const f32x4* p = feat + ((b * HW + 4 * q) * groups + g);
    const f32x4 v0 = p[0], v1 = p[groups], v2 = p[2 * groups], v3 = p[3 * groups];
    float* o = pooled + b * (quads * C) + q;
    unsigned code = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!IDX) {
        o[(long)(4 * g + k) * quads] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
      } else {      // the backward's routing decided here, 4 bits per window: first maximum (torch keeps the earliest) | (max > 0) << 2
        float m = v0[k];
        unsigned am = 0;
        if (v1[k] > m) { m = v1[k]; am = 1; }
        if (v2[k] > m) { m = v2[k]; am = 2; }
        if (v3[k] > m) { m = v3[k]; am = 3; }
        o[(long)(4 * g + k) * quads] = m;
        code |= (am | (m > 0.f ? 4u : 0u)) << (4 * k);
      }
    }
    if (IDX) idx[i] = (unsigned short)code;
  }
}

// Backward from the routing codes of pool4_fwd_quad<true>: the 481 MB feature is not read again (15 MB of codes instead).
__global__ __launch_bounds__(256) void pool4_bwd_idx_quad(const float* __restrict__ dpooled, const unsigned short* __restrict__ idx,
                                                          f32x4* __restrict__ dfeat, int B, long HW, int C) {
  const int groups = C / 4;
  const long quads = HW / 4;
  const long total = (long)B * quads * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long q = (i / groups) % quads;
    const long b = i / (groups * quads);
    const long base = (b * HW + 4 * q) * groups + g;
    const float* gp = dpooled + b * (quads * C) + q;
    const unsigned code = idx[i];
    f32x4 d0, d1, d2, d3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned c4 = (code >> (4 * k)) & 15u;
      const float gv = (c4 & 4u) ? gp[(long)(4 * g + k) * quads] : 0.f;
      const unsigned am = c4 & 3u;
      d0[k] = am == 0 ? gv : 0.f;
      d1[k] = am == 1 ? gv : 0.f;
      d2[k] = am == 2 ? gv : 0.f;
      d3[k] = am == 3 ? gv : 0.f;
    }
    __builtin_nontemporal_store(d0, dfeat + base);
    __builtin_nontemporal_store(d1, dfeat + base + groups);
    __builtin_nontemporal_store(d2, dfeat + base + 2 * groups);
    __builtin_nontemporal_store(d3, dfeat + base + 3 * groups);
  }
}

// ---- C == 32: the same two kernels in tiles of 64 windows (256 pixels = 32 KB of one image) that go through LDS, so that BOTH sides of the
// NHWC <-> NCHW-flat change of order move in whole lines: above, a wave's stores into a channel plane (forward) and its loads from one
// (backward) are 32-byte pieces.  Same codes, same results bit for bit.
__global__ __launch_bounds__(256) void pool4_fwd_tile32(const f32x4* __restrict__ feat, float* __restrict__ pooled,
                                                        unsigned* __restrict__ idx, long quads, int qblocks) {
  __shared__ float t[32][65];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / qblocks;
  const long q0 = (long)(blockIdx.x - b * qblocks) * 64;
  const int nq = (int)min(64L, quads - q0);
  const int q = tid >> 2, c8 = tid & 3;              // thread = (window, 8 channels = chunks 2 c8 and 2 c8 + 1 of a pixel's 8)
  if (q < nq) {
    const f32x4* src = feat + ((b * quads + q0 + q) * 4) * 8 + 2 * c8;
    f32x4 v[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k][0] = __builtin_nontemporal_load(src + 8 * k);
      v[k][1] = __builtin_nontemporal_load(src + 8 * k + 1);
    }
    unsigned codes = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float m = v[0][j >> 2][j & 3];
      unsigned am = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float f = v[k][j >> 2][j & 3];
        if (f > m) { m = f; am = k; }
      }
      t[8 * c8 + j][q] = m;
      codes |= (am | (m > 0.f ? 4u : 0u)) << (4 * j);
    }
    idx[(b * quads + q0 + q) * 4 + c8] = codes;      // two 16-bit words: channel groups 2 c8 and 2 c8 + 1
  }
  __syncthreads();
  float* o = pooled + b * (quads * 32) + q0;
  const int qq = tid & 63;
  if (qq < nq) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (tid >> 6) + 4 * j;
      o[(long)c * quads + qq] = t[c][qq];
    }
  }
}

__global__ __launch_bounds__(256) void pool4_bwd_tile32(const float* __restrict__ dpooled, const unsigned* __restrict__ idx,
                                                        f32x4* __restrict__ dfeat, long quads, int qblocks) {
  __shared__ float g[32][65];
  __shared__ unsigned cd[64][4];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / qblocks;
  const long q0 = (long)(blockIdx.x - b * qblocks) * 64;
  const int nq = (int)min(64L, quads - q0);
  const float* gp = dpooled + b * (quads * 32) + q0;
  const int qq = tid & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = (tid >> 6) + 4 * j;
    g[c][qq] = qq < nq ? gp[(long)c * quads + qq] : 0.f;
  }
  cd[tid >> 2][tid & 3] = (tid >> 2) < nq ? idx[(b * quads + q0) * 4 + tid] : 0u;
  __syncthreads();
  f32x4* out = dfeat + (b * quads + q0) * 32;       // 32 chunks of 16 bytes per window
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = tid + 256 * k;                    // chunk i of the tile: pixel i >> 3, channels 4 (i & 7) ..
    const int px = i >> 3, c4 = i & 7, qd = px >> 2, pos = px & 3;
    if (qd >= nq) continue;
    const unsigned codes = cd[qd][c4 >> 1] >> (16 * (c4 & 1));
    f32x4 d;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned c = (codes >> (4 * j)) & 15u;
      d[j] = ((c & 4u) && (c & 3u) == (unsigned)pos) ? g[4 * c4 + j][qd] : 0.f;
    }
    __builtin_nontemporal_store(d, out + i);
  }
}

// The c3 feature has TWO consumers in the joint roadmap + box model (the pool in front of fc1, and the box heads): its gradient is
//   dfeat = (feat > 0) * (gfeat + route(dpooled)),   route = dpooled at the first maximum of each window, 0 elsewhere
// -- three passes over 481 MB tensors as relu_bwd + pool4_bwd + add (0.75 ms at bs 32), one here: the gradient tile comes in through
// LDS, thread = (window, 4 channels) reads the window's four pixels of feat and gfeat and writes four whole 128-byte lines with its
// seven neighbours.  Same arithmetic: (feat > 0 ? gfeat : 0) + (routed, if the maximum is positive), one fp32 add per element.
__global__ __launch_bounds__(256) void pool4_bwd_add_tile32(const float* __restrict__ dpooled, const f32x4* __restrict__ feat,
                                                            const f32x4* __restrict__ gfeat, f32x4* __restrict__ dfeat, long quads,
                                                            int qblocks) {
  __shared__ float g[32][65];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / qblocks;
  const long q0 = (long)(blockIdx.x - b * qblocks) * 64;
  const int nq = (int)min(64L, quads - q0);
  const float* gp = dpooled + b * (quads * 32) + q0;
  const int qq = tid & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = (tid >> 6) + 4 * j;
    g[c][qq] = qq < nq ? gp[(long)c * quads + qq] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + 256 * k;                    // (window i >> 3, channel group i & 7) of the tile
    const int qd = i >> 3, c4 = i & 7;
    if (qd >= nq) continue;
    const long base = ((b * quads + q0 + qd) * 4) * 8 + c4;      // pixel 4 (q0 + qd), chunk c4; a pixel is 8 chunks of 16 bytes
    f32x4 v[4], u[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      v[p] = __builtin_nontemporal_load(feat + base + 8 * p);
      u[p] = __builtin_nontemporal_load(gfeat + base + 8 * p);
    }
    f32x4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float m = v[0][j];
      int am = 0;
#pragma unroll
      for (int p = 1; p < 4; ++p)
        if (v[p][j] > m) { m = v[p][j]; am = p; }
      const float gv = (m > 0.f) ? g[4 * c4 + j][qd] : 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) d[p][j] = (v[p][j] > 0.f ? u[p][j] : 0.f) + (am == p ? gv : 0.f);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) __builtin_nontemporal_store(d[p], dfeat + base + 8 * p);
  }
}

__global__ __launch_bounds__(256) void pool4_bwd_quad(const float* __restrict__ dpooled,
                                                      const f32x4* __restrict__ feat, f32x4* __restrict__ dfeat,
                                                      int B, long HW, int C) {
  const int groups = C / 4;
  const long quads = HW / 4;
  const long total = (long)B * quads * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long q = (i / groups) % quads;
    const long b = i / (groups * quads);
    const long base = (b * HW + 4 * q) * groups + g;
    const f32x4 v0 = feat[base], v1 = feat[base + groups], v2 = feat[base + 2 * groups], v3 = feat[base + 3 * groups];
    const float* gp = dpooled + b * (quads * C) + q;
    f32x4 d0, d1, d2, d3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gp[(long)(4 * g + k) * quads];
      // first maximum wins (torch max_pool1d keeps the earliest index on ties); ReLU backward fused: feat > 0
      float m = v0[k];
      int am = 0;
      if (v1[k] > m) { m = v1[k]; am = 1; }
      if (v2[k] > m) { m = v2[k]; am = 2; }
      if (v3[k] > m) { m = v3[k]; am = 3; }
      const float gv = (m > 0.f) ? gk : 0.f;
      d0[k] = am == 0 ? gv : 0.f;
      d1[k] = am == 1 ? gv : 0.f;
      d2[k] = am == 2 ? gv : 0.f;
      d3[k] = am == 3 ? gv : 0.f;
    }
    __builtin_nontemporal_store(d0, dfeat + base);
    __builtin_nontemporal_store(d1, dfeat + base + groups);
    __builtin_nontemporal_store(d2, dfeat + base + 2 * groups);
    __builtin_nontemporal_store(d3, dfeat + base + 3 * groups);
  }
}

// ---- pool, general path (any C, H, W): windows may straddle channel planes; tail elements are dropped.
__global__ __launch_bounds__(256) void pool4_fwd_any(const float* __restrict__ feat, float* __restrict__ pooled, int B,
                                                     long HW, int C) {
  const long per = ((long)C * HW) / 4;
  const long total = (long)B * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / per, g = i % per;
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long f = 4 * g + k;
      const float v = feat[(b * HW + f % HW) * C + f / HW];
      m = (k == 0 || v > m) ? v : m;
    }
    pooled[i] = m;
  }
}

// one thread per NHWC element of dfeat (so elements in the dropped tail get an explicit zero)
__global__ __launch_bounds__(256) void pool4_bwd_any(const float* __restrict__ dpooled, const float* __restrict__ feat,
                                                     float* __restrict__ dfeat, int B, long HW, int C) {
  const long per = ((long)C * HW) / 4;
  const long total = (long)B * HW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long p = (i / C) % HW;
    const long b = i / ((long)C * HW);
    const long f = (long)c * HW + p;
    const long g = f / 4;
    float out = 0.f;
    if (g < per) {
      float m = 0.f;
      int am = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long fk = 4 * g + k;
        const float v = feat[(b * HW + fk % HW) * C + fk / HW];
        if (k == 0 || v > m) { m = v; am = k; }
      }
      if (am == (int)(f - 4 * g) && m > 0.f) out = dpooled[b * per + g];
    }
    dfeat[i] = out;
  }
}

int grid_for(long n) { return (int)min((n + 255) / 256, (long)DD_NUM_CU * 8); }

// dst[b][u][v] = (src[b][stride*u + offset][stride*v + offset], 0, 0, 0), zero outside the image: the pixels a strided, equally
// dilated, single-channel convolution ever reads (rm_conv_1: k7 stride 3 dilation 3 pad 1 touches only rows / columns
// 3u - 1), as an NHWC4 image on which that convolution is dense
__global__ __launch_bounds__(256) void subsample_nhwc4_kernel(const float* __restrict__ src, f32x4* __restrict__ dst, int h, int w,
                                                              int oh, int ow, int stride, int offset, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int v = (int)(i % ow);
  const long r = i / ow;
  const int u = (int)(r % oh), b = (int)(r / oh);
  const int y = stride * u + offset, x = stride * v + offset;
  float val = 0.f;
  if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) val = src[((long)b * h + y) * w + x];
  dst[i] = f32x4{val, 0.f, 0.f, 0.f};
}

// the same from the collate's per-sample masks where they lie: bool / uint8 [h,w] tensors (data_helper.py:137-139 hands the
// road image over as a bool tensor), read through a pointer table -- no torch.stack, no .float() copy (spatial_w_rm.py:105)
struct MaskPtrs {
  const unsigned char* p[64];
};

__global__ __launch_bounds__(256) void subsample_nhwc4_u8_ptrs_kernel(const MaskPtrs masks, f32x4* __restrict__ dst, int h, int w,
                                                                      int oh, int ow, int stride, int offset, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int v = (int)(i % ow);
  const long r = i / ow;
  const int u = (int)(r % oh), b = (int)(r / oh);
  const int y = stride * u + offset, x = stride * v + offset;
  float val = 0.f;
  if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) val = masks.p[b][(long)y * w + x] ? 1.f : 0.f;
  dst[i] = f32x4{val, 0.f, 0.f, 0.f};
}

}  // namespace

extern "C" {

int dd_stitch6(const float* views, float* wide_nhwc4, float* wide_nchw, float* target, int32_t batch, int32_t height,
               int32_t width, int32_t mask_slot, void* stream) {
  DD_REQUIRE(views && (wide_nhwc4 || wide_nchw), DD_ERR_BAD_ARG, "stitch6: NULL pointer");
  DD_REQUIRE(batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6: non-positive size");
  DD_REQUIRE(mask_slot >= -1 && mask_slot < 6, DD_ERR_BAD_ARG, "stitch6: mask_slot %d", mask_slot);
  const long npx = (long)batch * height * 6 * width;
  hipLaunchKernelGGL(stitch6_kernel, dim3(grid_for(npx)), dim3(256), 0, (hipStream_t)stream, views, (f32x4*)wide_nhwc4,
                     wide_nchw, target, batch, height, width, mask_slot);
  DD_LAUNCH_CHECK("stitch6");
  return 0;
}

int dd_stitch6_ptrs(const float* const* sample_ptrs, float* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                    void* stream) {
  DD_REQUIRE(sample_ptrs && wide_nhwc4 && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6_ptrs: bad argument");
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    SamplePtrs tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "stitch6_ptrs: null sample pointer");
    const long npx = (long)nb * height * 6 * width;
    hipLaunchKernelGGL(stitch6_ptrs_kernel, dim3(grid_for(npx)), dim3(256), 0, (hipStream_t)stream, tab,
                       (f32x4*)wide_nhwc4 + (long)b0 * height * 6 * width, nb, height, width);
    DD_LAUNCH_CHECK("stitch6_ptrs");
  }
  return 0;
}

int dd_stitch6_u8(const unsigned char* frames, float* wide_nhwc4, int32_t batch, int32_t height, int32_t width, void* stream) {
  DD_REQUIRE(frames && wide_nhwc4 && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6_u8: bad argument");
  const long npx = (long)batch * height * 6 * width;
  hipLaunchKernelGGL(stitch6_u8_kernel, dim3(grid_for(npx)), dim3(256), 0, (hipStream_t)stream, frames, (f32x4*)wide_nhwc4,
                     batch, height, width);
  DD_LAUNCH_CHECK("stitch6_u8");
  return 0;
}

int dd_stitch6_u8_ptrs(const unsigned char* const* sample_ptrs, float* wide_nhwc4, float* target, int32_t batch, int32_t height,
                       int32_t width, int32_t mask_slot, void* stream) {
  DD_REQUIRE(sample_ptrs && wide_nhwc4 && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6_u8_ptrs: bad argument");
  DD_REQUIRE(mask_slot >= -1 && mask_slot < 6, DD_ERR_BAD_ARG, "stitch6_u8_ptrs: mask_slot %d", mask_slot);
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    SamplePtrsU8 tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "stitch6_u8_ptrs: null sample pointer");
    const long npx = (long)nb * height * 6 * width;
    hipLaunchKernelGGL(stitch6_u8_ptrs_kernel, dim3(grid_for(npx)), dim3(256), 0, (hipStream_t)stream, tab,
                       (f32x4*)wide_nhwc4 + (long)b0 * height * 6 * width, target ? target + (long)b0 * 3 * height * width : nullptr, nb,
                       height, width, mask_slot);
    DD_LAUNCH_CHECK("stitch6_u8_ptrs");
  }
  return 0;
}

int dd_subsample_nhwc4_u8_ptrs(const unsigned char* const* mask_ptrs, float* dst, int32_t batch, int32_t h, int32_t w, int32_t oh,
                               int32_t ow, int32_t stride, int32_t offset, void* stream) {
  DD_REQUIRE(mask_ptrs && dst && batch > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && stride > 0, DD_ERR_BAD_ARG,
             "subsample_nhwc4_u8_ptrs: bad argument");
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    MaskPtrs tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? mask_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "subsample_nhwc4_u8_ptrs: null mask pointer");
    const long total = (long)nb * oh * ow;
    hipLaunchKernelGGL(subsample_nhwc4_u8_ptrs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tab,
                       (f32x4*)dst + (long)b0 * oh * ow, h, w, oh, ow, stride, offset, total);
    DD_LAUNCH_CHECK("subsample_nhwc4_u8_ptrs");
  }
  return 0;
}

int dd_subsample_nhwc4(const float* src, float* dst, int32_t batch, int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t stride,
                       int32_t offset, void* stream) {
  DD_REQUIRE(src && dst && batch > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && stride > 0, DD_ERR_BAD_ARG, "subsample_nhwc4: bad argument");
  const long total = (long)batch * oh * ow;
  hipLaunchKernelGGL(subsample_nhwc4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (f32x4*)dst, h, w,
                     oh, ow, stride, offset, total);
  DD_LAUNCH_CHECK("subsample_nhwc4");
  return 0;
}

int dd_nchw_to_nhwc(const float* src, float* dst, int32_t batch, int32_t c, int32_t h, int32_t w, int32_t c_store,
                    void* stream) {
  DD_REQUIRE(src && dst && batch > 0 && c > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "nchw_to_nhwc: bad argument");
  DD_REQUIRE(c_store >= c && c_store % 4 == 0, DD_ERR_UNSUPPORTED, "nchw_to_nhwc: c_store %d for c %d", c_store, c);
  if (c_store >= 8 && c_store <= 64) {
    const long plane = (long)h * w;
    hipLaunchKernelGGL(layout_tile_kernel<true>, dim3((unsigned)((plane + 63) / 64), batch), dim3(256), 0, (hipStream_t)stream, src, dst, c,
                       plane, c_store);
    DD_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
  }
  const long total = (long)batch * h * w * (c_store / 4);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, (f32x4*)dst,
                     batch, c, h, w, c_store);
  DD_LAUNCH_CHECK("nchw_to_nhwc");
  return 0;
}

int dd_nhwc_to_nchw(const float* src, float* dst, int32_t batch, int32_t c, int32_t h, int32_t w, int32_t c_store,
                    void* stream) {
  DD_REQUIRE(src && dst && batch > 0 && c > 0 && h > 0 && w > 0 && c_store >= c, DD_ERR_BAD_ARG, "nhwc_to_nchw: bad argument");
  if (c_store >= 8 && c_store <= 64) {
    const long plane = (long)h * w;
    hipLaunchKernelGGL(layout_tile_kernel<false>, dim3((unsigned)((plane + 63) / 64), batch), dim3(256), 0, (hipStream_t)stream, src, dst, c,
                       plane, c_store);
    DD_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
  }
  const long total = (long)batch * c * h * w;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, batch, c,
                     h, w, c_store);
  DD_LAUNCH_CHECK("nhwc_to_nchw");
  return 0;
}

int dd_pool4_fwd(const float* feat, float* pooled, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  DD_REQUIRE(feat && pooled && batch > 0 && h > 0 && w > 0 && c > 0, DD_ERR_BAD_ARG, "pool4_fwd: bad argument");
  const long HW = (long)h * w;
  DD_REQUIRE(((long)c * HW) / 4 > 0, DD_ERR_UNSUPPORTED, "pool4_fwd: fewer than 4 elements");
  if (HW % 4 == 0 && c % 4 == 0) {
    const long total = (long)batch * (HW / 4) * (c / 4);
    hipLaunchKernelGGL(pool4_fwd_quad<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)feat,
                       pooled, (unsigned short*)nullptr, batch, HW, c);
  } else {
    const long total = (long)batch * (((long)c * HW) / 4);
    hipLaunchKernelGGL(pool4_fwd_any, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, feat, pooled, batch, HW,
                       c);
  }
  DD_LAUNCH_CHECK("pool4_fwd");
  return 0;
}

int dd_pool4_relu_bwd(const float* dpooled, const float* feat, float* dfeat, int32_t batch, int32_t h, int32_t w,
                      int32_t c, void* stream) {
  DD_REQUIRE(dpooled && feat && dfeat && batch > 0 && h > 0 && w > 0 && c > 0, DD_ERR_BAD_ARG, "pool4_bwd: bad argument");
  const long HW = (long)h * w;
  if (HW % 4 == 0 && c % 4 == 0) {
    const long total = (long)batch * (HW / 4) * (c / 4);
    hipLaunchKernelGGL(pool4_bwd_quad, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dpooled,
                       (const f32x4*)feat, (f32x4*)dfeat, batch, HW, c);
  } else {
    const long total = (long)batch * HW * c;
    hipLaunchKernelGGL(pool4_bwd_any, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dpooled, feat, dfeat,
                       batch, HW, c);
  }
  DD_LAUNCH_CHECK("pool4_bwd");
  return 0;
}

int dd_pool4_relu_bwd_add(const float* dpooled, const float* feat, const float* gfeat, float* dfeat, int32_t batch, int32_t h, int32_t w,
                          int32_t c, void* stream) {
  DD_REQUIRE(dpooled && feat && gfeat && dfeat && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "pool4_relu_bwd_add: bad argument");
  DD_REQUIRE(c == 32 && ((long)h * w) % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_relu_bwd_add: needs C == 32 and H*W %% 4 == 0 (got %dx%d, C %d)", h, w, c);
  DD_REQUIRE((((uintptr_t)feat | (uintptr_t)gfeat | (uintptr_t)dfeat) & 15) == 0, DD_ERR_BAD_ARG, "pool4_relu_bwd_add: 16-byte alignment");
  const long quads = (long)h * w / 4, qblocks = (quads + 63) / 64;
  DD_REQUIRE(batch * qblocks < (1L << 31), DD_ERR_UNSUPPORTED, "pool4_relu_bwd_add: too many tiles");
  hipLaunchKernelGGL(pool4_bwd_add_tile32, dim3((unsigned)(batch * qblocks)), dim3(256), 0, (hipStream_t)stream, dpooled, (const f32x4*)feat,
                     (const f32x4*)gfeat, (f32x4*)dfeat, quads, (int)qblocks);
  DD_LAUNCH_CHECK("pool4_relu_bwd_add");
  return 0;
}

int64_t dd_pool4_idx_elems(int32_t batch, int32_t h, int32_t w, int32_t c) {
  const long HW = (long)h * w;
  if (batch <= 0 || h <= 0 || w <= 0 || c <= 0 || HW % 4 != 0 || c % 4 != 0) {
    dd_fail(DD_ERR_UNSUPPORTED, "pool4_idx: needs H*W %% 4 == 0 and C %% 4 == 0 (windows inside one channel plane), got %d x %d x %d", h, w, c);
    return -1;
  }
  return (int64_t)batch * (HW / 4) * (c / 4);
}

int dd_pool4_fwd_idx(const float* feat, float* pooled, uint16_t* idx, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  DD_REQUIRE(feat && pooled && idx, DD_ERR_BAD_ARG, "pool4_fwd_idx: NULL pointer");
  const int64_t total = dd_pool4_idx_elems(batch, h, w, c);
  if (total < 0) return DD_ERR_UNSUPPORTED;
  const long quads = (long)h * w / 4, qblocks = (quads + 63) / 64;
  if (c == 32 && batch * qblocks < (1L << 31) && ((uintptr_t)feat & 15) == 0 && ((uintptr_t)idx & 3) == 0)
    hipLaunchKernelGGL(pool4_fwd_tile32, dim3((unsigned)(batch * qblocks)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)feat, pooled,
                       (unsigned*)idx, quads, (int)qblocks);
  else
    hipLaunchKernelGGL(pool4_fwd_quad<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)feat, pooled,
                       (unsigned short*)idx, batch, (long)h * w, c);
  DD_LAUNCH_CHECK("pool4_fwd_idx");
  return 0;
}

int dd_pool4_idx_relu_bwd(const float* dpooled, const uint16_t* idx, float* dfeat, int32_t batch, int32_t h, int32_t w, int32_t c,
                          void* stream) {
  DD_REQUIRE(dpooled && idx && dfeat, DD_ERR_BAD_ARG, "pool4_idx_bwd: NULL pointer");
  const int64_t total = dd_pool4_idx_elems(batch, h, w, c);
  if (total < 0) return DD_ERR_UNSUPPORTED;
  const long quads = (long)h * w / 4, qblocks = (quads + 63) / 64;
  if (c == 32 && batch * qblocks < (1L << 31) && ((uintptr_t)dfeat & 15) == 0 && ((uintptr_t)idx & 3) == 0)
    hipLaunchKernelGGL(pool4_bwd_tile32, dim3((unsigned)(batch * qblocks)), dim3(256), 0, (hipStream_t)stream, dpooled, (const unsigned*)idx,
                       (f32x4*)dfeat, quads, (int)qblocks);
  else
    hipLaunchKernelGGL(pool4_bwd_idx_quad, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dpooled, (const unsigned short*)idx,
                       (f32x4*)dfeat, batch, (long)h * w, c);
  DD_LAUNCH_CHECK("pool4_idx_bwd");
  return 0;
}

}  // extern "C"
