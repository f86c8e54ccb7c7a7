// Generic NHWC convolution on the fp32 matrix cores: every Conv2d / ConvTranspose2d of the path that is not one
// of the encoder's three k3 p1 layers (those have the LDS-ring kernels in conv3x3.hip).
//
//   decoder           ConvTranspose2d 64->32 k3 p1, 32->32 k3 p1, 32->32 k2 s2, 32->3 k1   components.py:70-73,89-92
//   SpatialMappingCNN Conv2d 3->32 k(1,50)/(52,1) s(3,2), 32->32 k3                         spatial_bb/components.py:18-26
//   *MergingCNN       Conv2d 32->32 k(1,24) s(1,7), 1->32 k7 s3 d3 p1, 32->32 k3 d3,
//                     ConvTranspose2d k2 s2, and the dilated k7/k8 d7/d8/d3 up-convs          spatial_bb/components.py:88-93,129-139
//
// GEMM view: M = 32 output pixels along x (one wave), N = 32 or 64 output channels, K = taps x Cin walked in
// 16-byte "chunks" (4 channels of one tap).  There is no input tile in LDS: dilated taps are up to 49 pixels
// apart, so a halo tile would be mostly holes; instead the A operand is gathered straight from L1/L2 with one
// buffer_load_dwordx4 per lane per 4 MFMAs (the fp32 MFMA needs 256 cycles for those four, the gather ~32), the
// B operand is the packed weight image streamed from L2 with lane-linear 16-byte loads.  A per-workgroup LDS
// table turns the chunk index into (row offset, column offset, channel) so the inner loop has no integer
// division.  Loads run one 4-group set ahead of the MFMAs (register ping-pong).
// Zero padding, ragged edges, divisibility holes of transposed gathers: out-of-range buffer offsets.
#include <stdlib.h>

#include "dd_common.h"

namespace {

constexpr int GU = 4;   // chunk-pair groups per pipeline set

template <bool DIV>
__device__ __forceinline__ int gather_offset(const dd_gconv_desc& d, int yo, int xo, int row_off, int col_off, int chan) {
  int yn = yo * d.stride_h + row_off, xn = xo * d.stride_w + col_off;
  bool ok = true;
  if (DIV) {
    ok = (yn >= 0) && (xn >= 0) && (yn % d.div_h == 0) && (xn % d.div_w == 0);
    yn = yn / d.div_h;
    xn = xn / d.div_w;
  }
  ok = ok && ((unsigned)yn < (unsigned)d.in_h) && ((unsigned)xn < (unsigned)d.in_w);
  return ok ? ((yn * d.in_w + xn) * d.in_cstore + chan) * 4 : -16;
}

__device__ __forceinline__ int out_offset(const dd_gconv_desc& d, int yo, int xo, int chan) {
  const bool ok = (xo < d.out_w) && (chan < d.cout);
  return ok ? (((yo * d.ostride_h + d.ooff_h) * d.omem_w + xo * d.ostride_w + d.ooff_w) * d.out_cstore + d.out_coff + chan) * 4
            : -16;
}

// Valid tap rows of output row yo / valid tap columns of the 32-pixel strip at x0 (taps that touch the image).
__device__ __forceinline__ void tap_rows(const dd_gconv_desc& d, int yo, int& k0, int& k1) {
  const int ry = yo * d.stride_h - d.pad_h;                         // input row of tap ky = ry + ky*dil_h
  k0 = ry >= 0 ? 0 : (-ry + d.dil_h - 1) / d.dil_h;
  k1 = min(d.kh - 1, (d.in_h - 1 - ry) >= 0 ? (d.in_h - 1 - ry) / d.dil_h : -1);
}
__device__ __forceinline__ void tap_cols(const dd_gconv_desc& d, int x0, int& k0, int& k1) {
  const int xlo = x0 * d.stride_w - d.pad_w;                        // leftmost / rightmost pixel of the strip
  const int xhi = min(x0 + 31, d.out_w - 1) * d.stride_w - d.pad_w;
  k0 = xhi >= 0 ? 0 : (-xhi + d.dil_w - 1) / d.dil_w;
  k1 = min(d.kw - 1, (d.in_w - 1 - xlo) >= 0 ? (d.in_w - 1 - xlo) / d.dil_w : -1);
}

// First row-tile index of piece `i` of `n` equal-cost pieces; cost(tile) = 1 + valid tap rows x valid tap columns
// (the 1 stands for the epilogue, so that all-padding tiles are still spread).  Wave-uniform scalar walk, once per wave.
__device__ __forceinline__ long cost_cut(const dd_gconv_desc& d, int nstrips, int i, int n) {
  long rows_cost = 0;      // sum over rows of nky
  for (int yo = 0; yo < d.out_h; ++yo) {
    int a, b;
    tap_rows(d, yo, a, b);
    rows_cost += max(b - a + 1, 0);
  }
  long img_cost = 0;
  for (int s = 0; s < nstrips; ++s) {
    int a, b;
    tap_cols(d, s * 32, a, b);
    img_cost += (long)max(b - a + 1, 0) * rows_cost + d.out_h;
  }
  const long total = img_cost * d.batch;
  if (i >= n) return (long)d.batch * nstrips * d.out_h;
  long target = (total * i) / n;
  const long bimg = target / img_cost;
  target -= bimg * img_cost;
  long idx = bimg * nstrips * d.out_h;
  for (int s = 0; s < nstrips; ++s) {
    int a, b;
    tap_cols(d, s * 32, a, b);
    const long nkx = max(b - a + 1, 0);
    const long col_cost = nkx * rows_cost + d.out_h;
    if (target >= col_cost) {
      target -= col_cost;
      idx += d.out_h;
      continue;
    }
    for (int yo = 0; yo < d.out_h; ++yo) {
      int c, e;
      tap_rows(d, yo, c, e);
      const long tc = nkx * max(e - c + 1, 0) + 1;
      if (target < tc) return idx + yo;
      target -= tc;
    }
    return idx + d.out_h;
  }
  return idx;
}

template <int NT, bool DIV>
__global__ __launch_bounds__(512) void gconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, const float* __restrict__ msk,
                                                        float* __restrict__ y, const dd_gconv_desc d, int ngroups,
                                                        int epi, int spt) {
  extern __shared__ __attribute__((aligned(16))) int2 tab[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C4 = d.cin >> 2, nchunks = d.kh * d.kw * C4;
  for (int q = tid; q < 2 * ngroups; q += blockDim.x) {
    int2 e;
    if (q < nchunks) {
      const int tap = q / C4, c4 = q - tap * C4;
      const int ky = tap / d.kw, kx = tap - ky * d.kw;
      e.x = ky * d.dil_h - d.pad_h;
      e.y = ((kx * d.dil_w - d.pad_w) << 10) | (d.in_coff + c4 * 4);
    } else {   // padding chunk: its packed weights are zero, its coordinate is out of range
      e.x = 1 << 24;
      e.y = 0;
    }
    tab[q] = e;
  }
  __syncthreads();
  const int h = lane >> 5, n = lane & 31;
  const int nstrips = (d.out_w + 31) / 32;
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;

  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    bv[nt] = (bias && (epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU || epi == DD_EPI_BIAS_SIGMOID) && nt * 32 + n < d.cout)
                 ? bias[nt * 32 + n] : 0.f;

  long idx, end;
  {
    const int gw = blockIdx.x * (blockDim.x >> 6) + wave, nw = gridDim.x * (blockDim.x >> 6);
    if (!DIV && spt > 0) {
      // Row tiles at the image border skip most taps (below), so equal COUNTS would leave the interior waves with
      // all the work: cut the (image, strip, row) sequence into pieces of equal COST = valid tap rows x valid tap columns.
      idx = cost_cut(d, nstrips, gw, nw);
      end = cost_cut(d, nstrips, gw + 1, nw);
    } else {
      dd_range((long)d.batch * nstrips * d.out_h, gw, nw, idx, end);
    }
  }
  for (; idx < end; ++idx) {
    const long col = idx / d.out_h;
    const int yo = (int)(idx - col * d.out_h);
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const int xo = x0 + n;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    f32x4 A0[GU], A1[GU], B0[GU][NT], B1[GU][NT];
#define DD_LOAD_SET(G0, A, Bw)                                                                       \
  _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                                   \
    const int2 e = tab[2 * ((G0) + u) + h];                                                          \
    A[u] = dd_bload4(xs, gather_offset<DIV>(d, yo, xo, e.x, e.y >> 10, e.y & 1023));                 \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                \
      Bw[u][nt] = *(const f32x4*)(wp + ((long)(((G0) + u) * NT + nt) * 64 + lane) * 4);              \
  }
#define DD_COMPUTE_SET(A, Bw)                                                                        \
  _Pragma("unroll") for (int u = 0; u < GU; ++u) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { \
    acc[nt] = DD_MFMA(A[u].x, Bw[u][nt].x, acc[nt]);                                                 \
    acc[nt] = DD_MFMA(A[u].y, Bw[u][nt].y, acc[nt]);                                                 \
    acc[nt] = DD_MFMA(A[u].z, Bw[u][nt].z, acc[nt]);                                                 \
    acc[nt] = DD_MFMA(A[u].w, Bw[u][nt].w, acc[nt]);                                                 \
  }
    // Taps whose input row lies outside the image, or whose input columns lie outside it for ALL 32 pixels of the
    // strip, contribute nothing: they are skipped for the whole wave (a flipped-tap transposed conv with
    // pad = d(k-1) would otherwise spend (out/in)^2 - 1 = 26..36 % of its MFMAs on zeros).  Possible when a pipeline
    // set (GU chunk pairs) never straddles two taps: spt = sets per tap.
    int ky0 = 0, ky1 = d.kh - 1, kx0 = 0, kx1 = d.kw - 1;
    if (!DIV && spt > 0) {
      tap_rows(d, yo, ky0, ky1);
      tap_cols(d, x0, kx0, kx1);
    }
    const int nky = max(ky1 - ky0 + 1, 0), nkx = max(kx1 - kx0 + 1, 0);
    const int per_tap = spt > 0 ? spt : (ngroups / GU);               // spt == 0: one "tap" = all sets, no skipping
    const int nsets = spt > 0 ? nky * nkx * spt : per_tap;
    // iterator over the valid sets: (ky, kx, cs) -> first group of the set
    int it_ky = ky0, it_kx = kx0, it_cs = 0;
#define DD_NEXT_SET(G)                                                                               \
  {                                                                                                  \
    G = (spt > 0 ? ((it_ky * d.kw + it_kx) * spt + it_cs) : it_cs) * GU;                             \
    if (++it_cs == per_tap) { it_cs = 0; if (++it_kx > kx1) { it_kx = kx0; ++it_ky; } }              \
  }
    if (nsets > 0) {
      int g;
      DD_NEXT_SET(g)
      DD_LOAD_SET(g, A0, B0)
      for (int i = 0; i < nsets; i += 2) {
        const bool more1 = i + 1 < nsets, more2 = i + 2 < nsets;
        if (more1) { DD_NEXT_SET(g) DD_LOAD_SET(g, A1, B1) }
        __builtin_amdgcn_sched_barrier(0);          // set k+1 is requested before set k is multiplied
        DD_COMPUTE_SET(A0, B0)
        if (more2) { DD_NEXT_SET(g) DD_LOAD_SET(g, A0, B0) }
        __builtin_amdgcn_sched_barrier(0);
        if (more1) { DD_COMPUTE_SET(A1, B1) }
      }
    }
#undef DD_NEXT_SET
#undef DD_LOAD_SET
#undef DD_COMPUTE_SET

    const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
    const __amdgpu_buffer_rsrc_t ms = dd_rsrc(msk ? msk + (long)b * d.omem_h * d.omem_w * d.out_cstore : y, msk ? out_bytes : 0);
    // All mask values of a 32 x 32 tile are requested before the first is used (as in dconv.hip): one conditional load per element made
    // the compiler wait for each load in turn -- 16 serial memory round trips per column tile, which for the short reductions of the
    // phase launches (ss_conv's data gradient: 64 MFMAs per tile) was most of the launch.
    const bool masked = epi == DD_EPI_RELU_MASK;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ch = d.out_coff + nt * 32 + n;      // channels [mask_pass_lo, mask_pass_hi) of the buffer are not ReLU outputs
      const bool pass = ch >= d.mask_pass_lo && ch < d.mask_pass_hi;
      int off[16];
      float mv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        off[r] = out_offset(d, yo, x0 + dd_acc_row(r, lane), nt * 32 + n);
        mv[r] = 1.f;
      }
      if (masked) {      // lanes of an exempt channel ask an out-of-range offset (no memory request) and keep 1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m = dd_bload1(ms, pass ? -16 : off[r]);
          mv[r] = pass ? 1.f : m;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nt][r] + bv[nt];
        if (epi == DD_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        if (epi == DD_EPI_BIAS_SIGMOID) v = 1.f / (1.f + expf(-v));
        v = mv[r] > 0.f ? v : 0.f;
        dd_bstore1(ys, off[r], v);
      }
    }
  }
}

// packed[((g*NT + nt)*64 + lane)*4 + i] = W(n = nt*32 + lane&31, c = 4*(chunk % C4) + i, tap = chunk / C4),
// chunk = 2g + (lane>>5).
__global__ void gconv_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int ngroups, int nt_count, int C4,
                                  int T, long w_off, long sn, long sc, int flip, int n_real, int c_real) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)ngroups * nt_count * 256;
  if (idx >= total) return;
  const int i = idx & 3, lane = (idx >> 2) & 63;
  const long gn = idx >> 8;
  const int nt = (int)(gn % nt_count), g = (int)(gn / nt_count);
  const int q = 2 * g + (lane >> 5);
  const int tap = q / C4, c = 4 * (q - tap * C4) + i, n = nt * 32 + (lane & 31);
  float v = 0.f;
  if (tap < T && n < n_real && c < c_real) v = w[w_off + n * sn + c * sc + (flip ? T - 1 - tap : tap)];
  p[idx] = v;
}

// ---------------------------------------------------------------------------------------------- weight gradient
// One wave = one 32-channel slice of dy (A operand) x NB column tiles of the (tap, cin) space (B operand), over a
// contiguous range of output row tiles; waves of a workgroup share the range so dy comes out of L1.
template <int NO, int NB>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ part, float* __restrict__ bpart,
                                                          const dd_gconv_desc d, int njg, int nog, int nranges) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
  const int njobs = nog * njg;
  const int job = gw % njobs, range = gw / njobs;
  if (range >= nranges) return;
  const int og = job / njg, jg = job - og * njg;
  const int h = lane >> 5, n = lane & 31;
  const int J = d.kh * d.kw * d.cin;

  int rowoff[NB], coloff[NB], chan[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int j = (jg * NB + k) * 32 + n;
    if (j < J) {
      const int tap = j / d.cin, c = j - tap * d.cin;
      const int ky = tap / d.kw, kx = tap - ky * d.kw;
      rowoff[k] = ky * d.dil_h - d.pad_h;
      coloff[k] = kx * d.dil_w - d.pad_w;
      chan[k] = d.in_coff + c;
    } else {
      rowoff[k] = 1 << 24;
      coloff[k] = 0;
      chan[k] = 0;
    }
  }
  f32x16 acc[NO][NB];
#pragma unroll
  for (int a = 0; a < NO; ++a)
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][k][r] = 0.f;
  float bsum[NO];
#pragma unroll
  for (int a = 0; a < NO; ++a) bsum[a] = 0.f;

  const int nstrips = (d.out_w + 31) / 32;
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  long idx, end;
  dd_range((long)d.batch * nstrips * d.out_h, range, nranges, idx, end);
  for (; idx < end; ++idx) {
    const long col = idx / d.out_h;
    const int yo = (int)(idx - col * d.out_h);
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const __amdgpu_buffer_rsrc_t gs = dd_rsrc(dy + (long)b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
#pragma unroll
    for (int half = 0; half < 2; ++half) {   // 8 pixel pairs at a time: 8 x (NO + NB) loads in flight, then 8 x NO x NB MFMAs
      float av[8][NO], bw[8][NB];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int xo = x0 + 2 * (half * 8 + s) + h;
#pragma unroll
        for (int a = 0; a < NO; ++a) av[s][a] = dd_bload1(gs, out_offset(d, yo, xo, (og * NO + a) * 32 + n));
#pragma unroll
        for (int k = 0; k < NB; ++k)
          bw[s][k] = dd_bload1(xs, gather_offset<false>(d, yo, xo, rowoff[k], coloff[k], chan[k]));   // av = 0 beyond out_w
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int a = 0; a < NO; ++a) {
          bsum[a] += av[s][a];
#pragma unroll
          for (int k = 0; k < NB; ++k) acc[a][k] = DD_MFMA(av[s][a], bw[s][k], acc[a][k]);
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NO; ++a) {
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part[(((((long)range * njobs + job) * NO + a) * NB + k) * 16 + r) * 64 + lane] = acc[a][k][r];
    if (jg == 0) bpart[((long)range * nog * NO + og * NO + a) * 64 + lane] = bsum[a];
  }
}

// dw[w_off + o*sn + c*sc + tap'] (+)= sum over ranges.  One block = 256 consecutive accumulator elements (four register
// rows of one tile) x 16 groups of ranges; a lane reads 16 bytes, so every wave-load is 1 KB of contiguous partials
// (the partials of one range are `per` floats apart: 256-byte pieces ran at a quarter of the bandwidth).
__device__ __forceinline__ void gconv_wgrad_reduce_block(f32x4 (*red)[64], int block, const float* __restrict__ part, float* __restrict__ dw,
                                                         int nranges, int njobs, int no, int nb, int njg, int T, int cin,
                                                         long w_off, long sn, long sc, int flip, int n_real, int c_real,
                                                         int accumulate) {
  const long per = (long)njobs * no * nb * 1024;
  const int l64 = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long e0 = (long)block * 256 + 4 * l64;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 s0 = z, s1 = z, s2 = z, s3 = z;
  int r = g;
  for (; r + 48 < nranges; r += 64) {
    s0 += *(const f32x4*)(part + (long)r * per + e0);
    s1 += *(const f32x4*)(part + (long)(r + 16) * per + e0);
    s2 += *(const f32x4*)(part + (long)(r + 32) * per + e0);
    s3 += *(const f32x4*)(part + (long)(r + 48) * per + e0);
  }
  for (; r < nranges; r += 16) s0 += *(const f32x4*)(part + (long)r * per + e0);
  red[g][l64] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0) return;
  f32x4 sv = z;
#pragma unroll
  for (int i = 0; i < 16; ++i) sv += red[i][l64];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long e = e0 + q;
    const int lane = (int)(e & 63);
    const int reg = (int)((e >> 6) & 15);
    const long jk = e >> 10;
    const int k = (int)(jk % nb), a = (int)((jk / nb) % no), job = (int)(jk / ((long)nb * no));
    const int og = job / njg, jg = job - og * njg;
    const int o = (og * no + a) * 32 + dd_acc_row(reg, lane);
    const int j = (jg * nb + k) * 32 + (lane & 31);
    if (j >= T * cin) continue;
    const int tap = j / cin, c = j - tap * cin;
    if (o >= n_real || c >= c_real) continue;
    const long wi = w_off + o * sn + c * sc + (flip ? T - 1 - tap : tap);
    dw[wi] = accumulate ? dw[wi] + sv[q] : sv[q];
  }
}

__device__ __forceinline__ void gconv_bias_reduce_block(float (*red)[64], int ot, const float* __restrict__ bpart, float* __restrict__ db,
                                                        int nranges, int nto, int n_real, int accumulate) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  float s = 0.f;
  for (int r = g; r < nranges; r += 16) s += bpart[((long)r * nto + ot) * 64 + lane];
  red[g][lane] = s;
  __syncthreads();
  if (g != 0) return;
  s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i][lane];
  s += __shfl_xor(s, 32);
  const int o = ot * 32 + lane;
  if (lane < 32 && o < n_real) db[o] = accumulate ? db[o] + s : s;
}

// Both second stages in ONE launch: blocks [0, wblocks) sum the weight partials, the nto blocks behind them the bias partials (two
// launches of 16-30 us per weight gradient before: 13 pairs per step of the box-head model).
__global__ __launch_bounds__(1024) void gconv_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int wblocks,
                                                           const float* __restrict__ bpart, float* __restrict__ db, int nto, int baccumulate,
                                                           int nranges, int njobs, int no, int nb, int njg, int T, int cin,
                                                           long w_off, long sn, long sc, int flip, int n_real, int c_real,
                                                           int accumulate) {
  __shared__ f32x4 red[16][64];
  if ((int)blockIdx.x < wblocks)
    gconv_wgrad_reduce_block(red, blockIdx.x, part, dw, nranges, njobs, no, nb, njg, T, cin, w_off, sn, sc, flip, n_real, c_real, accumulate);
  else
    gconv_bias_reduce_block((float (*)[64])red, blockIdx.x - wblocks, bpart, db, nranges, nto, n_real, baccumulate);
}

// ---------------------------------------------------------------------------------------------- small helpers
__global__ __launch_bounds__(256) void view_to_nhwc4_kernel(const float* __restrict__ views, f32x4* __restrict__ out, int B,
                                                            int H, int W, int view, int tf) {
  const int Ho = (tf == 1 || tf == 2) ? W : H, Wo = (tf == 1 || tf == 2) ? H : W;
  const long total = (long)B * Ho * Wo, plane = (long)H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int j = (int)(p % Wo), i = (int)((p / Wo) % Ho);
    const int b = (int)(p / ((long)Wo * Ho));
    int ys, xs;
    if (tf == 0) { ys = i; xs = j; }
    else if (tf == 1) { ys = j; xs = W - 1 - i; }         // rot90(k=1, dims [2,3]): out[i][j] = in[j][W-1-i]
    else if (tf == 2) { ys = H - 1 - j; xs = i; }         // rot90(k=1, dims [3,2]): out[i][j] = in[H-1-j][i]
    else { ys = H - 1 - i; xs = W - 1 - j; }              // flip([2,3])
    const float* src = views + (((long)b * 6 + view) * 3) * plane + (long)ys * W + xs;
    out[p] = f32x4{src[0], src[plane], src[2 * plane], 0.f};
  }
}

// the same from a table of per-sample base pointers (each [6,3,H,W]): the collate's tuple (helper.py:22-23) without torch.stack
struct ViewSamplePtrs {
  const float* p[64];
};

__global__ __launch_bounds__(256) void view_to_nhwc4_ptrs_kernel(const ViewSamplePtrs samples, f32x4* __restrict__ out, int B,
                                                                 int H, int W, int view, int tf) {
  const int Ho = (tf == 1 || tf == 2) ? W : H, Wo = (tf == 1 || tf == 2) ? H : W;
  const long total = (long)B * Ho * Wo, plane = (long)H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int j = (int)(p % Wo), i = (int)((p / Wo) % Ho);
    const int b = (int)(p / ((long)Wo * Ho));
    int ys, xs;
    if (tf == 0) { ys = i; xs = j; }
    else if (tf == 1) { ys = j; xs = W - 1 - i; }
    else if (tf == 2) { ys = H - 1 - j; xs = i; }
    else { ys = H - 1 - i; xs = W - 1 - j; }
    const float* src = samples.p[b] + ((long)view * 3) * plane + (long)ys * W + xs;
    out[p] = f32x4{src[0], src[plane], src[2 * plane], 0.f};
  }
}

// the same from uint8 HWC frames (per-sample [6,H,W,3]): ToTensor's /255 (true division) fused with the re-layout
struct ViewSamplePtrsU8 {
  const unsigned char* p[64];
};

__global__ __launch_bounds__(256) void view_to_nhwc4_u8_ptrs_kernel(const ViewSamplePtrsU8 samples, f32x4* __restrict__ out, int B,
                                                                    int H, int W, int view, int tf) {
  const int Ho = (tf == 1 || tf == 2) ? W : H, Wo = (tf == 1 || tf == 2) ? H : W;
  const long total = (long)B * Ho * Wo;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
    const int j = (int)(p % Wo), i = (int)((p / Wo) % Ho);
    const int b = (int)(p / ((long)Wo * Ho));
    int ys, xs;
    if (tf == 0) { ys = i; xs = j; }
    else if (tf == 1) { ys = j; xs = W - 1 - i; }
    else if (tf == 2) { ys = H - 1 - j; xs = i; }
    else { ys = H - 1 - i; xs = W - 1 - j; }
    const unsigned char* src = samples.p[b] + (((long)view * H + ys) * W + xs) * 3;
    out[p] = f32x4{(float)src[0] / 255.0f, (float)src[1] / 255.0f, (float)src[2] / 255.0f, 0.f};
  }
}

__global__ __launch_bounds__(256) void add_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                  f32x4* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

// ---- a channel slice of one NHWC buffer into a channel slice of another (torch.cat / tensor slicing of the merging heads'
// concat buffer, components.py:109,159): 16 bytes per lane, both sides in whole lines when the slices are 32 channels wide
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, long npix, int c4,
                                                            int src_cstore, int src_coff, int dst_cstore, int dst_coff) {
  const long total = npix * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    *(f32x4*)(dst + p * dst_cstore + dst_coff + c) = *(const f32x4*)(src + p * src_cstore + src_coff + c);
  }
}

// the same between rectangular WINDOWS of the two buffers ([B, mem_h, mem_w, cstore] each, window origin (y0, x0), h x w pixels): the
// interior of a padded activation into a concat slice, and back
__global__ __launch_bounds__(256) void copy_channels_window_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int h, int w,
                                                                   int c4, int s_mh, int s_mw, int s_y0, int s_x0, int s_cstore, int s_coff,
                                                                   int d_mh, int d_mw, int d_y0, int d_x0, int d_cstore, int d_coff) {
  const long total = (long)B * h * w * c4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    long p = i / c4;
    const int x = (int)(p % w);
    p /= w;
    const int y = (int)(p % h);
    const long b = p / h;
    const long sp = (b * s_mh + s_y0 + y) * s_mw + s_x0 + x, dp = (b * d_mh + d_y0 + y) * d_mw + d_x0 + x;
    *(f32x4*)(dst + dp * d_cstore + d_coff + c) = *(const f32x4*)(src + sp * s_cstore + s_coff + c);
  }
}

// ---- ConvTranspose2d(32 -> 32, k2 s2) + bias + ReLU in ONE launch (the decoder's dc3, components.py:72,91; ss_deconv of the
// box heads): the four output phases are four column tiles of one GEMM -- M = 32 input pixels, N = 4 x 32, K = 32 -- so
// the input is read once (four generic 1x1 launches read it four times, 0.19 ms each at bs 32 for 80 MB in / 80 MB out).
// Lane (m, h) holds channels 16h..16h+15 of pixel m (64 contiguous bytes); MFMA s pairs channels (s, 16 + s); the 64
// weight values a lane needs stay in registers.  Output pixel (2y + a, 2x + b) of phase p = 2a + b: one 128-byte line.
__global__ __launch_bounds__(256) void deconv2x2_c32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                long npix, int w, int relu, int ocs, int ocoff) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, n = lane & 31;
  float bw[4][16];                                       // wt[ci][co][a][b] (IOHW): phase p, k-step s -> ci = s + 16h, co = n
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int s = 0; s < 16; ++s) bw[p][s] = wt[((s + 16 * h) * 32 + n) * 4 + p];
  const float bv = bias ? bias[n] : 0.f;
  const long ntile = (npix + 31) >> 5;
  const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x, (int)min(npix * 128, 0x7fffffffL));
  for (long t = (long)blockIdx.x * 4 + wave; t < ntile; t += (long)gridDim.x * 4) {
    const long px = t * 32 + n;
    f32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = dd_bload4(xs, px < npix ? (int)(px * 128 + h * 64 + j * 16) : -16);
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = DD_MFMA(a[j][i], bw[p][4 * j + i], acc[p]);
    // rows of the accumulator: pixels t*32 + 4h + (e&3) + 8(e>>2); (global row R = image*h + y, column) of the first, then at
    // most one wrap into the next row (w >= 32)
    const long p0 = t * 32 + 4 * h;
    const long r0 = p0 / w;
    const int c0 = (int)(p0 - r0 * w);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int off = (e & 3) + 8 * (e >> 2);
      int c = c0 + off;
      long r = r0;
      if (c >= w) { c -= w; r += 1; }
      if (p0 + off < npix) {
        float* o = out + ((2 * r) * (2L * w) + 2 * c) * ocs + ocoff + n;      // channels [ocoff, ocoff + 32) of an ocs-channel buffer
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          float v = acc[p][e] + bv;
          if (relu) v = fmaxf(v, 0.f);
          o[((p >> 1) * (2L * w) + (p & 1)) * ocs] = v;
        }
      }
    }
  }
}

// ---- ConvTranspose2d(32 -> 3, k1) + bias, written as NCHW (the decoder's dc4, components.py:73,92, and its output layout):
// three dot products per pixel -- a VALU kernel, one 128-byte line read per pixel, coalesced plane writes.
__global__ __launch_bounds__(256) void conv1x1_c32_c3_nchw_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                  const float* __restrict__ bias, float* __restrict__ out, long plane, long npix) {
  __shared__ float wl[32 * 3 + 3];
  if (threadIdx.x < 96) wl[threadIdx.x] = wt[threadIdx.x];                      // wt[ci][k] (IOHW, k1)
  if (threadIdx.x < 3) wl[96 + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    float z0 = wl[96], z1 = wl[97], z2 = wl[98];
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const f32x4 v = *(const f32x4*)(x + p * 32 + c4 * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = c4 * 4 + i;
        z0 += v[i] * wl[ci * 3];
        z1 += v[i] * wl[ci * 3 + 1];
        z2 += v[i] * wl[ci * 3 + 2];
      }
    }
    const long b = p / plane, q = p - b * plane;
    float* o = out + b * 3 * plane + q;
    o[0] = z0;
    o[plane] = z1;
    o[2 * plane] = z2;
  }
}

// ---- last layer of the box heads: ConvTranspose2d(C -> 1, k2 s2) + sigmoid (spatial_bb/components.py:93,139,117,168).
// One output channel cannot feed a 32-wide MFMA column; it is 4*C MACs per input pixel: a VALU kernel, HBM-bound.
template <int C>
__global__ __launch_bounds__(256) void deconv2x2_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               long npix, int w) {
  float wr[C][4];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) wr[c][ph] = wt[c * 4 + ph];
  const float b0 = bias[0];
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    float z[4] = {b0, b0, b0, b0};
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const f32x4 v = *(const f32x4*)(x + p * C + c4 * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) z[ph] += v[i] * wr[c4 * 4 + i][ph];
    }
    const long row = p / w, col = p - row * w;          // row runs over B*h
    float* o = out + (2 * row) * (2L * w) + 2 * col;
    o[0] = 1.f / (1.f + expf(-z[0]));
    o[1] = 1.f / (1.f + expf(-z[1]));
    o[2L * w] = 1.f / (1.f + expf(-z[2]));
    o[2L * w + 1] = 1.f / (1.f + expf(-z[3]));
  }
}

// dz = dprob * p * (1-p);  dx[c] = (x[c] > 0) * sum_ph dz[ph] * Wt[c][ph];  per-block partials of dWt, dbias.
template <int C>
__global__ __launch_bounds__(256) void deconv2x2_c1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                               const float* __restrict__ probs,
                                                               const float* __restrict__ dprobs, float* __restrict__ dx,
                                                               float* __restrict__ partial, long npix, int w) {
  __shared__ float red[4][C * 4 + 1];
  float wr[C][4];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) wr[c][ph] = wt[c * 4 + ph];
  float gw[C][4], gb = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) gw[c][ph] = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long row = p / w, col = p - row * w;
    const long o = (2 * row) * (2L * w) + 2 * col;
    const long oo[4] = {o, o + 1, o + 2L * w, o + 2L * w + 1};
    float dz[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const float pr = probs[oo[ph]];
      dz[ph] = dprobs[oo[ph]] * pr * (1.f - pr);
      gb += dz[ph];
    }
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const f32x4 v = *(const f32x4*)(x + p * C + c4 * 4);
      f32x4 g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c4 * 4 + i;
        float s = 0.f;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          s += dz[ph] * wr[c][ph];
          gw[c][ph] += v[i] * dz[ph];
        }
        g[i] = v[i] > 0.f ? s : 0.f;
      }
      *(f32x4*)(dx + p * C + c4 * 4) = g;
    }
  }
  // block reduction of the C*4 + 1 sums: wave shuffle, then the 4 waves in a fixed order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      float v = gw[c][ph];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if (lane == 0) red[wave][c * 4 + ph] = v;
    }
  {
    float v = gb;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][C * 4] = v;
  }
  __syncthreads();
  if (threadIdx.x <= C * 4)
    partial[(long)blockIdx.x * (C * 4 + 1) + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void deconv2x2_c1_reduce(const float* __restrict__ partial, float* __restrict__ dwt,
                                                           float* __restrict__ db, int nblocks, int nvals) {
  __shared__ float red[4][64];
  const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
  float s = 0.f;
  if (i < nvals && g < nblocks) dd_sum_strided(s, partial + (long)g * nvals + i, 4L * nvals, (nblocks - g + 3) / 4);
  red[g][i] = s;
  __syncthreads();
  if (g != 0 || i >= nvals) return;
  s = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  if (i < nvals - 1) dwt[i] = s; else db[0] = s;
}

int check_gdesc(const dd_gconv_desc* d, int max_cout = 64) {
  DD_REQUIRE(d != nullptr, DD_ERR_BAD_ARG, "gconv: NULL descriptor");
  DD_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->out_h > 0 && d->out_w > 0 && d->omem_h > 0 && d->omem_w > 0,
             DD_ERR_BAD_ARG, "gconv: non-positive size");
  DD_REQUIRE(d->cin > 0 && d->cin % 4 == 0 && d->in_coff % 4 == 0 && d->in_cstore % 4 == 0 && d->in_coff + d->cin <= d->in_cstore,
             DD_ERR_UNSUPPORTED, "gconv: input channel slice [%d,+%d) of %d must be 4-aligned", d->in_coff, d->cin, d->in_cstore);
  DD_REQUIRE(d->in_cstore < 1024, DD_ERR_UNSUPPORTED, "gconv: more than 1020 stored input channels");
  DD_REQUIRE(d->cout > 0 && d->cout <= max_cout && d->out_coff >= 0 && d->out_coff + d->cout <= d->out_cstore, DD_ERR_UNSUPPORTED,
             "gconv: output channel slice [%d,+%d) of %d (Cout <= %d)", d->out_coff, d->cout, d->out_cstore, max_cout);
  DD_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->div_h > 0 &&
                 d->div_w > 0 && d->ostride_h > 0 && d->ostride_w > 0 && d->ooff_h >= 0 && d->ooff_w >= 0,
             DD_ERR_BAD_ARG, "gconv: bad kernel geometry");
  DD_REQUIRE((long)d->kw * d->dil_w + d->pad_w < (1 << 20), DD_ERR_UNSUPPORTED, "gconv: tap offset too large");
  DD_REQUIRE((d->out_h - 1) * d->ostride_h + d->ooff_h < d->omem_h && (d->out_w - 1) * d->ostride_w + d->ooff_w < d->omem_w,
             DD_ERR_BAD_ARG, "gconv: output lattice leaves the output buffer");
  DD_REQUIRE((long)d->in_h * d->in_w * d->in_cstore * 4 < (1L << 31) && (long)d->omem_h * d->omem_w * d->out_cstore * 4 < (1L << 31),
             DD_ERR_UNSUPPORTED, "gconv: one image exceeds 2 GB");
  return 0;
}

int groups_of(const dd_gconv_desc* d) {
  const int nchunks = d->kh * d->kw * (d->cin / 4);
  const int g = (nchunks + 1) / 2;
  return (g + GU - 1) / GU * GU;
}

// Column tiles per wave: padded MFMA work x (1 + half the operand loads per MFMA) -- wider register tiles reuse the dy
// operand, too wide ones multiply zero columns.
int pick_nb(int nj, int no, int max_nb) {
  int best = 1;
  float best_score = 1e30f;
  for (int nb = 1; nb <= max_nb; ++nb) {
    const int padded = (nj + nb - 1) / nb * nb;
    const float score = padded * (1.f + 0.5f * (no + nb) / (float)(no * nb));
    if (score < best_score) { best_score = score; best = nb; }
  }
  return best;
}

// A wave owns NO dy channel tiles x NB (tap, cin) column tiles: NO + NB operand loads feed NO x NB MFMAs.
struct WgradPlan { int nto, no, nog, nj, nb, njg, njobs, nranges, blocks; };

WgradPlan wgrad_plan(const dd_gconv_desc* d) {
  WgradPlan p;
  p.nto = (d->cout + 31) / 32;
  p.no = p.nto <= 3 ? p.nto : 2;
  p.nog = (p.nto + p.no - 1) / p.no;
  p.nj = (d->kh * d->kw * d->cin + 31) / 32;
  constexpr int nb3 = 3, nb2 = 4;      // measured at bs 32 against (2, 2): up_conv_1 12.19 -> 11.39 ms, up_conv_2 6.37 -> 6.24
  p.nb = pick_nb(p.nj, p.no, p.no == 1 ? 4 : (p.no == 2 ? nb2 : nb3));
  p.njg = (p.nj + p.nb - 1) / p.nb;
  p.njobs = p.nog * p.njg;
  const long tiles = (long)d->batch * ((d->out_w + 31) / 32) * d->out_h;
  const long waves = 2L * dd_cu_budget_internal() * 4;   // two 4-wave workgroups per CU
  p.nranges = (int)max(1L, min(tiles, waves / p.njobs));
  p.blocks = (p.nranges * p.njobs + 3) / 4;
  return p;
}

// per-channel sum over the pixels of a dense NHWC buffer (bias gradient of a transposed conv whose weight
// gradient is taken in the role-swapped, waste-free form)
__global__ __launch_bounds__(256) void channel_sum_partial(const float* __restrict__ buf, float* __restrict__ partial,
                                                           long npix, int cstore, int coff, int cout) {
  __shared__ float red[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  long p = gw;
  for (; p + nw < npix; p += 2L * nw) {          // two pixels in flight per lane
    const float* px = buf + p * cstore + coff;
    const float* qx = px + (long)nw * cstore;
    if (lane < cout) { s0 += px[lane]; t0 += qx[lane]; }
    if (lane + 64 < cout) { s1 += px[lane + 64]; t1 += qx[lane + 64]; }
  }
  for (; p < npix; p += nw) {
    const float* px = buf + p * cstore + coff;
    if (lane < cout) s0 += px[lane];
    if (lane + 64 < cout) s1 += px[lane + 64];
  }
  red[wave][lane] = s0 + t0;
  red[wave][64 + lane] = s1 + t1;
  __syncthreads();
  if (threadIdx.x < 128)
    partial[(long)blockIdx.x * 128 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(1024) void channel_sum_final(const float* __restrict__ partial, float* __restrict__ out, int nrows,
                                                          int cout, int accumulate) {
  __shared__ float red[8][128];
  const int c = threadIdx.x & 127, g = threadIdx.x >> 7;
  float s = 0.f;
  for (int w = g; w < nrows; w += 8) s += partial[(long)w * 128 + c];
  red[g][c] = s;
  __syncthreads();
  if (g != 0 || c >= cout) return;
  s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += red[i][c];
  out[c] = accumulate ? out[c] + s : s;
}

// Streaming variant (cstore % 4 == 0): the buffer is read as one flat float4 stream, 16 bytes per lane and four loads
// in flight; the grid's thread count is a multiple of the quads per pixel, so a thread keeps ONE channel quad for its
// whole stride loop and needs no index arithmetic.  Partials are written channel-major so the final pass reads each
// channel's partials contiguously (one block per channel).  Summation order is fixed: deterministic.
__global__ __launch_bounds__(256) void channel_sum_stream(const f32x4* __restrict__ buf, float* __restrict__ partial_t, long nquads,
                                                          int quads_per_px, int coff, int cout) {
  __shared__ f32x4 red[256];
  const int bs = blockDim.x;
  const long t = (long)blockIdx.x * bs + threadIdx.x, T = (long)gridDim.x * bs;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  long i = t;
  for (; i + 3 * T < nquads; i += 4 * T) {
    const f32x4 a = buf[i], b = buf[i + T], c = buf[i + 2 * T], d = buf[i + 3 * T];
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; i < nquads; i += T) s0 += buf[i];
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if ((int)threadIdx.x < quads_per_px) {
    f32x4 s = red[threadIdx.x];
    for (int k = threadIdx.x + quads_per_px; k < bs; k += quads_per_px) s += red[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * threadIdx.x + j - coff;
      if (c >= 0 && c < cout) partial_t[(long)c * gridDim.x + blockIdx.x] = s[j];
    }
  }
}

__global__ __launch_bounds__(256) void channel_sum_stream_final(const float* __restrict__ partial_t, float* __restrict__ out, int nblocks,
                                                                int accumulate) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial_t[(long)c * nblocks + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + red[0] : red[0];
}


// ---------------------------------------------------------------------------------------------- ConvTranspose2d(32, 32, k2, s2): weight gradient
// dW[ci][co][a][b] = sum over input pixels p = (r, c) of x[p][ci] * g[2r + a][2c + b][co]; db[co] = sum of g.  Four GEMMs (one per output
// phase) with M = ci, N = co, K = pixels that share their A operand.  As four launches of the generic weight-gradient kernel each phase
// cost a 27 us launch plus two reduce launches of 25-30 us (0.33 ms for ss_deconv at bs 32).  Here: one launch, a wave walks a contiguous
// range of pixel pairs with four accumulator tiles (lane (h, m) supplies x[pixel h][ci = m] and the four g values of that pixel's 2 x 2
// output block at co = m: five 128-byte-coalesced loads per four MFMAs, requested a group of four pixel pairs ahead), one partial per
// wave, then a fixed-order fp64 second stage that also lays the result out as the layer's [ci][co][2][2] weight.
constexpr int D2W_WAVES = 1024;      // partials (256 workgroups of 4 waves)
constexpr int D2W_U = 4;             // pixel pairs per pipeline group

__global__ __launch_bounds__(256) void deconv2x2_c32_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                  float* __restrict__ part, long npix, int w, int gcs, int gcoff) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
  const int h = lane >> 5, m = lane & 31;
  const long npairs = (npix + 1) / 2;
  const long per = (npairs + D2W_WAVES - 1) / D2W_WAVES;
  const long p0 = min((long)gw * per, npairs), p1 = min(p0 + per, npairs);
  const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x, (int)(npix * 128));
  const __amdgpu_buffer_rsrc_t gs = dd_rsrc(g, (int)(npix * 4 * gcs * 4));
  f32x16 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
  float bsum = 0.f;
  // this lane's pixel: 2 * pair + h, as (row of the [batch * h] row list, column)
  long pix = 2 * p0 + h;
  long row = pix / w;
  int col = (int)(pix - row * w);
  auto offsets = [&](int& xo, int (&go)[4]) {
    const bool ok = pix < npix;
    xo = ok ? (int)(pix * 32 + m) * 4 : -16;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      go[p] = ok ? (int)((((2 * row + (p >> 1)) * (2L * w) + 2 * col + (p & 1)) * gcs + gcoff + m) * 4) : -16;
    pix += 2;
    col += 2;
    if (col >= w) { col -= w; row += 1; }
  };
  float av[D2W_U], gv[D2W_U][4];
  auto request = [&](int n) {      // the operands of the next n pixel pairs (n <= D2W_U; the others read zeros)
#pragma unroll
    for (int u = 0; u < D2W_U; ++u) {
      int xo, go[4];
      if (u < n) {
        offsets(xo, go);
      } else {
        xo = -16;
#pragma unroll
        for (int p = 0; p < 4; ++p) go[p] = -16;
      }
      av[u] = dd_bload1(xs, xo);
#pragma unroll
      for (int p = 0; p < 4; ++p) gv[u][p] = dd_bload1(gs, go[p]);
    }
  };
  long left = p1 - p0;
  if (left > 0) request((int)min(left, (long)D2W_U));
  while (left > 0) {
    float ac[D2W_U], gc[D2W_U][4];
#pragma unroll
    for (int u = 0; u < D2W_U; ++u) {
      ac[u] = av[u];
#pragma unroll
      for (int p = 0; p < 4; ++p) gc[u][p] = gv[u][p];
    }
    left -= D2W_U;
    if (left > 0) request((int)min(left, (long)D2W_U));      // the next group is on its way while this one is multiplied
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < D2W_U; ++u)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        acc[p] = DD_MFMA(ac[u], gc[u][p], acc[p]);
        bsum += gc[u][p];
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float* out = part + (long)gw * (4 * 1024 + 64);
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) out[(p * 16 + e) * 64 + lane] = acc[p][e];
  out[4096 + lane] = bsum;
}

// Second stage: block b < 64 owns the 64 consecutive elements t = 64 b + lane, block 64 the bias columns.  Its 16 thread groups each add
// a contiguous sixteenth of the D2W_WAVES partials (fp64, in order, sixteen loads in flight), LDS adds the groups in order: a fixed order
// whatever the grid.  (One thread per element walking all 1024 partials alone -- 17 workgroups on the chip -- was 53 us of load latency.)
__global__ __launch_bounds__(1024) void deconv2x2_c32_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double red[16][64];
  constexpr long STRIDE = 4 * 1024 + 64;
  constexpr int PER = D2W_WAVES / 16;
  static_assert(D2W_WAVES % 16 == 0, "sixteen equal ranges");
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const bool bias = blockIdx.x == 64;
  if (bias && !db) return;
  const int t = bias ? 4096 + lane : blockIdx.x * 64 + lane;      // element (p * 16 + e) * 64 + lane of phase p = 2a + b, or bias half-column
  double s = 0.0;
  dd_sum_strided(s, part + (long)g * PER * STRIDE + t, STRIDE, PER);
  red[g][lane] = s;
  __syncthreads();
  if (g != 0) return;
  s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i][lane];
  if (bias) {
    const double other = __shfl_xor(s, 32);      // the two half-columns of an output channel
    if (lane < 32) db[lane] = (float)(s + other);
  } else {
    const int l = t & 63, e = (t >> 6) & 15, p = t >> 10;
    const int ci = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), co = l & 31;
    dw[(ci * 32 + co) * 4 + p] = (float)s;
  }
}

}  // namespace

extern "C" {

int64_t dd_gconv_packed_floats(const dd_gconv_desc* d) {
  if (check_gdesc(d)) return -1;
  return (int64_t)groups_of(d) * ((d->cout + 31) / 32) * 256;
}

int dd_gconv_pack(const float* w, float* packed, const dd_gconv_desc* d, int64_t w_off, int64_t sn, int64_t sc, int32_t flip,
                  int32_t n_real, int32_t c_real, void* stream) {
  if (int rc = check_gdesc(d)) return rc;
  DD_REQUIRE(w && packed, DD_ERR_BAD_ARG, "gconv_pack: NULL pointer");
  DD_REQUIRE(n_real > 0 && n_real <= d->cout && c_real > 0 && c_real <= d->cin, DD_ERR_BAD_ARG, "gconv_pack: n_real/c_real");
  const int ng = groups_of(d), nt = (d->cout + 31) / 32;
  const long total = (long)ng * nt * 256;
  hipLaunchKernelGGL(gconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, ng,
                     nt, d->cin / 4, d->kh * d->kw, (long)w_off, (long)sn, (long)sc, flip, n_real, c_real);
  DD_LAUNCH_CHECK("gconv_pack");
  return 0;
}

int dd_gconv_fwd(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                 int32_t epilogue, void* stream) {
  if (int rc = check_gdesc(d)) return rc;
  DD_REQUIRE(x && packed && y, DD_ERR_BAD_ARG, "gconv_fwd: NULL pointer");
  DD_REQUIRE(epilogue >= DD_EPI_NONE && epilogue <= DD_EPI_BIAS_SIGMOID, DD_ERR_BAD_ARG, "gconv_fwd: epilogue %d", epilogue);
  DD_REQUIRE(epilogue != DD_EPI_RELU_MASK || mask, DD_ERR_BAD_ARG, "gconv_fwd: RELU_MASK needs a mask");
  DD_REQUIRE(!(epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU || epilogue == DD_EPI_BIAS_SIGMOID) || bias, DD_ERR_BAD_ARG,
             "gconv_fwd: bias epilogue needs a bias");
  hipStream_t st = (hipStream_t)stream;
  const int ng = groups_of(d), nt = (d->cout + 31) / 32;
  const bool div = d->div_h > 1 || d->div_w > 1;
  const long tiles = (long)d->batch * ((d->out_w + 31) / 32) * d->out_h;
  const int grid = (int)max(1L, min((long)dd_cu_budget_internal(), (tiles + 7) / 8));   // one 8-wave workgroup per CU, all resident
  const size_t lds = (size_t)2 * ng * sizeof(int2);
  DD_REQUIRE(lds <= 64 * 1024, DD_ERR_UNSUPPORTED, "gconv_fwd: tap table of %zu bytes", lds);
  // sets (GU chunk pairs) per tap when a set never straddles taps, else 0 = no tap skipping
  const int pairs_per_tap = d->cin / 8;
  const int spt = (d->cin % 8 == 0 && pairs_per_tap % GU == 0) ? pairs_per_tap / GU : 0;
#define DD_GF(NT, DIV) hipLaunchKernelGGL((gconv_fwd_kernel<NT, DIV>), dim3(grid), dim3(512), lds, st, x, packed, bias, mask, y, *d, ng, epilogue, spt)
  if (nt == 1) { if (div) DD_GF(1, true); else DD_GF(1, false); }
  else { if (div) DD_GF(2, true); else DD_GF(2, false); }
#undef DD_GF
  DD_LAUNCH_CHECK("gconv_fwd");
  return 0;
}

int64_t dd_gconv_wgrad_workspace_bytes(const dd_gconv_desc* d) {
  if (check_gdesc(d, 96)) return -1;
  const WgradPlan p = wgrad_plan(d);
  return ((int64_t)p.nranges * p.njobs * p.no * p.nb * 1024 + (int64_t)p.nranges * p.nog * p.no * 64) * 4;
}

int dd_gconv_wgrad(const float* x, const float* dy, float* dw, float* dbias, const dd_gconv_desc* d, int64_t w_off, int64_t sn,
                   int64_t sc, int32_t flip, int32_t n_real, int32_t c_real, int32_t accumulate, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  if (int rc = check_gdesc(d, 96)) return rc;
  DD_REQUIRE(x && dy && dw && workspace, DD_ERR_BAD_ARG, "gconv_wgrad: NULL pointer");
  DD_REQUIRE(d->div_h == 1 && d->div_w == 1, DD_ERR_UNSUPPORTED, "gconv_wgrad: divisibility mode has no weight gradient");
  DD_REQUIRE(workspace_bytes >= dd_gconv_wgrad_workspace_bytes(d), DD_ERR_WORKSPACE, "gconv_wgrad: workspace too small");
  DD_REQUIRE(n_real > 0 && n_real <= d->cout && c_real > 0 && c_real <= d->cin, DD_ERR_BAD_ARG, "gconv_wgrad: n_real/c_real");
  hipStream_t st = (hipStream_t)stream;
  const WgradPlan p = wgrad_plan(d);
  float* part = (float*)workspace;
  float* bpart = part + (size_t)p.nranges * p.njobs * p.no * p.nb * 1024;
#define DD_GW(NO, NB) hipLaunchKernelGGL((gconv_wgrad_kernel<NO, NB>), dim3(p.blocks), dim3(256), 0, st, x, dy, part, bpart, *d, p.njg, p.nog, p.nranges)
  switch (p.no * 10 + p.nb) {
    case 11: DD_GW(1, 1); break;
    case 12: DD_GW(1, 2); break;
    case 13: DD_GW(1, 3); break;
    case 14: DD_GW(1, 4); break;
    case 21: DD_GW(2, 1); break;
    case 22: DD_GW(2, 2); break;
    case 23: DD_GW(2, 3); break;
    case 24: DD_GW(2, 4); break;
    case 31: DD_GW(3, 1); break;
    case 32: DD_GW(3, 2); break;
    case 33: DD_GW(3, 3); break;
    default:      // a plan this switch has no kernel for: refuse instead of launching one built for another tiling
      DD_REQUIRE(false, DD_ERR_UNSUPPORTED, "gconv_wgrad: no kernel for the tiling (%d, %d)", p.no, p.nb);
  }
#undef DD_GW
  DD_LAUNCH_CHECK("gconv_wgrad");
  const long per = (long)p.njobs * p.no * p.nb * 1024;
  const int wblocks = (int)(per / 256), bblocks = dbias ? p.nto : 0;
  hipLaunchKernelGGL(gconv_wgrad_reduce, dim3((unsigned)(wblocks + bblocks)), dim3(1024), 0, st, part, dw, wblocks, bpart, dbias,
                     p.nog * p.no, accumulate & 2, p.nranges, p.njobs, p.no, p.nb, p.njg, d->kh * d->kw, d->cin, (long)w_off, (long)sn,
                     (long)sc, flip, n_real, c_real, accumulate & 1);
  DD_LAUNCH_CHECK("gconv_wgrad_reduce");
  return 0;
}

int64_t dd_channel_sum_workspace_bytes(void) { return (int64_t)DD_NUM_CU * 4 * 128 * 4; }   // one 128-channel row per block

int dd_channel_sum(const float* buf, float* out, int64_t npix, int32_t cstore, int32_t coff, int32_t cout, int32_t accumulate,
                   void* workspace, void* stream) {
  DD_REQUIRE(buf && out && workspace && npix > 0 && cstore > 0 && coff >= 0 && cout > 0 && coff + cout <= cstore, DD_ERR_BAD_ARG,
             "channel_sum: bad argument");
  DD_REQUIRE(cout <= 128, DD_ERR_UNSUPPORTED, "channel_sum: more than 128 channels");
  if (cstore % 4 == 0 && cstore <= 256 && (uintptr_t)buf % 16 == 0) {
    const int q = cstore / 4, bs = (256 / q) * q;
    const long nquads = (long)npix * q;
    const int grid = (int)max(1L, min((nquads + 4L * bs - 1) / (4L * bs), (long)DD_NUM_CU * 4));
    hipLaunchKernelGGL(channel_sum_stream, dim3(grid), dim3(bs), 0, (hipStream_t)stream, (const f32x4*)buf, (float*)workspace, nquads, q,
                       coff, cout);
    DD_LAUNCH_CHECK("channel_sum");
    hipLaunchKernelGGL(channel_sum_stream_final, dim3(cout), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, grid, accumulate);
    DD_LAUNCH_CHECK("channel_sum final");
    return 0;
  }
  const int grid = (int)min((npix + 3) / 4, (long)DD_NUM_CU * 4);
  hipLaunchKernelGGL(channel_sum_partial, dim3(grid), dim3(256), 0, (hipStream_t)stream, buf, (float*)workspace, (long)npix, cstore,
                     coff, cout);
  DD_LAUNCH_CHECK("channel_sum");
  hipLaunchKernelGGL(channel_sum_final, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, out, grid, cout,
                     accumulate);
  DD_LAUNCH_CHECK("channel_sum final");
  return 0;
}

int dd_deconv2x2_c32_fwd_slice(const float* x, const float* wt, const float* bias, float* out, int32_t batch, int32_t h, int32_t w, int32_t relu,
                               int32_t out_cstore, int32_t out_coff, void* stream) {
  DD_REQUIRE(x && wt && out && batch > 0 && h > 0 && w >= 32, DD_ERR_BAD_ARG, "deconv2x2_c32_fwd: bad argument (width at least 32)");
  DD_REQUIRE(out_cstore >= 32 && out_coff >= 0 && out_coff + 32 <= out_cstore, DD_ERR_BAD_ARG, "deconv2x2_c32_fwd: output channel slice");
  const long npix = (long)batch * h * w;
  DD_REQUIRE(npix * 128 < (1L << 31), DD_ERR_UNSUPPORTED, "deconv2x2_c32_fwd: input exceeds 2 GB");
  const long ntile = (npix + 31) / 32;
  hipLaunchKernelGGL(deconv2x2_c32_fwd_kernel, dim3((unsigned)min((ntile + 3) / 4, (long)DD_NUM_CU * 8)), dim3(256), 0, (hipStream_t)stream, x, wt,
                     bias, out, npix, w, relu, out_cstore, out_coff);
  DD_LAUNCH_CHECK("deconv2x2_c32_fwd");
  return 0;
}

int64_t dd_deconv2x2_c32_wgrad_workspace_bytes(void) { return (int64_t)D2W_WAVES * (4 * 1024 + 64) * 4; }

int dd_deconv2x2_c32_wgrad(const float* x, const float* g, float* dw, float* db, int32_t batch, int32_t h, int32_t w, int32_t g_cstore,
                           int32_t g_coff, void* workspace, int64_t workspace_bytes, void* stream) {
  DD_REQUIRE(x && g && dw && workspace && batch > 0 && h > 0 && w >= 2, DD_ERR_BAD_ARG, "deconv2x2_c32_wgrad: bad argument");
  DD_REQUIRE(g_cstore >= 32 && g_coff >= 0 && g_coff + 32 <= g_cstore, DD_ERR_BAD_ARG, "deconv2x2_c32_wgrad: gradient channel slice");
  DD_REQUIRE(workspace_bytes >= dd_deconv2x2_c32_wgrad_workspace_bytes(), DD_ERR_BAD_ARG, "deconv2x2_c32_wgrad: workspace too small");
  const long npix = (long)batch * h * w;
  DD_REQUIRE(npix * 128 < (1L << 31) && npix * 16 * g_cstore < (1L << 31), DD_ERR_UNSUPPORTED, "deconv2x2_c32_wgrad: a tensor exceeds 2 GB");
  hipLaunchKernelGGL(deconv2x2_c32_wgrad_kernel, dim3(D2W_WAVES / 4), dim3(256), 0, (hipStream_t)stream, x, g, (float*)workspace, npix, w,
                     g_cstore, g_coff);
  DD_LAUNCH_CHECK("deconv2x2_c32_wgrad");
  hipLaunchKernelGGL(deconv2x2_c32_wgrad_reduce, dim3(65), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, dw,
                     db);
  DD_LAUNCH_CHECK("deconv2x2_c32_wgrad reduce");
  return 0;
}

int dd_deconv2x2_c32_fwd(const float* x, const float* wt, const float* bias, float* out, int32_t batch, int32_t h, int32_t w, int32_t relu,
                         void* stream) {
  return dd_deconv2x2_c32_fwd_slice(x, wt, bias, out, batch, h, w, relu, 32, 0, stream);
}

int dd_conv1x1_c32_c3_nchw(const float* x, const float* wt, const float* bias, float* out, int32_t batch, int32_t h, int32_t w, void* stream) {
  DD_REQUIRE(x && wt && out && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "conv1x1_c32_c3_nchw: bad argument");
  const long plane = (long)h * w, npix = plane * batch;
  hipLaunchKernelGGL(conv1x1_c32_c3_nchw_kernel, dim3((unsigned)min((npix + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0, (hipStream_t)stream,
                     x, wt, bias, out, plane, npix);
  DD_LAUNCH_CHECK("conv1x1_c32_c3_nchw");
  return 0;
}

int64_t dd_deconv2x2_c1_workspace_bytes(int32_t c) { return (int64_t)DD_NUM_CU * 4 * (c * 4 + 1) * 4; }

int dd_deconv2x2_c1_fwd(const float* x, const float* wt, const float* bias, float* probs, int32_t batch, int32_t h, int32_t w,
                        int32_t c, void* stream) {
  DD_REQUIRE(x && wt && bias && probs && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "deconv2x2_c1_fwd: bad argument");
  DD_REQUIRE(c == 8, DD_ERR_UNSUPPORTED, "deconv2x2_c1: Cin %d (the box heads end in 8 -> 1)", c);
  const long npix = (long)batch * h * w;
  hipLaunchKernelGGL(deconv2x2_c1_fwd_kernel<8>, dim3((unsigned)min((npix + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                     (hipStream_t)stream, x, wt, bias, probs, npix, w);
  DD_LAUNCH_CHECK("deconv2x2_c1_fwd");
  return 0;
}

int dd_deconv2x2_c1_bwd(const float* x, const float* wt, const float* probs, const float* dprobs, float* dx, float* dwt,
                        float* dbias, int32_t batch, int32_t h, int32_t w, int32_t c, void* workspace, void* stream) {
  DD_REQUIRE(x && wt && probs && dprobs && dx && dwt && dbias && workspace && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG,
             "deconv2x2_c1_bwd: bad argument");
  DD_REQUIRE(c == 8, DD_ERR_UNSUPPORTED, "deconv2x2_c1: Cin %d (the box heads end in 8 -> 1)", c);
  const long npix = (long)batch * h * w;
  const int grid = (int)min((npix + 255) / 256, (long)DD_NUM_CU * 4);
  hipLaunchKernelGGL(deconv2x2_c1_bwd_kernel<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, wt, probs, dprobs, dx,
                     (float*)workspace, npix, w);
  DD_LAUNCH_CHECK("deconv2x2_c1_bwd");
  hipLaunchKernelGGL(deconv2x2_c1_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dwt, dbias, grid,
                     c * 4 + 1);
  DD_LAUNCH_CHECK("deconv2x2_c1_reduce");
  return 0;
}

int dd_view_to_nhwc4(const float* views, float* out, int32_t batch, int32_t height, int32_t width, int32_t view,
                     int32_t transform, void* stream) {
  DD_REQUIRE(views && out && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "view_to_nhwc4: bad argument");
  DD_REQUIRE(view >= 0 && view < 6 && transform >= 0 && transform <= 3, DD_ERR_BAD_ARG, "view_to_nhwc4: view %d transform %d", view, transform);
  const long total = (long)batch * height * width;
  hipLaunchKernelGGL(view_to_nhwc4_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                     (hipStream_t)stream, views, (f32x4*)out, batch, height, width, view, transform);
  DD_LAUNCH_CHECK("view_to_nhwc4");
  return 0;
}

int dd_view_to_nhwc4_ptrs(const float* const* sample_ptrs, float* out, int32_t batch, int32_t height, int32_t width, int32_t view,
                          int32_t transform, void* stream) {
  DD_REQUIRE(sample_ptrs && out && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "view_to_nhwc4_ptrs: bad argument");
  DD_REQUIRE(view >= 0 && view < 6 && transform >= 0 && transform <= 3, DD_ERR_BAD_ARG, "view_to_nhwc4_ptrs: view %d transform %d", view, transform);
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    ViewSamplePtrs tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "view_to_nhwc4_ptrs: null sample pointer");
    const long total = (long)nb * height * width;
    hipLaunchKernelGGL(view_to_nhwc4_ptrs_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                       (hipStream_t)stream, tab, (f32x4*)out + (long)b0 * height * width, nb, height, width, view, transform);
    DD_LAUNCH_CHECK("view_to_nhwc4_ptrs");
  }
  return 0;
}

int dd_view_to_nhwc4_u8_ptrs(const unsigned char* const* sample_ptrs, float* out, int32_t batch, int32_t height, int32_t width,
                             int32_t view, int32_t transform, void* stream) {
  DD_REQUIRE(sample_ptrs && out && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "view_to_nhwc4_u8_ptrs: bad argument");
  DD_REQUIRE(view >= 0 && view < 6 && transform >= 0 && transform <= 3, DD_ERR_BAD_ARG, "view_to_nhwc4_u8_ptrs: view %d transform %d", view, transform);
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    ViewSamplePtrsU8 tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "view_to_nhwc4_u8_ptrs: null sample pointer");
    const long total = (long)nb * height * width;
    hipLaunchKernelGGL(view_to_nhwc4_u8_ptrs_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                       (hipStream_t)stream, tab, (f32x4*)out + (long)b0 * height * width, nb, height, width, view, transform);
    DD_LAUNCH_CHECK("view_to_nhwc4_u8_ptrs");
  }
  return 0;
}

int dd_copy_channels(const float* src, float* dst, int64_t npix, int32_t channels, int32_t src_cstore, int32_t src_coff, int32_t dst_cstore,
                     int32_t dst_coff, void* stream) {
  DD_REQUIRE(src && dst && npix > 0 && channels > 0, DD_ERR_BAD_ARG, "copy_channels: bad argument");
  DD_REQUIRE(channels % 4 == 0 && src_cstore % 4 == 0 && src_coff % 4 == 0 && dst_cstore % 4 == 0 && dst_coff % 4 == 0 &&
                 src_coff + channels <= src_cstore && dst_coff + channels <= dst_cstore,
             DD_ERR_UNSUPPORTED, "copy_channels: channel slices must be 4-aligned and inside their buffers");
  const long total = npix * (channels / 4);
  hipLaunchKernelGGL(copy_channels_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, (long)npix, channels / 4, src_cstore, src_coff, dst_cstore, dst_coff);
  DD_LAUNCH_CHECK("copy_channels");
  return 0;
}

int dd_copy_channels_window(const float* src, float* dst, int32_t batch, int32_t h, int32_t w, int32_t channels, int32_t src_mem_h,
                            int32_t src_mem_w, int32_t src_y0, int32_t src_x0, int32_t src_cstore, int32_t src_coff, int32_t dst_mem_h,
                            int32_t dst_mem_w, int32_t dst_y0, int32_t dst_x0, int32_t dst_cstore, int32_t dst_coff, void* stream) {
  DD_REQUIRE(src && dst && batch > 0 && h > 0 && w > 0 && channels > 0, DD_ERR_BAD_ARG, "copy_channels_window: bad argument");
  DD_REQUIRE(channels % 4 == 0 && src_cstore % 4 == 0 && src_coff % 4 == 0 && dst_cstore % 4 == 0 && dst_coff % 4 == 0 && src_coff >= 0 &&
                 dst_coff >= 0 && src_coff + channels <= src_cstore && dst_coff + channels <= dst_cstore,
             DD_ERR_UNSUPPORTED, "copy_channels_window: channel slices must be 4-aligned and inside their buffers");
  DD_REQUIRE(src_y0 >= 0 && src_x0 >= 0 && src_y0 + h <= src_mem_h && src_x0 + w <= src_mem_w && dst_y0 >= 0 && dst_x0 >= 0 &&
                 dst_y0 + h <= dst_mem_h && dst_x0 + w <= dst_mem_w,
             DD_ERR_BAD_ARG, "copy_channels_window: a window leaves its buffer");
  const long total = (long)batch * h * w * (channels / 4);
  hipLaunchKernelGGL(copy_channels_window_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, batch, h, w, channels / 4, src_mem_h, src_mem_w, src_y0, src_x0, src_cstore, src_coff,
                     dst_mem_h, dst_mem_w, dst_y0, dst_x0, dst_cstore, dst_coff);
  DD_LAUNCH_CHECK("copy_channels_window");
  return 0;
}

int dd_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  DD_REQUIRE(a && b && out && n > 0 && n % 4 == 0, DD_ERR_BAD_ARG, "add: bad argument");
  const long n4 = n / 4;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)min((n4 + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)a, (const f32x4*)b, (f32x4*)out, n4);
  DD_LAUNCH_CHECK("add");
  return 0;
}

}  // extern "C"
