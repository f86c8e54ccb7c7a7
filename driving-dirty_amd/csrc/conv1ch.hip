// rm_conv_1 of RoadMapBoxesMergingCNN (spatial_bb/components.py:80,148): relu(Conv2d(1 -> 32, k7, stride 3, dilation 3, padding 1))
// on the road map, as a DENSE 7x7 convolution on the pixels it reads (dd_subsample_nhwc4: s[u][v] = rm[3u - 1][3v - 1]).
//
// One input channel: the K dimension of the GEMM is the TAPS.  The generic engine pads the channel to 4 and issues 98
// v_mfma_f32_32x32x2_f32 per 32-pixel x 32-channel tile, three quarters of them on zeros; here a tap row is 4 column pairs
// (kx = 2j + h, the 8th column has zero weights), 28 MFMAs per tile, the weights live in 28 registers and the A operand is
// one ds_read_b32 per MFMA out of a 7-row LDS patch of the image.
//
// Weight gradient: dW[co][ky][kx] = sum over pixels of g[px][co] * s[px + (ky, kx)], M = 32 output channels, N = taps (two
// column tiles: taps 0..31, then 32..48 + a column of ones that yields the bias gradient), K = pixels: per pixel pair one
// 256-byte row of g from memory, two ds_read_b32 and two MFMAs.  Per-workgroup partial sums, fp64 fixed-order second stage.
#include "dd_common.h"

namespace {

constexpr int C1_K = 7, C1_T = 49, C1_CO = 32;
constexpr int C1_PITCH = 352;      // floats of a patch row: image width (<= 320) + a 32-pixel tile's overhang
constexpr int C1_MAXW = 320;

// "Phase-major" layout of a dense NHWC tensor T [B][oh][ow][32] for a dilation-3 consumer (rm_conv_2, components.py:81,131): the nine residue
// classes (i mod 3, j mod 3) as nine images, P[b * 9 + (i % 3) * 3 + j % 3][i / 3][j / 3][:] = T[b][i][j][:], each ph x pw = ceil(oh / 3) x
// ceil(ow / 3) pixels; cells past the last row / column of a class are zero.  On P a 3x3 convolution with dilation 3 is nine plain 3x3
// convolutions -- which run on the encoder c2 layer's Winograd kernels.
__device__ __forceinline__ long c1_phase_pixel(int b, int i, int j, int ph, int pw) {
  return (((long)b * 9 + (i % 3) * 3 + (j % 3)) * ph + i / 3) * pw + j / 3;
}

// ---- forward: one workgroup (4 waves) per output row.  PHASE: y (and the optional sign words) in the phase-major layout above.
template <bool PHASE>
__global__ __launch_bounds__(256) void conv1ch_fwd_kernel(const float* __restrict__ s4, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, unsigned* __restrict__ bits,
                                                          int sh, int sw, int relu) {
  __shared__ float patch[C1_K][C1_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, n = lane & 31;
  const int oh = sh - (C1_K - 1), ow = sw - (C1_K - 1);
  const int b = blockIdx.x / oh, oy = blockIdx.x - b * oh;
  // weights of this lane: column pair j of tap row ky -> w[co = n][ky][2j + h] (zero for the 8th column)
  float bw[C1_K][4];
#pragma unroll
  for (int ky = 0; ky < C1_K; ++ky)
#pragma unroll
    for (int j = 0; j < 4; ++j) bw[ky][j] = (2 * j + h < C1_K) ? w[n * C1_T + ky * C1_K + 2 * j + h] : 0.f;
  const float bv = bias ? bias[n] : 0.f;
  for (int i = tid; i < C1_K * C1_PITCH; i += 256) {
    const int r = i / C1_PITCH, c = i - r * C1_PITCH;
    patch[r][c] = c < sw ? s4[(((long)b * sh + oy + r) * sw + c) * 4] : 0.f;
  }
  __syncthreads();
  const int ntile = (ow + 31) >> 5;
  for (int t = wave; t < ntile; t += 4) {
    const int ox0 = t * 32;
    float a[C1_K][4];
#pragma unroll
    for (int ky = 0; ky < C1_K; ++ky)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[ky][j] = patch[ky][ox0 + n + 2 * j + h];
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < C1_K; ++ky)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = DD_MFMA(a[ky][j], bw[ky][j], acc);
    float* yr = y + (((long)b * oh + oy) * ow) * C1_CO;
    const int ph = (oh + 2) / 3, pw = (ow + 2) / 3;
    unsigned word = 0;                                        // PHASE: sign word of pixel ox0 + (lane & 31), kept by lanes 0 .. 31
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ox = ox0 + dd_acc_row(e, lane);
      float v = acc[e] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (PHASE) {
        if (ox < ow) __builtin_nontemporal_store(v, y + c1_phase_pixel(b, oy, ox, ph, pw) * C1_CO + n);
        if (bits) {      // lanes 0-31: the 32 channels of pixel i0 = (e & 3) + 8 (e >> 2), lanes 32-63: those of i0 + 4
          const unsigned long long bal = __ballot(v > 0.f);
          const int i0 = (e & 3) + 8 * (e >> 2);
          if (lane == i0) word = (unsigned)bal;
          if (lane == i0 + 4) word = (unsigned)(bal >> 32);
        }
      } else {
        if (ox < ow) __builtin_nontemporal_store(v, yr + ox * C1_CO + n);
      }
    }
    if (PHASE && bits && lane < 32 && ox0 + lane < ow) bits[c1_phase_pixel(b, oy, ox0 + lane, ph, pw)] = word;
  }
  if (PHASE) {      // the padding cells of the classes this row belongs to: past the last column, and the row below the class's last one
    const int ph = (oh + 2) / 3, pw = (ow + 2) / 3;
    const int a = oy % 3, srow = oy / 3;
    const bool below = oy + 3 >= oh && srow + 1 < ph;       // this is the last row of its class and the class is one row short
    for (int i = tid; i < 3 * (pw + 1) * 8; i += 256) {      // (column class bb, cell, 16-byte piece): cell pw = "the last column of this row"
      const int q = i & 7, cell = (i >> 3) % (pw + 1), bb = (i >> 3) / (pw + 1);
      const long img = ((long)b * 9 + a * 3 + bb) * ph;
      const bool col_pad = 3 * (pw - 1) + bb >= ow;          // class bb has no pixel in column pw - 1
      if (cell == pw) {
        if (col_pad) {
          *(f32x4*)(y + ((img + srow) * pw + pw - 1) * C1_CO + 4 * q) = f32x4{0.f, 0.f, 0.f, 0.f};
          if (bits && q == 0) bits[(img + srow) * pw + pw - 1] = 0u;
        }
      } else if (below) {
        *(f32x4*)(y + ((img + srow + 1) * pw + cell) * C1_CO + 4 * q) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bits && q == 0) bits[(img + srow + 1) * pw + cell] = 0u;
      }
    }
  }
}

// ---- weight gradient: persistent workgroups walk output rows; the 4 waves split a row's pixel pairs
template <bool PHASE>
__global__ __launch_bounds__(256) void conv1ch_wgrad_kernel(const float* __restrict__ s4, const float* __restrict__ g,
                                                            float* __restrict__ part, int batch, int sh, int sw) {
  __shared__ float patch[C1_K + 1][C1_PITCH];      // row 7: ones (the bias column, and what the padding columns read)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, n = lane & 31;
  const int oh = sh - (C1_K - 1), ow = sw - (C1_K - 1);
  // B operand of this lane: column tile 0 = tap n, column tile 1 = tap 32 + n (< 49), else the ones row
  int boff[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int t = 32 * nt + n;
    boff[nt] = t < C1_T ? ((t / C1_K) * C1_PITCH + t % C1_K + h) * 4 : (C1_K * C1_PITCH + h) * 4;
  }
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
  for (int i = tid; i < C1_PITCH; i += 256) patch[C1_K][i] = 1.f;
  const int npair = (ow + 1) >> 1;
  for (long row = blockIdx.x; row < (long)batch * oh; row += gridDim.x) {
    const int b = (int)(row / oh), oy = (int)(row - (long)b * oh);
    __syncthreads();
    for (int i = tid; i < C1_K * C1_PITCH; i += 256) {
      const int r = i / C1_PITCH, c = i - r * C1_PITCH;
      patch[r][c] = c < sw ? s4[(((long)b * sh + oy + r) * sw + c) * 4] : 0.f;
    }
    __syncthreads();
    const float* gr = g + ((long)b * oh + oy) * ow * C1_CO;
    const char* pb = (const char*)&patch[0][0];
    // pairs wave, wave + 4, ...: eight at a time, loads first (four: 225 us at bs 32 -- 2 MB in flight on the chip, 1.25 TB/s of the 281 MB of g)
    constexpr int NU = 8;
    for (int p0 = wave; p0 < npair; p0 += 4 * NU) {
      float av[NU], b0[NU], b1[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int p = p0 + 4 * u, px = 2 * p + h;
        if (PHASE)
          av[u] = (p < npair && px < ow) ? g[c1_phase_pixel(b, oy, px, (oh + 2) / 3, (ow + 2) / 3) * C1_CO + n] : 0.f;
        else
          av[u] = (p < npair && px < ow) ? gr[px * C1_CO + n] : 0.f;      // a pixel past the row end contributes nothing
        const int po = min(p, npair - 1) * 8;
        b0[u] = *(const float*)(pb + boff[0] + po);
        b1[u] = *(const float*)(pb + boff[1] + po);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        acc[0] = DD_MFMA(av[u], b0[u], acc[0]);
        acc[1] = DD_MFMA(av[u], b1[u], acc[1]);
      }
    }
  }
  // the four waves' sums meet in LDS (wave order), one partial image per workgroup goes to the second stage
  __shared__ float wsum[4][2 * 16 * 64];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) wsum[wave][(nt * 16 + e) * 64 + lane] = acc[nt][e];
  __syncthreads();
  for (int i = tid; i < 2 * 16 * 64; i += 256) part[(long)blockIdx.x * (2 * 16 * 64) + i] = ((wsum[0][i] + wsum[1][i]) + wsum[2][i]) + wsum[3][i];
}

// one block per tap (or bias) column: 32 output channels x 16 slices of the workgroups' partials; a slice is summed in
// fp64 in index order, the 16 slice sums in slice order -- a fixed order whatever the grid
__global__ __launch_bounds__(512) void conv1ch_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nwg) {
  __shared__ double red[16][C1_CO];
  const int col = blockIdx.x;               // 0..48 taps, 49 bias
  const int co = threadIdx.x & 31, slice = threadIdx.x >> 5;
  // D[row = co][col]: register e, lane l with dd_acc_row(e, l) == co and (l & 31) == col & 31
  const int nt = col >> 5, n = col & 31;
  const int hh = (co >> 2) & 1, e = (co & 3) + 4 * (co >> 3);
  const int lane = 32 * hh + n;
  double s = 0.0;
  if (slice < nwg) dd_sum_strided(s, part + (((long)slice * 2 + nt) * 16 + e) * 64 + lane, 16L * 2 * 16 * 64, (nwg - slice + 15) / 16);
  red[slice][co] = s;
  __syncthreads();
  if (slice == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][co];
    if (col < C1_T) dw[co * C1_T + col] = (float)t;
    else if (db) db[co] = (float)t;
  }
}

// dst [B][oh][ow][dst_cstore] channels [dst_coff, +32) = the phase-major image src [B * 9][sph][spw][32], read `off` cells in from its
// corner (the interior of a padding-1 convolution's output): dst[b][i][j] = src[b * 9 + (i % 3) * 3 + j % 3][i / 3 + off][j / 3 + off]
__global__ __launch_bounds__(256) void phase3_scatter_kernel(const f32x4* __restrict__ src, float* __restrict__ dst, int B, int oh, int ow,
                                                             int sph, int spw, int off, int dst_cstore, int dst_coff) {
  const long total = (long)B * oh * ow * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 7);
    long p = idx >> 3;
    const int j = (int)(p % ow);
    p /= ow;
    const int i = (int)(p % oh);
    const long b = p / oh;
    const long sp = ((b * 9 + (i % 3) * 3 + (j % 3)) * sph + i / 3 + off) * spw + j / 3 + off;
    *(f32x4*)(dst + ((b * oh + i) * ow + j) * dst_cstore + dst_coff + 4 * q) = __builtin_nontemporal_load(src + sp * 8 + q);
  }
}

// ... and back: dst [B * 9][dph][dpw][32] (every cell written) = src's pixel where the cell, read `off` cells in, is one, zero elsewhere
__global__ __launch_bounds__(256) void phase3_gather_kernel(const float* __restrict__ src, f32x4* __restrict__ dst, int B, int oh, int ow,
                                                            int dph, int dpw, int off, int src_cstore, int src_coff) {
  const long total = (long)B * 9 * dph * dpw * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx & 7);
    long p = idx >> 3;
    const int t = (int)(p % dpw);
    p /= dpw;
    const int sr = (int)(p % dph);
    p /= dph;
    const int cls = (int)(p % 9);
    const long b = p / 9;
    const int i = 3 * (sr - off) + cls / 3, j = 3 * (t - off) + cls % 3;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sr >= off && t >= off && i < oh && j < ow) v = *(const f32x4*)(src + ((b * oh + i) * ow + j) * src_cstore + src_coff + 4 * q);
    __builtin_nontemporal_store(v, dst + idx);
  }
}

}  // namespace

extern "C" {

int dd_conv1ch_fwd_phase3(const float* taps4, const float* w, const float* bias, float* y_phase, uint32_t* relu_bits, int32_t batch,
                          int32_t sh, int32_t sw, int32_t relu, void* stream) {
  DD_REQUIRE(taps4 && w && y_phase && batch > 0, DD_ERR_BAD_ARG, "conv1ch_fwd_phase3: bad argument");
  DD_REQUIRE(sh >= C1_K && sw >= C1_K && sw <= C1_MAXW, DD_ERR_UNSUPPORTED, "conv1ch_fwd_phase3: image %dx%d (width at most %d)", sh, sw, C1_MAXW);
  DD_REQUIRE(((uintptr_t)y_phase & 15) == 0, DD_ERR_BAD_ARG, "conv1ch_fwd_phase3: y_phase must be 16-byte aligned");
  const int oh = sh - (C1_K - 1);
  hipLaunchKernelGGL(conv1ch_fwd_kernel<true>, dim3((unsigned)(batch * oh)), dim3(256), 0, (hipStream_t)stream, taps4, w, bias, y_phase,
                     (unsigned*)relu_bits, sh, sw, relu);
  DD_LAUNCH_CHECK("conv1ch_fwd_phase3");
  return 0;
}

int dd_conv1ch_wgrad_phase3(const float* taps4, const float* g_phase, float* dw, float* dbias, int32_t batch, int32_t sh, int32_t sw,
                            void* workspace, void* stream) {
  DD_REQUIRE(taps4 && g_phase && dw && workspace && batch > 0, DD_ERR_BAD_ARG, "conv1ch_wgrad_phase3: bad argument");
  DD_REQUIRE(sh >= C1_K && sw >= C1_K && sw <= C1_MAXW, DD_ERR_UNSUPPORTED, "conv1ch_wgrad_phase3: image %dx%d (width at most %d)", sh, sw, C1_MAXW);
  const long rows = (long)batch * (sh - (C1_K - 1));
  const int nwg = (int)max(1L, min((long)dd_cu_budget_internal() * 2, rows));
  hipLaunchKernelGGL(conv1ch_wgrad_kernel<true>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, taps4, g_phase, (float*)workspace, batch, sh, sw);
  hipLaunchKernelGGL(conv1ch_wgrad_reduce, dim3(C1_T + 1), dim3(512), 0, (hipStream_t)stream, (const float*)workspace, dw, dbias, nwg);
  DD_LAUNCH_CHECK("conv1ch_wgrad_phase3");
  return 0;
}

int dd_phase3_scatter(const float* src_phase, float* dst, int32_t batch, int32_t oh, int32_t ow, int32_t src_ph, int32_t src_pw, int32_t off,
                      int32_t dst_cstore, int32_t dst_coff, void* stream) {
  DD_REQUIRE(src_phase && dst && batch > 0 && oh > 0 && ow > 0 && off >= 0, DD_ERR_BAD_ARG, "phase3_scatter: bad argument");
  DD_REQUIRE((oh + 2) / 3 + off <= src_ph && (ow + 2) / 3 + off <= src_pw, DD_ERR_BAD_ARG, "phase3_scatter: the phase images (%d x %d) do not hold %d x %d pixels at offset %d", src_ph, src_pw, oh, ow, off);
  DD_REQUIRE(dst_cstore % 4 == 0 && dst_coff % 4 == 0 && dst_coff >= 0 && dst_coff + 32 <= dst_cstore, DD_ERR_BAD_ARG, "phase3_scatter: channel slice");
  const long total = (long)batch * oh * ow * 8;
  hipLaunchKernelGGL(phase3_scatter_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)src_phase, dst, batch, oh, ow, src_ph, src_pw, off, dst_cstore, dst_coff);
  DD_LAUNCH_CHECK("phase3_scatter");
  return 0;
}

int dd_phase3_gather(const float* src, float* dst_phase, int32_t batch, int32_t oh, int32_t ow, int32_t dst_ph, int32_t dst_pw, int32_t off,
                     int32_t src_cstore, int32_t src_coff, void* stream) {
  DD_REQUIRE(src && dst_phase && batch > 0 && oh > 0 && ow > 0 && off >= 0 && dst_ph > 0 && dst_pw > 0, DD_ERR_BAD_ARG, "phase3_gather: bad argument");
  DD_REQUIRE(src_cstore % 4 == 0 && src_coff % 4 == 0 && src_coff >= 0 && src_coff + 32 <= src_cstore, DD_ERR_BAD_ARG, "phase3_gather: channel slice");
  const long total = (long)batch * 9 * dst_ph * dst_pw * 8;
  hipLaunchKernelGGL(phase3_gather_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0, (hipStream_t)stream,
                     src, (f32x4*)dst_phase, batch, oh, ow, dst_ph, dst_pw, off, src_cstore, src_coff);
  DD_LAUNCH_CHECK("phase3_gather");
  return 0;
}

int dd_conv1ch_fwd(const float* taps4, const float* w, const float* bias, float* y, int32_t batch, int32_t sh, int32_t sw, int32_t relu,
                   void* stream) {
  DD_REQUIRE(taps4 && w && y && batch > 0, DD_ERR_BAD_ARG, "conv1ch_fwd: bad argument");
  DD_REQUIRE(sh >= C1_K && sw >= C1_K && sw <= C1_MAXW, DD_ERR_UNSUPPORTED, "conv1ch_fwd: image %dx%d (width at most %d)", sh, sw, C1_MAXW);
  const int oh = sh - (C1_K - 1);
  hipLaunchKernelGGL(conv1ch_fwd_kernel<false>, dim3((unsigned)(batch * oh)), dim3(256), 0, (hipStream_t)stream, taps4, w, bias, y,
                     (unsigned*)nullptr, sh, sw, relu);
  DD_LAUNCH_CHECK("conv1ch_fwd");
  return 0;
}

int64_t dd_conv1ch_wgrad_workspace_bytes(void) { return (int64_t)DD_NUM_CU * 2 * 2 * 16 * 64 * 4; }

int dd_conv1ch_wgrad(const float* taps4, const float* g, float* dw, float* dbias, int32_t batch, int32_t sh, int32_t sw, void* workspace,
                     void* stream) {
  DD_REQUIRE(taps4 && g && dw && workspace && batch > 0, DD_ERR_BAD_ARG, "conv1ch_wgrad: bad argument");
  DD_REQUIRE(sh >= C1_K && sw >= C1_K && sw <= C1_MAXW, DD_ERR_UNSUPPORTED, "conv1ch_wgrad: image %dx%d (width at most %d)", sh, sw, C1_MAXW);
  const long rows = (long)batch * (sh - (C1_K - 1));
  const int nwg = (int)max(1L, min((long)dd_cu_budget_internal() * 2, rows));
  hipLaunchKernelGGL(conv1ch_wgrad_kernel<false>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, taps4, g, (float*)workspace, batch, sh, sw);
  hipLaunchKernelGGL(conv1ch_wgrad_reduce, dim3(C1_T + 1), dim3(512), 0, (hipStream_t)stream, (const float*)workspace, dw, dbias, nwg);
  DD_LAUNCH_CHECK("conv1ch_wgrad");
  return 0;
}

}  // extern "C"
