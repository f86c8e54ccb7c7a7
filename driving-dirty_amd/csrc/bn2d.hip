// BatchNorm2d pieces of the Conv -> BN2d -> ReLU encoder variant (reference src/autoencoder/components_v2.py:19-24,
// 43-46; its `bn3 = nn.Conv2d(32)` is read as the evident `BatchNorm2d(32)`).
//
// The conv kernels (conv3x3.hip) write the PRE-normalisation tensor u and gather per-wave-lane sum / sum-of-squares
// in their epilogue (lane = channel there, so no cross-lane traffic).  Here:
//   bn2d_finalize     partials -> batch mean / inv-std (fp64), running statistics, and the (scale, shift) table the
//                     consumers apply on the fly:  y = relu(u*scale + shift)
//   bn2d_apply_relu   materialise y (only for the c3_only exit, where the caller wants the tensor)
//   bn2d_bwd_reduce   dbeta = sum g, dgamma = sum g*xhat per channel: registers -> wavefront shuffles -> LDS -> partials
//   bn2d_bwd_apply    du = scale * (g - dbeta/N - xhat*dgamma/N)
//   pool4_*_aff       the NCHW-order max_pool1d(4) (components_v2.py:49-50) reading u through (scale, shift)
#include "dd_common.h"

namespace {

constexpr int kStatRows = DD_NUM_CU * 2 * 8;   // wave slots of dd_conv_fwd_stats (dd_conv_stats_floats / 128)

// table layout handed to the conv kernels: [0:32) input scale, [32:64) input shift, [64:96) mask scale, [96:128) mask shift
__global__ __launch_bounds__(256) void bn2d_finalize_kernel(const float* __restrict__ stats, double count,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ rmean, float* __restrict__ rvar,
                                                            float momentum, float eps, int training,
                                                            float* __restrict__ affine, float* __restrict__ save_mean,
                                                            float* __restrict__ save_invstd) {
  __shared__ double red[8][32][2];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  double s = 0.0, q = 0.0;
  if (training) {
    for (int w = g; w < kStatRows; w += 8) {
      const float* row = stats + (long)w * 128;
      s += (double)row[2 * c] + (double)row[2 * (c + 32)];          // lanes c and c+32 hold channel c
      q += (double)row[2 * c + 1] + (double)row[2 * (c + 32) + 1];
    }
  }
  red[g][c][0] = s;
  red[g][c][1] = q;
  __syncthreads();
  if (g != 0) return;
  float mean, invstd;
  if (training) {
    s = q = 0.0;
    for (int i = 0; i < 8; ++i) {
      s += red[i][c][0];
      q += red[i][c][1];
    }
    const double m = s / count;
    const double var = fmax(q / count - m * m, 0.0);      // fp64: the cancellation costs nothing at these magnitudes
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  } else {
    mean = rmean[c];
    invstd = 1.0f / sqrtf(rvar[c] + eps);
  }
  const float sc = gamma[c] * invstd, sh = beta[c] - mean * sc;
  affine[c] = sc;
  affine[32 + c] = sh;
  affine[64 + c] = sc;
  affine[96 + c] = sh;
  save_mean[c] = mean;
  save_invstd[c] = invstd;
}

// Stand-alone batch statistics of a 32-channel NHWC tensor (the v2 DECODER's ConvTranspose2d outputs come from the generic
// conv kernels, which have no statistics epilogue): per-thread sums over a grid-stride loop, lanes with equal channel group
// combined by wavefront shuffles, the four waves through LDS; one row of the finalize kernel's table per block, rows past
// the grid zeroed.  Row layout: [2c] = sum, [2c+1] = sum of squares of channel c; the entries of lanes 32..63 stay zero.
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const f32x4* __restrict__ u, float* __restrict__ stats, long n4) {
  __shared__ float red[4][8][8];
  f32x4 ss = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {   // stride % 8 == 0
    const f32x4 v = u[i];
    ss += v;
    sq += v * v;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      ss[k] += __shfl_xor(ss[k], o);
      sq[k] += __shfl_xor(sq[k], o);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[wave][lane][k] = ss[k];
      red[wave][lane][4 + k] = sq[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    float v = 0.f;
    if (threadIdx.x < 64) {      // entry 2c + j of the row: channel c = 4*grp + k, j = 0 sum / 1 sum of squares
      const int c = threadIdx.x >> 1, j = threadIdx.x & 1, grp = c >> 2, k = (c & 3) + 4 * j;
      v = (red[0][grp][k] + red[1][grp][k]) + (red[2][grp][k] + red[3][grp][k]);
    }
    stats[(long)blockIdx.x * 128 + threadIdx.x] = v;
    for (int r = blockIdx.x + gridDim.x; r < kStatRows; r += gridDim.x) stats[(long)r * 128 + threadIdx.x] = 0.f;
  }
}

__global__ __launch_bounds__(256) void bn2d_apply_relu_kernel(const f32x4* __restrict__ u, const float* __restrict__ affine,
                                                              f32x4* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i & 7);
    const f32x4 sc = *(const f32x4*)(affine + 4 * cg), sh = *(const f32x4*)(affine + 32 + 4 * cg);
    const f32x4 v = u[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
    y[i] = o;
  }
}

// thread = (pixel lane, 4-channel group): tid & 7 = channel group.  Sums stay in registers over the grid-stride loop,
// then lanes with equal channel group combine by wavefront shuffles (xor 8, 16, 32), the 4 waves through LDS.
__global__ __launch_bounds__(256) void bn2d_bwd_reduce_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ u,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              double* __restrict__ partial, long n4) {
  __shared__ float red[4][8][8];
  const int cg = threadIdx.x & 7;
  const f32x4 mu = *(const f32x4*)(mean + 4 * cg), is = *(const f32x4*)(invstd + 4 * cg);
  f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sg = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {   // stride % 8 == 0
    const f32x4 gv = g[i], uv = u[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sb[k] += gv[k];
      sg[k] += gv[k] * (uv[k] - mu[k]) * is[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      sb[k] += __shfl_xor(sb[k], o);
      sg[k] += __shfl_xor(sg[k], o);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[wave][lane][k] = sb[k];
      red[wave][lane][4 + k] = sg[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // 8 channel groups x (4 dbeta + 4 dgamma)
    const int grp = threadIdx.x >> 3, k = threadIdx.x & 7;
    const double v = ((double)red[0][grp][k] + (double)red[1][grp][k]) + ((double)red[2][grp][k] + (double)red[3][grp][k]);
    const int c = 4 * grp + (k & 3);
    partial[(long)blockIdx.x * 64 + (k < 4 ? c : 32 + c)] = v;
  }
}

__global__ __launch_bounds__(64) void bn2d_bwd_final_kernel(const double* __restrict__ partial, int nblocks,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int t = threadIdx.x;   // 0..31 dbeta, 32..63 dgamma
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(long)b * 64 + t];
  if (t < 32) dbeta[t] = (float)s; else dgamma[t - 32] = (float)s;
}

__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ u,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ dgamma,
                                                             const float* __restrict__ dbeta, float inv_count, int training,
                                                             f32x4* __restrict__ du, long n4) {
  const int cg = threadIdx.x & 7;
  f32x4 sc, mu, is, kb, kg;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * cg + k;
    is[k] = invstd[c];
    mu[k] = mean[c];
    sc[k] = gamma[c] * is[k];
    kb[k] = training ? dbeta[c] * inv_count : 0.f;
    kg[k] = training ? dgamma[c] * inv_count : 0.f;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 gv = g[i], uv = u[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = sc[k] * (gv[k] - kb[k] - (uv[k] - mu[k]) * is[k] * kg[k]);
    du[i] = o;
  }
}

// ---- pool (fast path only: H*W % 4 == 0, C == 32): the same quad kernels as layout_pool.hip, reading relu(u*scale+shift)
__global__ __launch_bounds__(256) void pool4_fwd_quad_aff(const f32x4* __restrict__ feat, const float* __restrict__ affine,
                                                          float* __restrict__ pooled, int B, long HW) {
  const long quads = HW / 4, total = (long)B * quads * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i & 7);
    const long q = (i >> 3) % quads, b = i / (8 * quads);
    const f32x4 sc = *(const f32x4*)(affine + 4 * g), sh = *(const f32x4*)(affine + 32 + 4 * g);
    const f32x4* p = feat + ((b * HW + 4 * q) * 8 + g);
    const f32x4 v0 = p[0], v1 = p[8], v2 = p[16], v3 = p[24];
    float* o = pooled + b * (quads * 32) + q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a0 = fmaxf(v0[k] * sc[k] + sh[k], 0.f), a1 = fmaxf(v1[k] * sc[k] + sh[k], 0.f);
      const float a2 = fmaxf(v2[k] * sc[k] + sh[k], 0.f), a3 = fmaxf(v3[k] * sc[k] + sh[k], 0.f);
      o[(long)(4 * g + k) * quads] = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    }
  }
}

__global__ __launch_bounds__(256) void pool4_bwd_quad_aff(const float* __restrict__ dpooled, const f32x4* __restrict__ feat,
                                                          const float* __restrict__ affine, f32x4* __restrict__ dfeat, int B,
                                                          long HW) {
  const long quads = HW / 4, total = (long)B * quads * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i & 7);
    const long q = (i >> 3) % quads, b = i / (8 * quads);
    const f32x4 sc = *(const f32x4*)(affine + 4 * g), sh = *(const f32x4*)(affine + 32 + 4 * g);
    const long base = (b * HW + 4 * q) * 8 + g;
    const f32x4 v0 = feat[base], v1 = feat[base + 8], v2 = feat[base + 16], v3 = feat[base + 24];
    const float* gp = dpooled + b * (quads * 32) + q;
    f32x4 d0, d1, d2, d3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a[4] = {fmaxf(v0[k] * sc[k] + sh[k], 0.f), fmaxf(v1[k] * sc[k] + sh[k], 0.f),
                          fmaxf(v2[k] * sc[k] + sh[k], 0.f), fmaxf(v3[k] * sc[k] + sh[k], 0.f)};
      float m = a[0];
      int am = 0;
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (a[j] > m) { m = a[j]; am = j; }
      const float gv = (m > 0.f) ? gp[(long)(4 * g + k) * quads] : 0.f;   // gradient w.r.t. the BN output (ReLU applied)
      d0[k] = am == 0 ? gv : 0.f;
      d1[k] = am == 1 ? gv : 0.f;
      d2[k] = am == 2 ? gv : 0.f;
      d3[k] = am == 3 ? gv : 0.f;
    }
    dfeat[base] = d0;
    dfeat[base + 8] = d1;
    dfeat[base + 16] = d2;
    dfeat[base + 24] = d3;
  }
}

int grid_for(long n) { return (int)min((n + 255) / 256, (long)DD_NUM_CU * 8); }

}  // namespace

extern "C" {

int dd_bn2d_finalize(const float* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int32_t training, float* affine, float* save_mean,
                     float* save_invstd, void* stream) {
  DD_REQUIRE(gamma && beta && running_mean && running_var && affine && save_mean && save_invstd, DD_ERR_BAD_ARG, "bn2d_finalize: NULL pointer");
  DD_REQUIRE(!training || (stats && count > 1), DD_ERR_BAD_ARG, "bn2d_finalize: training mode needs statistics of more than one value");
  hipLaunchKernelGGL(bn2d_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, training, affine, save_mean, save_invstd);
  DD_LAUNCH_CHECK("bn2d_finalize");
  return 0;
}

int dd_bn2d_stats(const float* u, float* stats, int64_t npix, void* stream) {
  DD_REQUIRE(u && stats && npix > 0, DD_ERR_BAD_ARG, "bn2d_stats: bad argument");
  DD_REQUIRE((uintptr_t)u % 16 == 0, DD_ERR_BAD_ARG, "bn2d_stats: buffer must be 16-byte aligned");
  const long n4 = npix * 8;
  const int grid = (int)min((long)grid_for(n4), (long)kStatRows);
  hipLaunchKernelGGL(bn2d_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)u, stats, n4);
  DD_LAUNCH_CHECK("bn2d_stats");
  return 0;
}

int dd_bn2d_apply_relu(const float* u, const float* affine, float* y, int64_t npix, void* stream) {
  DD_REQUIRE(u && affine && y && npix > 0, DD_ERR_BAD_ARG, "bn2d_apply_relu: bad argument");
  const long n4 = npix * 8;
  hipLaunchKernelGGL(bn2d_apply_relu_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)u, affine,
                     (f32x4*)y, n4);
  DD_LAUNCH_CHECK("bn2d_apply_relu");
  return 0;
}

int64_t dd_bn2d_workspace_bytes(void) { return (int64_t)DD_NUM_CU * 8 * 64 * sizeof(double); }

int dd_bn2d_bwd(const float* g, const float* u, const float* gamma, const float* save_mean, const float* save_invstd,
                float* du, float* dgamma, float* dbeta, int64_t npix, int32_t training, void* workspace, void* stream) {
  DD_REQUIRE(g && u && gamma && save_mean && save_invstd && du && dgamma && dbeta && workspace && npix > 0, DD_ERR_BAD_ARG,
             "bn2d_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long n4 = npix * 8;
  const int grid = grid_for(n4);
  hipLaunchKernelGGL(bn2d_bwd_reduce_kernel, dim3(grid), dim3(256), 0, st, (const f32x4*)g, (const f32x4*)u, save_mean,
                     save_invstd, (double*)workspace, n4);
  DD_LAUNCH_CHECK("bn2d_bwd_reduce");
  hipLaunchKernelGGL(bn2d_bwd_final_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, grid, dgamma, dbeta);
  DD_LAUNCH_CHECK("bn2d_bwd_final");
  hipLaunchKernelGGL(bn2d_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, (const f32x4*)g, (const f32x4*)u, gamma, save_mean,
                     save_invstd, dgamma, dbeta, 1.0f / (float)npix, training, (f32x4*)du, n4);
  DD_LAUNCH_CHECK("bn2d_bwd_apply");
  return 0;
}

int dd_pool4_bn_fwd(const float* u, const float* affine, float* pooled, int32_t batch, int32_t h, int32_t w, void* stream) {
  DD_REQUIRE(u && affine && pooled && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "pool4_bn_fwd: bad argument");
  DD_REQUIRE(((long)h * w) % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_bn: H*W must be a multiple of 4");
  const long HW = (long)h * w, total = (long)batch * (HW / 4) * 8;
  hipLaunchKernelGGL(pool4_fwd_quad_aff, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)u, affine, pooled,
                     batch, HW);
  DD_LAUNCH_CHECK("pool4_bn_fwd");
  return 0;
}

int dd_pool4_bn_bwd(const float* dpooled, const float* u, const float* affine, float* dfeat, int32_t batch, int32_t h, int32_t w,
                    void* stream) {
  DD_REQUIRE(dpooled && u && affine && dfeat && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "pool4_bn_bwd: bad argument");
  DD_REQUIRE(((long)h * w) % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_bn: H*W must be a multiple of 4");
  const long HW = (long)h * w, total = (long)batch * (HW / 4) * 8;
  hipLaunchKernelGGL(pool4_bwd_quad_aff, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dpooled, (const f32x4*)u, affine,
                     (f32x4*)dfeat, batch, HW);
  DD_LAUNCH_CHECK("pool4_bn_bwd");
  return 0;
}

}  // extern "C"
