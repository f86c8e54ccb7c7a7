// Dense-head pieces of the hot path (all HBM- or latency-bound, no matrix work):
//   dd_bn_relu_drop_*   BatchNorm1d -> ReLU -> dropout(mask)   reference components.py:104-109 (DenseBlock.forward)
//   dd_bce_logits       mean BCE-with-logits fwd + bwd + sigmoid in ONE pass over the 640000-wide maps
//                       (roadmap_bce_v2.py:81,106)
//   dd_mse              mean squared error fwd + bwd (autoencoder.py:91, roadmap_pretrain_ae.py:100)
//   dd_adam_step        torch.optim.Adam over one flat buffer (autoencoder.py:119-120)
#include <stdlib.h>

#include "dd_common.h"
#include "dd_adam.h"

namespace {

// ---- BatchNorm1d + ReLU + dropout.  One thread per feature; consecutive threads read consecutive
// features, so every row access is a coalesced 256-byte wave load.  Batch statistics are two-pass
// (mean, then centred second moment) like torch's CPU kernel, not E[x^2]-E[x]^2.
// R > 0: the batch column of the feature (rows <= R) is loaded ONCE, all loads in flight together, and every pass runs on
// registers (the plain loops below re-read it three times, each pass a chain of dependent-looking loads: 23 us for a
// 32 x 128 block, and three HBM passes over the decoder's 32 x 1.25 M block).  Sums run in the same order either way.
template <int R>
__global__ __launch_bounds__(256) void bn_relu_drop_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ rmean, float* __restrict__ rvar, const float* __restrict__ keep, float* __restrict__ y,
    float* __restrict__ smean, float* __restrict__ sinv, int rows, int feat, float eps, float momentum, float scale,
    int training, long long* __restrict__ nbt) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= feat) return;
  if (nbt && f == 0) *nbt += 1;      // BatchNorm's num_batches_tracked.add_(1): no launch of its own
  constexpr int RR = R > 0 ? R : 1;
  float xv[RR];
  if (R > 0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) xv[r] = (r < rows) ? x[(long)r * feat + f] : 0.f;
  }
  auto xat = [&](int r) { return R > 0 ? xv[r] : x[(long)r * feat + f]; };
  float mean, invstd;
  if (training) {
    float s = 0.f, ss = 0.f;
    if (R > 0) {
#pragma unroll
      for (int r = 0; r < RR; ++r) s += (r < rows) ? xv[r] : 0.f;
      mean = s / rows;
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const float d = xv[r] - mean;
        ss += (r < rows) ? d * d : 0.f;
      }
    } else {
      for (int r = 0; r < rows; ++r) s += xat(r);
      mean = s / rows;
      for (int r = 0; r < rows; ++r) {
        const float d = xat(r) - mean;
        ss += d * d;
      }
    }
    const float var = ss / rows;
    invstd = 1.0f / sqrtf(var + eps);
    smean[f] = mean;
    sinv[f] = invstd;
    const float unbiased = rows > 1 ? ss / (rows - 1) : var;
    rmean[f] = (1.f - momentum) * rmean[f] + momentum * mean;
    rvar[f] = (1.f - momentum) * rvar[f] + momentum * unbiased;
  } else {
    mean = rmean[f];
    invstd = 1.0f / sqrtf(rvar[f] + eps);
  }
  const float g = gamma[f] * invstd, b = beta[f];
  if (R > 0) {
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      if (r < rows) {
        const long i = (long)r * feat + f;
        float v = fmaxf((xv[r] - mean) * g + b, 0.f);
        if (keep) v = v * keep[i] * scale;
        y[i] = v;
      }
    }
  } else {
    for (int r = 0; r < rows; ++r) {
      const long i = (long)r * feat + f;
      float v = fmaxf((x[i] - mean) * g + b, 0.f);
      if (keep) v = v * keep[i] * scale;
      y[i] = v;
    }
  }
}

template <int R>
__global__ __launch_bounds__(256) void bn_relu_drop_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ keep, const float* __restrict__ smean,
    const float* __restrict__ sinv, const float* __restrict__ rmean, const float* __restrict__ rvar,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int feat, float eps,
    float scale, int training) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= feat) return;
  const float mean = training ? smean[f] : rmean[f];
  const float invstd = training ? sinv[f] : 1.0f / sqrtf(rvar[f] + eps);
  const float ks = keep ? scale : 1.f;
  const float g = gamma[f] * invstd;
  const float inv_rows = 1.f / rows;
  float sdz = 0.f, sdzx = 0.f;
  if (R > 0) {
    constexpr int RR = R > 0 ? R : 1;
    float dzv[RR], xh[RR];      // one trip to memory: dz and x-hat of the whole batch column stay in registers
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const long i = (long)r * feat + f;
      const bool ok = r < rows;
      const float yy = ok ? y[i] : 0.f, dd = ok ? dy[i] : 0.f, xx = ok ? x[i] : mean;
      dzv[r] = (yy > 0.f) ? dd * ks : 0.f;   // y > 0  <=>  ReLU open AND unit kept
      xh[r] = (xx - mean) * invstd;
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      sdz += dzv[r];
      sdzx += dzv[r] * xh[r];
    }
    dbeta[f] = sdz;
    dgamma[f] = sdzx;
#pragma unroll
    for (int r = 0; r < RR; ++r)
      if (r < rows) dx[(long)r * feat + f] = training ? g * (dzv[r] - inv_rows * (sdz + xh[r] * sdzx)) : g * dzv[r];
    return;
  }
  for (int r = 0; r < rows; ++r) {
    const long i = (long)r * feat + f;
    const float dz = (y[i] > 0.f) ? dy[i] * ks : 0.f;   // y > 0  <=>  ReLU open AND unit kept
    sdz += dz;
    sdzx += dz * (x[i] - mean) * invstd;
  }
  dbeta[f] = sdz;
  dgamma[f] = sdzx;
  for (int r = 0; r < rows; ++r) {
    const long i = (long)r * feat + f;
    const float dz = (y[i] > 0.f) ? dy[i] * ks : 0.f;
    if (training) {
      const float xhat = (x[i] - mean) * invstd;
      dx[i] = g * (dz - inv_rows * (sdz + xhat * sdzx));
    } else {
      dx[i] = g * dz;
    }
  }
}

// ---- the same two kernels for WIDE layers (the decoder's DenseBlock: 32 rows x 1.25 M features): V consecutive features per thread,
// 16- / 8-byte accesses (a quarter / half of the memory instructions, 1 KB / 512 B per wave instruction), non-temporal: the 160 MB
// activations stream through once.  Per feature the sums run over the rows in the same order as above: the same bits.
template <int V>
struct BnVec;
template <>
struct BnVec<4> { typedef f32x4 T; };
template <>
struct BnVec<2> { typedef float T __attribute__((ext_vector_type(2))); };

__global__ __launch_bounds__(256) void bn_relu_drop_fwd_vec4(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ rmean,
                                                             float* __restrict__ rvar, const float* __restrict__ keep, float* __restrict__ y,
                                                             float* __restrict__ smean, float* __restrict__ sinv, int rows, int feat, float eps,
                                                             float momentum, float scale, int training, long long* __restrict__ nbt) {
  const int f = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (f >= feat) return;
  if (nbt && f == 0) *nbt += 1;
  f32x4 xv[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) xv[r] = (r < rows) ? __builtin_nontemporal_load((const f32x4*)(x + (long)r * feat + f)) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 mean, invstd;
  if (training) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 32; ++r) s += (r < rows) ? xv[r] : f32x4{0.f, 0.f, 0.f, 0.f};
    mean = s / (float)rows;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const f32x4 dd = xv[r] - mean;
      ss += (r < rows) ? dd * dd : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 var = ss / (float)rows;
    const f32x4 unbiased = rows > 1 ? ss / (float)(rows - 1) : var;
    const f32x4 rm = *(const f32x4*)(rmean + f), rv = *(const f32x4*)(rvar + f);
#pragma unroll
    for (int j = 0; j < 4; ++j) invstd[j] = 1.0f / sqrtf(var[j] + eps);
    *(f32x4*)(smean + f) = mean;
    *(f32x4*)(sinv + f) = invstd;
    *(f32x4*)(rmean + f) = (1.f - momentum) * rm + momentum * mean;
    *(f32x4*)(rvar + f) = (1.f - momentum) * rv + momentum * unbiased;
  } else {
    mean = *(const f32x4*)(rmean + f);
    const f32x4 rv = *(const f32x4*)(rvar + f);
#pragma unroll
    for (int j = 0; j < 4; ++j) invstd[j] = 1.0f / sqrtf(rv[j] + eps);
  }
  const f32x4 g = *(const f32x4*)(gamma + f) * invstd, b = *(const f32x4*)(beta + f);
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    if (r < rows) {
      const long i = (long)r * feat + f;
      f32x4 v = (xv[r] - mean) * g + b;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      if (keep) v = v * __builtin_nontemporal_load((const f32x4*)(keep + i)) * scale;
      __builtin_nontemporal_store(v, (f32x4*)(y + i));
    }
  }
}

__global__ __launch_bounds__(256) void bn_relu_drop_bwd_vec2(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ keep,
                                                             const float* __restrict__ smean, const float* __restrict__ sinv,
                                                             const float* __restrict__ rmean, const float* __restrict__ rvar, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int feat, float eps,
                                                             float scale, int training) {
  typedef BnVec<2>::T f2;
  const int f = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (f >= feat) return;
  f2 mean, invstd;
  if (training) {
    mean = *(const f2*)(smean + f);
    invstd = *(const f2*)(sinv + f);
  } else {
    mean = *(const f2*)(rmean + f);
    const f2 rv = *(const f2*)(rvar + f);
    invstd = f2{1.0f / sqrtf(rv.x + eps), 1.0f / sqrtf(rv.y + eps)};
  }
  const float ks = keep ? scale : 1.f;
  const f2 g = *(const f2*)(gamma + f) * invstd;
  const float inv_rows = 1.f / rows;
  f2 dzv[32], xh[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const long i = (long)r * feat + f;
    const bool ok = r < rows;
    const f2 yy = ok ? __builtin_nontemporal_load((const f2*)(y + i)) : f2{0.f, 0.f};
    const f2 dd = ok ? __builtin_nontemporal_load((const f2*)(dy + i)) : f2{0.f, 0.f};
    const f2 xx = ok ? __builtin_nontemporal_load((const f2*)(x + i)) : mean;
    dzv[r] = f2{yy.x > 0.f ? dd.x * ks : 0.f, yy.y > 0.f ? dd.y * ks : 0.f};
    xh[r] = (xx - mean) * invstd;
  }
  f2 sdz = {0.f, 0.f}, sdzx = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    sdz += dzv[r];
    sdzx += dzv[r] * xh[r];
  }
  *(f2*)(dbeta + f) = sdz;
  *(f2*)(dgamma + f) = sdzx;
#pragma unroll
  for (int r = 0; r < 32; ++r)
    if (r < rows) {
      const f2 o = training ? g * (dzv[r] - inv_rows * (sdz + xh[r] * sdzx)) : g * dzv[r];
      __builtin_nontemporal_store(o, (f2*)(dx + (long)r * feat + f));
    }
}

// ---- losses: grid-stride pass with per-thread fp32 partials, wave shuffle + LDS block reduce,
// one fp64 partial per block, then a single-block fixed-order final sum (deterministic).
__device__ __forceinline__ double block_sum(float v) {
  __shared__ double red[4];
  double d = (double)v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = d;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// TT = float (0/1 values) or unsigned char (bool masks exactly as the dataset hands them over: road_image, data_helper.py)
template <typename TT>
__device__ __forceinline__ f32x4 load_target4(const TT* t, long i);
template <>
__device__ __forceinline__ f32x4 load_target4<float>(const float* t, long i) { return ((const f32x4*)t)[i]; }
template <>
__device__ __forceinline__ f32x4 load_target4<unsigned char>(const unsigned char* t, long i) {
  const unsigned w = ((const unsigned*)t)[i];
  return f32x4{(float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24)};
}

// sigmoid(z) and softplus(-|z|) = log1p(exp(-|z|)) from the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32,
// 1 ulp each) instead of libm's expf + log1pf + IEEE divisions: 145 -> ~25 instructions per element, which is what
// made the loss pass compute-bound (87 us for 184 MB).  log1p(e) = log(u) * e / (u - 1) with u = fl(1 + e) undoes the
// rounding of 1 + e (u - 1 is exact); one Newton step on the reciprocal keeps sigmoid within 1 ulp.
__device__ __forceinline__ void sigmoid_softplus(float z, float& sig, float& l1p) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(z) * 1.4426950408889634f);      // in [0, 1]; flushes to 0 below 2^-126
  const float u = 1.f + e, d = u - 1.f;
  float r = __builtin_amdgcn_rcpf(u);
  r = fmaf(r, fmaf(-u, r, 1.f), r);
  sig = z >= 0.f ? r : e * r;
  l1p = d == 0.f ? e : (__builtin_amdgcn_logf(u) * 0.6931471805599453f) * (e * __builtin_amdgcn_rcpf(d));
}

// The road masks as the collate hands them over: a TUPLE of per-sample bool tensors (helper.py:22-23), which the reference
// stacks (and casts) first (roadmap_bce_v2.py:87).  Reading through a table of per-sample pointers skips that copy.
struct MaskPtrs {
  const unsigned char* p[64];      // by value in the kernel arguments
  long per;                        // elements per sample (a multiple of 4)
};
template <typename TT>
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ z, const TT* __restrict__ t,
                                                         float* __restrict__ dz, float* __restrict__ probs,
                                                         double* __restrict__ partial, long n, float gscale) {
  float s = 0.f;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 zv = ((const f32x4*)z)[i], tv = load_target4<TT>(t, i);
    f32x4 g, p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sig, l1p;
      sigmoid_softplus(zv[k], sig, l1p);
      s += fmaxf(zv[k], 0.f) - zv[k] * tv[k] + l1p;
      p[k] = sig;
      g[k] = (sig - tv[k]) * gscale;
    }
    if (dz) ((f32x4*)dz)[i] = g;
    if (probs) ((f32x4*)probs)[i] = p;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {   // tail (n % 4 elements)
    const long i = 4 * n4 + threadIdx.x;
    const float ti = (float)t[i];
    float sig, l1p;
    sigmoid_softplus(z[i], sig, l1p);
    s += fmaxf(z[i], 0.f) - z[i] * ti + l1p;
    if (dz) dz[i] = (sig - ti) * gscale;
    if (probs) probs[i] = sig;
  }
  const double tot = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void bce_logits_ptrs_kernel(const float* __restrict__ z, const MaskPtrs tab, float* __restrict__ dz,
                                                              float* __restrict__ probs, double* __restrict__ partial, long n,
                                                              float gscale) {
  float s = 0.f;
  const long n4 = n / 4;      // n = batch * tab.per, per % 4 == 0: no tail, a float4 never straddles two samples
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long e = 4 * i, b = e / tab.per;
    const unsigned w = *(const unsigned*)(tab.p[b] + (e - b * tab.per));
    const f32x4 zv = ((const f32x4*)z)[i];
    const f32x4 tv = {(float)(w & 0xff), (float)((w >> 8) & 0xff), (float)((w >> 16) & 0xff), (float)(w >> 24)};
    f32x4 g, p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sig, l1p;
      sigmoid_softplus(zv[k], sig, l1p);
      s += fmaxf(zv[k], 0.f) - zv[k] * tv[k] + l1p;
      p[k] = sig;
      g[k] = (sig - tv[k]) * gscale;
    }
    if (dz) ((f32x4*)dz)[i] = g;
    if (probs) ((f32x4*)probs)[i] = p;
  }
  const double tot = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// x *= *s, unless *s == 1 (then nothing is touched).  The losses write their gradient already scaled by 1/n in the forward
// pass; autograd hands the upstream gradient of the scalar loss over as a device scalar, 1.0 for a plain loss.backward() --
// which a host-side multiply cannot know without a sync, and so pays a full read-modify-write pass for.
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(f32x4* __restrict__ x, const float* __restrict__ s, long n4, float* __restrict__ tail,
                                                              int ntail) {
  const float g = *s;
  if (g == 1.f) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) x[i] = x[i] * g;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] *= g;
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const f32x4* __restrict__ z, f32x4* __restrict__ p, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = z[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sg, l1p;
      sigmoid_softplus(v[k], sg, l1p);
      o[k] = sg;
    }
    p[i] = o;
  }
}

// dz = dp * p * (1 - p): backward of p = sigmoid(z) from the saved probabilities (RoadMap.forward, roadmap_pretrain_ae.py:76).
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const f32x4* __restrict__ dp, const f32x4* __restrict__ p, f32x4* __restrict__ dz,
                                                          long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g = dp[i], s = p[i];
    dz[i] = g * s * (1.f - s);
  }
}

// BCE on probabilities with torch's clamp of the logs at -100 (F.binary_cross_entropy, spatial_w_rm.py:131).
__global__ __launch_bounds__(256) void bce_probs_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                        float* __restrict__ dp, double* __restrict__ partial, long n,
                                                        float gscale) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pv = p[i], tv = t[i];
    const float lp = fmaxf(logf(pv), -100.f), lq = fmaxf(logf(1.f - pv), -100.f);
    s -= tv * lp + (1.f - tv) * lq;
    if (dp) {   // d/dp of the clamped logs: zero where the clamp is active (p < e^-100 never happens in fp32 except p = 0)
      const float gp = (pv > 0.f) ? 1.f / pv : 0.f, gq = (pv < 1.f) ? 1.f / (1.f - pv) : 0.f;
      dp[i] = (-(tv * gp) + (1.f - tv) * gq) * gscale;
    }
  }
  const double tot = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ da, double* __restrict__ partial, long n,
                                                  float gscale) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    s += d * d;
    if (da) da[i] = 2.f * d * gscale;
  }
  const double tot = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void loss_final_kernel(const double* __restrict__ partial, int nblocks, double inv_n,
                                                         float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}

// Threat score tp / (sum a + sum b - tp) (reference src/utils/helper.py:74-77), optionally on round(b)
// (roadmap_bce_v2.py:140): one pass, three sums.
__global__ __launch_bounds__(256) void threat_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             double* __restrict__ partial, long n, int round_b) {
  float tp = 0.f, sa = 0.f, sb = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float av = a[i];
    const float bv = round_b ? rintf(b[i]) : b[i];      // torch.round = round half to even = rintf
    tp += av * bv;
    sa += av;
    sb += bv;
  }
  const double t0 = block_sum(tp);
  __syncthreads();
  const double t1 = block_sum(sa);
  __syncthreads();
  const double t2 = block_sum(sb);
  if (threadIdx.x == 0) {
    partial[3L * blockIdx.x + 0] = t0;
    partial[3L * blockIdx.x + 1] = t1;
    partial[3L * blockIdx.x + 2] = t2;
  }
}

__global__ __launch_bounds__(64) void threat_final_kernel(const double* __restrict__ partial, int nblocks,
                                                          float* __restrict__ out) {
  double s = 0.0;
  const int k = threadIdx.x;
  if (k < 3)
    for (int b = 0; b < nblocks; ++b) s += partial[3L * b + k];
  const double tp = __shfl(s, 0), sa = __shfl(s, 1), sb = __shfl(s, 2);
  if (k == 0) out[0] = (float)(tp / (sa + sb - tp));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale) {
  const long n4 = n / 4;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    // streaming accesses: this pass runs beside the conv backward and should not push its 4.5 GB through the caches
    f32x4 pv = __builtin_nontemporal_load((f32x4*)p + i), mv = __builtin_nontemporal_load((f32x4*)m + i),
          vv = __builtin_nontemporal_load((f32x4*)v + i);
    const f32x4 gv = __builtin_nontemporal_load((const f32x4*)g + i);
    const float inv_bc2 = 1.f / bc2_sqrt;
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
      f32x2a pe = {pv[k], pv[k + 1]}, me = {mv[k], mv[k + 1]}, ve = {vv[k], vv[k + 1]};
      adam_elem2(pe, me, ve, f32x2a{gv[k], gv[k + 1]}, gscale, b1, b2, eps, step_size, inv_bc2);
      pv[k] = pe.x; pv[k + 1] = pe.y; mv[k] = me.x; mv[k + 1] = me.y; vv[k] = ve.x; vv[k + 1] = ve.y;
    }
    __builtin_nontemporal_store(pv, (f32x4*)p + i);
    __builtin_nontemporal_store(mv, (f32x4*)m + i);
    __builtin_nontemporal_store(vv, (f32x4*)v + i);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
    const long i = 4 * n4 + threadIdx.x;
    float pe = p[i], me = m[i], ve = v[i];
    adam_elem(pe, me, ve, g[i], gscale, b1, b2, eps, step_size, bc2_sqrt);
    p[i] = pe; m[i] = me; v[i] = ve;
  }
}

// The small tensors of a model (biases, BatchNorm vectors, 3x3 filters) in ONE launch: sixteen 6-us launches at the very
// end of the step are 0.1 ms nothing overlaps.  Block b works on tensor t with first[t] <= b < first[t+1].
constexpr int ADAM_MULTI_MAX = 48;
struct AdamTable {
  float* p[ADAM_MULTI_MAX];
  const float* g[ADAM_MULTI_MAX];
  float* m[ADAM_MULTI_MAX];
  float* v[ADAM_MULTI_MAX];
  int n[ADAM_MULTI_MAX];
  int first[ADAM_MULTI_MAX + 1];
  int count;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamTable tab, float lr, float b1, float b2, float eps, float bc1,
                                                         float bc2_sqrt, float gscale) {
  int t = 0;
  while (t + 1 < tab.count && (int)blockIdx.x >= tab.first[t + 1]) ++t;
  float* __restrict__ p = tab.p[t];
  const float* __restrict__ g = tab.g[t];
  float* __restrict__ m = tab.m[t];
  float* __restrict__ v = tab.v[t];
  const int i = ((int)blockIdx.x - tab.first[t]) * 256 + threadIdx.x;
  if (i >= tab.n[t]) return;
  const float step_size = lr / bc1;
  float pe = p[i], me = m[i], ve = v[i];
  adam_elem(pe, me, ve, g[i], gscale, b1, b2, eps, step_size, bc2_sqrt);
  p[i] = pe; m[i] = me; v[i] = ve;
}

constexpr int kLossBlocks = DD_NUM_CU * 8;

}  // namespace

extern "C" {

int dd_bn_relu_drop_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        const float* keep, float* y, float* save_mean, float* save_invstd, int32_t rows, int32_t feat,
                        float eps, float momentum, float scale, int32_t training, int64_t* num_batches_tracked, void* stream) {
  DD_REQUIRE(x && gamma && beta && running_mean && running_var && y, DD_ERR_BAD_ARG, "bn_fwd: NULL pointer");
  DD_REQUIRE(rows > 0 && feat > 0, DD_ERR_BAD_ARG, "bn_fwd: non-positive size");
  DD_REQUIRE(!training || (save_mean && save_invstd), DD_ERR_BAD_ARG, "bn_fwd: training mode needs save buffers");
  DD_REQUIRE(!training || rows > 1, DD_ERR_UNSUPPORTED, "bn_fwd: batch statistics need more than 1 row (torch raises too)");
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if (rows <= 32 && feat >= (1 << 16) && feat % 4 == 0 && al16(x) && al16(y) && al16(gamma) && al16(beta) && al16(running_mean) &&
      al16(running_var) && (!keep || al16(keep)) && (!training || (al16(save_mean) && al16(save_invstd)))) {
    hipLaunchKernelGGL(bn_relu_drop_fwd_vec4, dim3((feat / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, running_mean,
                       running_var, keep, y, save_mean, save_invstd, rows, feat, eps, momentum, scale, training,
                       (long long*)(training ? num_batches_tracked : nullptr));
    DD_LAUNCH_CHECK("bn_relu_drop_fwd");
    return 0;
  }
  auto k = rows <= 32 ? bn_relu_drop_fwd_kernel<32> : rows <= 64 ? bn_relu_drop_fwd_kernel<64> : bn_relu_drop_fwd_kernel<0>;
  hipLaunchKernelGGL(k, dim3((feat + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, running_mean, running_var, keep, y,
                     save_mean, save_invstd, rows, feat, eps, momentum, scale, training, (long long*)(training ? num_batches_tracked : nullptr));
  DD_LAUNCH_CHECK("bn_relu_drop_fwd");
  return 0;
}

int dd_bn_relu_drop_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* keep,
                        const float* save_mean, const float* save_invstd, const float* running_mean,
                        const float* running_var, float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t feat, float eps, float scale,
                        int32_t training, void* stream) {
  DD_REQUIRE(dy && x && y && gamma && dx && dgamma && dbeta, DD_ERR_BAD_ARG, "bn_bwd: NULL pointer");
  DD_REQUIRE(training ? (save_mean && save_invstd) : (running_mean && running_var), DD_ERR_BAD_ARG, "bn_bwd: missing statistics");
  DD_REQUIRE(rows > 0 && feat > 0, DD_ERR_BAD_ARG, "bn_bwd: non-positive size");
  auto al8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
  if (rows <= 32 && feat >= (1 << 16) && feat % 2 == 0 && al8(dy) && al8(x) && al8(y) && al8(gamma) && al8(dx) && al8(dgamma) && al8(dbeta) &&
      (training ? (al8(save_mean) && al8(save_invstd)) : (al8(running_mean) && al8(running_var)))) {
    hipLaunchKernelGGL(bn_relu_drop_bwd_vec2, dim3((feat / 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, x, y, gamma, keep, save_mean,
                       save_invstd, running_mean, running_var, dx, dgamma, dbeta, rows, feat, eps, scale, training);
    DD_LAUNCH_CHECK("bn_relu_drop_bwd");
    return 0;
  }
  auto k = rows <= 32 ? bn_relu_drop_bwd_kernel<32> : rows <= 64 ? bn_relu_drop_bwd_kernel<64> : bn_relu_drop_bwd_kernel<0>;
  hipLaunchKernelGGL(k, dim3((feat + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, x, y, gamma, keep, save_mean, save_invstd,
                     running_mean, running_var, dx, dgamma, dbeta, rows, feat, eps, scale, training);
  DD_LAUNCH_CHECK("bn_relu_drop_bwd");
  return 0;
}

int64_t dd_loss_workspace_bytes(int64_t n) {
  (void)n;
  return (int64_t)kLossBlocks * sizeof(double);
}

int dd_bce_logits(const float* logits, const float* target, float* loss_out, float* dlogits, float* probs, int64_t n,
                  float grad_scale, void* workspace, void* stream) {
  DD_REQUIRE(logits && target && loss_out && workspace && n > 0, DD_ERR_BAD_ARG, "bce_logits: bad argument");
  DD_REQUIRE(((uintptr_t)logits | (uintptr_t)target | (uintptr_t)dlogits | (uintptr_t)probs) % 16 == 0, DD_ERR_BAD_ARG,
             "bce_logits: buffers must be 16-byte aligned");
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long)kLossBlocks);
  hipLaunchKernelGGL(bce_logits_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, dlogits, probs,
                     (double*)workspace, (long)n, grad_scale / (float)n);
  DD_LAUNCH_CHECK("bce_logits");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, grid,
                     1.0 / (double)n, loss_out);
  DD_LAUNCH_CHECK("loss_final");
  return 0;
}

int dd_bce_logits_u8(const float* logits, const unsigned char* target, float* loss_out, float* dlogits, float* probs, int64_t n,
                     float grad_scale, void* workspace, void* stream) {
  DD_REQUIRE(logits && target && loss_out && workspace && n > 0, DD_ERR_BAD_ARG, "bce_logits_u8: bad argument");
  DD_REQUIRE(((uintptr_t)logits | (uintptr_t)dlogits | (uintptr_t)probs) % 16 == 0 && (uintptr_t)target % 4 == 0, DD_ERR_BAD_ARG,
             "bce_logits_u8: misaligned buffer");
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long)kLossBlocks);
  hipLaunchKernelGGL(bce_logits_kernel<unsigned char>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, dlogits, probs,
                     (double*)workspace, (long)n, grad_scale / (float)n);
  DD_LAUNCH_CHECK("bce_logits_u8");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, grid,
                     1.0 / (double)n, loss_out);
  DD_LAUNCH_CHECK("loss_final");
  return 0;
}

int dd_bce_logits_u8_ptrs(const float* logits, const unsigned char* const* target_ptrs, int32_t batch, int64_t per_sample,
                          float* loss_out, float* dlogits, float* probs, float grad_scale, void* workspace, void* stream) {
  DD_REQUIRE(logits && target_ptrs && loss_out && workspace && batch > 0 && per_sample > 0, DD_ERR_BAD_ARG, "bce_logits_u8_ptrs: bad argument");
  DD_REQUIRE(batch <= 64 && per_sample % 4 == 0, DD_ERR_UNSUPPORTED, "bce_logits_u8_ptrs: up to 64 samples of a multiple of 4 elements (got %d x %ld)",
             batch, (long)per_sample);
  DD_REQUIRE(((uintptr_t)logits | (uintptr_t)dlogits | (uintptr_t)probs) % 16 == 0, DD_ERR_BAD_ARG, "bce_logits_u8_ptrs: misaligned buffer");
  MaskPtrs tab;
  tab.per = per_sample;
  for (int i = 0; i < 64; ++i) tab.p[i] = i < batch ? target_ptrs[i] : nullptr;
  for (int i = 0; i < batch; ++i)
    DD_REQUIRE(tab.p[i] && (uintptr_t)tab.p[i] % 4 == 0, DD_ERR_BAD_ARG, "bce_logits_u8_ptrs: sample %d: NULL or misaligned mask", i);
  const int64_t n = (int64_t)batch * per_sample;
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long)kLossBlocks);
  hipLaunchKernelGGL(bce_logits_ptrs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, tab, dlogits, probs, (double*)workspace,
                     (long)n, grad_scale / (float)n);
  DD_LAUNCH_CHECK("bce_logits_u8_ptrs");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, grid, 1.0 / (double)n, loss_out);
  DD_LAUNCH_CHECK("loss_final");
  return 0;
}

int dd_scale_by_device_scalar(float* x, const float* scalar, int64_t n, void* stream) {
  DD_REQUIRE(x && scalar && n > 0, DD_ERR_BAD_ARG, "scale_by_device_scalar: bad argument");
  DD_REQUIRE((uintptr_t)x % 16 == 0, DD_ERR_BAD_ARG, "scale_by_device_scalar: buffer must be 16-byte aligned");
  const long n4 = n / 4;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)max(1L, min((n4 + 255) / 256, (long)kLossBlocks))), dim3(256), 0, (hipStream_t)stream,
                     (f32x4*)x, scalar, n4, x + 4 * n4, (int)(n - 4 * n4));
  DD_LAUNCH_CHECK("scale_by_device_scalar");
  return 0;
}

int dd_sigmoid(const float* z, float* p, int64_t n, void* stream) {
  DD_REQUIRE(z && p && n > 0 && n % 4 == 0, DD_ERR_BAD_ARG, "sigmoid: bad argument (n must be a multiple of 4)");
  const long n4 = n / 4;
  hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)min((n4 + 255) / 256, (long)kLossBlocks)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)z, (f32x4*)p, n4);
  DD_LAUNCH_CHECK("sigmoid");
  return 0;
}

int dd_sigmoid_bwd(const float* dprobs, const float* probs, float* dlogits, int64_t n, void* stream) {
  DD_REQUIRE(dprobs && probs && dlogits && n > 0 && n % 4 == 0, DD_ERR_BAD_ARG, "sigmoid_bwd: bad argument (n must be a multiple of 4)");
  DD_REQUIRE(((uintptr_t)dprobs | (uintptr_t)probs | (uintptr_t)dlogits) % 16 == 0, DD_ERR_BAD_ARG, "sigmoid_bwd: buffers must be 16-byte aligned");
  const long n4 = n / 4;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)min((n4 + 255) / 256, (long)kLossBlocks)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)dprobs, (const f32x4*)probs, (f32x4*)dlogits, n4);
  DD_LAUNCH_CHECK("sigmoid_bwd");
  return 0;
}

int dd_bce_probs(const float* probs, const float* target, float* loss_out, float* dprobs, int64_t n, float grad_scale,
                 void* workspace, void* stream) {
  DD_REQUIRE(probs && target && loss_out && workspace && n > 0, DD_ERR_BAD_ARG, "bce_probs: bad argument");
  const int grid = (int)min((n + 255) / 256, (long)kLossBlocks);
  hipLaunchKernelGGL(bce_probs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, probs, target, dprobs,
                     (double*)workspace, (long)n, grad_scale / (float)n);
  DD_LAUNCH_CHECK("bce_probs");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, grid,
                     1.0 / (double)n, loss_out);
  DD_LAUNCH_CHECK("loss_final");
  return 0;
}

int dd_mse(const float* a, const float* b, float* loss_out, float* da, int64_t n, float grad_scale, void* workspace,
           void* stream) {
  DD_REQUIRE(a && b && loss_out && workspace && n > 0, DD_ERR_BAD_ARG, "mse: bad argument");
  const int grid = (int)min((n + 255) / 256, (long)kLossBlocks);
  hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, da, (double*)workspace, (long)n,
                     grad_scale / (float)n);
  DD_LAUNCH_CHECK("mse");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, grid,
                     1.0 / (double)n, loss_out);
  DD_LAUNCH_CHECK("loss_final");
  return 0;
}

int64_t dd_threat_score_workspace_bytes(void) { return (int64_t)kLossBlocks * 3 * sizeof(double); }

int dd_threat_score(const float* a, const float* b, float* out, int64_t n, int32_t round_b, void* workspace, void* stream) {
  DD_REQUIRE(a && b && out && workspace && n > 0, DD_ERR_BAD_ARG, "threat_score: bad argument");
  const int grid = (int)min((n + 255) / 256, (long)kLossBlocks);
  hipLaunchKernelGGL(threat_partial_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, (double*)workspace, (long)n,
                     round_b);
  DD_LAUNCH_CHECK("threat_score");
  hipLaunchKernelGGL(threat_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, grid, out);
  DD_LAUNCH_CHECK("threat_score final");
  return 0;
}

int dd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 int32_t step, float grad_scale, void* stream) {
  DD_REQUIRE(p && g && m && v && n > 0 && step >= 1, DD_ERR_BAD_ARG, "adam: bad argument");
  DD_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, DD_ERR_BAD_ARG, "adam: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  // ONE persistent block per CU.  This pass usually runs BESIDE the conv backward (optim.HipAdam.overlap_with_backward), whose
  // one-wave-per-SIMD kernels take 440-464 of a SIMD's 512 registers: one Adam wave (48) fits beside them, and -- the point --
  // nothing of this launch is ever left QUEUED: blocks that wait for a CU are dispatched ahead of the next conv kernel when the
  // current one ends, and a kernel of >= 456 registers then waits until they have all drained (tools/ubench/residency.hip; in the
  // step: the c2 data gradient 2.2 ms from dispatch to end instead of 1.6).  Alone the pass is also fastest this way (0.536 ms for
  // fc1's 481 MB = 6.3 TB/s; 4 blocks per CU: 0.586, 8: 0.595).
  const int per_cu = dd_adam_blocks_internal();
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long)DD_NUM_CU * per_cu);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2,
                     eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  DD_LAUNCH_CHECK("adam");
  return 0;
}

int dd_adam_step_multi(const dd_adam_tensor* tensors, int32_t count, float lr, float beta1, float beta2, float eps, int32_t step,
                       float grad_scale, void* stream) {
  DD_REQUIRE(tensors && count > 0 && step >= 1, DD_ERR_BAD_ARG, "adam_multi: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  for (int32_t base = 0; base < count; base += ADAM_MULTI_MAX) {
    AdamTable tab;
    tab.count = min(ADAM_MULTI_MAX, count - base);
    int blocks = 0;
    for (int i = 0; i < tab.count; ++i) {
      const dd_adam_tensor& t = tensors[base + i];
      DD_REQUIRE(t.p && t.g && t.m && t.v && t.n > 0 && t.n < (1 << 30), DD_ERR_BAD_ARG, "adam_multi: tensor %d: NULL pointer or bad size", base + i);
      tab.p[i] = t.p;
      tab.g[i] = t.g;
      tab.m[i] = t.m;
      tab.v[i] = t.v;
      tab.n[i] = (int)t.n;
      tab.first[i] = blocks;
      blocks += (int)((t.n + 255) / 256);
    }
    tab.first[tab.count] = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), grad_scale);
    DD_LAUNCH_CHECK("adam_multi");
  }
  return 0;
}

}  // extern "C"
