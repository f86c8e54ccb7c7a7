// The encoder's small tail in ONE launch each way (reference components.py:48-51, 104-109):
//
//     lin1 [M,H1] (the big fc1 GEMM's output)
//       -> BatchNorm1d -> ReLU -> dropout            (DenseBlock fc1, second half)
//       -> Linear(H1 -> H2) -> BatchNorm1d -> ReLU -> dropout   (DenseBlock fc2)
//       -> Linear(H2 -> L)                            (fc_z_out)  -> z [M,L]
//
// At the roadmap model's sizes (M = 32, H1 = H2 = 128, L = 64) this is ~1 MFLOP, but as separate kernels it is 13
// launches forward and 8 backward, every one of them bounded below by the ~5.6 us a dependent launch costs: 90 + 144 us
// of an 8 ms step.  Here one 1024-thread workgroup keeps every operand in LDS and walks the chain: plain fp32 FMAs on 4 x 4 register tiles, both GEMM operands
// contraction-major in LDS so that 16 FMAs cost two ds_read_b128 (the matrix cores would not notice this much work).  Sums over the batch run in row order, as the stand-alone
// dd_bn_relu_drop kernels do.  Limits: M <= 32, H1, H2, L <= 128 (the caller falls back to the separate kernels).
#include <hip/hip_runtime.h>

#include "dd_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TM = 32, TH = 128;
constexpr int TAIL_THREADS = 1024;      // one workgroup: the more waves, the more loads in flight per memory round trip
constexpr int LDF = TH + 4;      // pitch of a row-major tile  [row m][feature]   (float4 along the features)
constexpr int LDM = TM + 4;      // pitch of a feature-major tile [feature][row m] (float4 along the rows)

struct TailFwd {
  const float *lin1, *g1, *b1, *keep1, *w2, *bias2, *g2, *b2, *keep2, *wz, *bz;
  float *rm1, *rv1, *rm2, *rv2;
  long long *nbt1, *nbt2;
  float *y1, *lin2, *y2, *z, *mean1, *inv1, *mean2, *inv2;
  int M, H1, H2, L, training;
  float eps1, eps2, mom1, mom2, scale1, scale2;
};

struct TailBwd {
  const float *dz, *lin1, *y1, *lin2, *y2, *g1, *g2, *keep1, *keep2, *w2, *wz;
  const float *mean1, *inv1, *mean2, *inv2, *rm1, *rv1, *rm2, *rv2;
  float *dlin1, *dg1, *db1, *dw2, *dbias2, *dg2, *db2, *dwz, *dbz;
  int M, H1, H2, L, training;
  float eps1, eps2, scale1, scale2;
};

// global [rows][cols] (cols % 4 == 0) -> LDS, row-major [r][ld] (TR = false) or transposed [c][ld] (TR = true); rows in
// [rows, pad_rows) are zeroed.  Eight float4 loads are in flight per thread before the first store: one workgroup has
// nobody else to hide its memory latency behind (a load-store-load-store loop took 60 us for 64 KB).
template <bool TR>
__device__ __forceinline__ void load_tile(float* dst, int ld, const float* src, int rows, int cols, int pad_rows) {
  const int total = pad_rows * cols, valid = rows * cols;
  for (int base = threadIdx.x * 4; base < total; base += TAIL_THREADS * 4 * 8) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * TAIL_THREADS * 4;
      v[u] = (i < valid) ? *(const f4*)(src + i) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * TAIL_THREADS * 4;
      if (i < total) {
        const int r = i / cols, c = i % cols;
        if (TR) {
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[(c + q) * ld + r] = v[u][q];
        } else {
          *(f4*)(dst + r * ld + c) = v[u];
        }
      }
    }
  }
}
__device__ __forceinline__ void load_rows(float* dst, int ld, const float* src, int rows, int cols, int pad_rows) {
  load_tile<false>(dst, ld, src, rows, cols, pad_rows);
}
__device__ __forceinline__ void load_cols(float* dst, int ld, const float* src, int rows, int cols, int pad_rows) {
  load_tile<true>(dst, ld, src, rows, cols, pad_rows);
}

// C[i][j] (+)= sum_k A[k][i] * B[k][j]: both operands k-major in LDS (pitches lda, ldb), 4 x 4 register tiles, two float4
// LDS reads per 16 FMAs.  I, J multiples of 4.  `emit(i0, j0, acc)` receives each finished tile (acc[ii][jj]).
template <typename Emit>
__device__ __forceinline__ void gemm_kmajor(const float* A, int lda, const float* B, int ldb, int I, int J, int K, Emit emit) {
  const int tj = J / 4, tiles = (I / 4) * tj;
  for (int t = threadIdx.x; t < tiles; t += TAIL_THREADS) {
    const int i0 = 4 * (t / tj), j0 = 4 * (t % tj);
    f4 acc[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) acc[ii] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < K; ++k) {      // unrolled: eight pairs of LDS reads in flight (one workgroup = nothing else hides LDS latency)
      const f4 av = *(const f4*)(A + k * lda + i0), bv = *(const f4*)(B + k * ldb + j0);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) acc[ii] += av[ii] * bv;
    }
    emit(i0, j0, acc);
  }
}

// BatchNorm (batch or running statistics) + ReLU + dropout.  Xt, Yt: feature-major tiles [F][LDM] in LDS; ysave [M][F] global.
__device__ __forceinline__ void bn_relu_drop(const float* Xt, float* Yt, float* ysave, int M, int F, const float* gamma,
                                             const float* beta, float* rmean, float* rvar, long long* nbt, const float* keep,
                                             float* smean, float* sinv, float eps, float mom, float scale, int training) {
  const int f = threadIdx.x;
  if (f < F) {
    const float* x = Xt + f * LDM;
    float mean, invstd;
    if (training) {
      float s = 0.f, ss = 0.f;
      float xr[TM];      // the batch column in registers (rows beyond M are zero in the tile)
#pragma unroll
      for (int r = 0; r < TM; r += 4) *(f4*)(xr + r) = *(const f4*)(x + r);
#pragma unroll
      for (int r = 0; r < TM; ++r) s += (r < M) ? xr[r] : 0.f;
      mean = s / M;
#pragma unroll
      for (int r = 0; r < TM; ++r) {
        const float d = xr[r] - mean;
        ss += (r < M) ? d * d : 0.f;
      }
      const float var = ss / M;
      invstd = 1.0f / sqrtf(var + eps);
      smean[f] = mean;
      sinv[f] = invstd;
      const float unbiased = M > 1 ? ss / (M - 1) : var;
      rmean[f] = (1.f - mom) * rmean[f] + mom * mean;
      rvar[f] = (1.f - mom) * rvar[f] + mom * unbiased;
      if (nbt && f == 0) *nbt += 1;
    } else {
      mean = rmean[f];
      invstd = 1.0f / sqrtf(rvar[f] + eps);
    }
    const float g = gamma[f] * invstd, b = beta[f];
    for (int r = 0; r < TM; ++r) {
      float v = 0.f;
      if (r < M) {
        v = fmaxf((x[r] - mean) * g + b, 0.f);
        if (keep) v = v * keep[r * LDF + f] * scale;      // keep: row-major LDS tile
        ysave[r * F + f] = v;
      }
      Yt[f * LDM + r] = v;      // rows beyond the batch stay zero: they take part in the 4-row GEMM tiles
    }
  }
}

__global__ __launch_bounds__(TAIL_THREADS) void mlp_tail_fwd_kernel(TailFwd a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *At = sm, *Bt = sm + TH * LDM, *Wt = sm + 2 * TH * LDM;      // activations feature-major [F][LDM]; weight k-major [K][LDF]
  float* Kp = Wt + TH * LDF;                                           // dropout keep mask of the current block, row-major [m][LDF]
  const int Mp = (a.M + 3) & ~3;
  load_cols(At, LDM, a.lin1, a.M, a.H1, TM);
  load_cols(Wt, LDF, a.w2, a.H2, a.H1, a.H2);                        // W2 [H2][H1] -> Wt[k = h1][n = h2]
  if (a.keep1) load_rows(Kp, LDF, a.keep1, a.M, a.H1, a.M);
  __syncthreads();
  bn_relu_drop(At, Bt, a.y1, a.M, a.H1, a.g1, a.b1, a.rm1, a.rv1, a.nbt1, a.keep1 ? Kp : nullptr, a.mean1, a.inv1, a.eps1, a.mom1,
               a.scale1, a.training);
  __syncthreads();
  // lin2[m][n] = bias2[n] + sum_k y1[m][k] W2[n][k]  -> At (feature-major [n][m]) and global
  gemm_kmajor(Bt, LDM, Wt, LDF, Mp, a.H2, a.H1, [&](int m0, int n0, const f4 (&acc)[4]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float bb = a.bias2[n0 + jj];
      f4 col;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        col[ii] = (m0 + ii < a.M) ? acc[ii][jj] + bb : 0.f;
        if (m0 + ii < a.M) a.lin2[(m0 + ii) * a.H2 + n0 + jj] = col[ii];
      }
      *(f4*)(At + (n0 + jj) * LDM + m0) = col;
    }
  });
  __syncthreads();
  load_cols(Wt, LDF, a.wz, a.L, a.H2, a.L);                          // Wz [L][H2] -> Wt[k = h2][n = l]
  if (a.keep2) load_rows(Kp, LDF, a.keep2, a.M, a.H2, a.M);
  __syncthreads();
  bn_relu_drop(At, Bt, a.y2, a.M, a.H2, a.g2, a.b2, a.rm2, a.rv2, a.nbt2, a.keep2 ? Kp : nullptr, a.mean2, a.inv2, a.eps2, a.mom2,
               a.scale2, a.training);
  __syncthreads();
  gemm_kmajor(Bt, LDM, Wt, LDF, Mp, a.L, a.H2, [&](int m0, int n0, const f4 (&acc)[4]) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
      if (m0 + ii < a.M) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) a.z[(m0 + ii) * a.L + n0 + jj] = acc[ii][jj] + a.bz[n0 + jj];
      }
  });
}

// BatchNorm + ReLU + dropout backward for feature f = thread: dY row-major [m][LDF] in LDS (overwritten with dX), also written
// feature-major into dXt [f][LDM]; yv, xv: the block's output / input as row-major LDS tiles [m][LDF].
__device__ __forceinline__ void bn_relu_drop_bwd(float* dY, float* dXt, const float* yv, const float* xv, int M, int F,
                                                 const float* gamma, const float* keep, const float* smean, const float* sinv,
                                                 const float* rmean, const float* rvar, float* dgamma, float* dbeta, float eps,
                                                 float scale, int training) {
  const int f = threadIdx.x;
  if (f < F) {
    const float mean = training ? smean[f] : rmean[f];
    const float invstd = training ? sinv[f] : 1.0f / sqrtf(rvar[f] + eps);
    const float ks = keep ? scale : 1.f;
    float sdz = 0.f, sdzx = 0.f;
    for (int r = 0; r < M; ++r) {
      const float dzv = (yv[r * LDF + f] > 0.f) ? dY[r * LDF + f] * ks : 0.f;   // y > 0  <=>  ReLU open AND unit kept
      sdz += dzv;
      sdzx += dzv * (xv[r * LDF + f] - mean) * invstd;
    }
    dbeta[f] = sdz;
    dgamma[f] = sdzx;
    const float g = gamma[f] * invstd, inv_rows = 1.f / M;
    for (int r = 0; r < TM; ++r) {
      float v = 0.f;
      if (r < M) {
        const float dzv = (yv[r * LDF + f] > 0.f) ? dY[r * LDF + f] * ks : 0.f;
        const float xhat = (xv[r * LDF + f] - mean) * invstd;
        v = training ? g * (dzv - inv_rows * (sdz + xhat * sdzx)) : g * dzv;
      }
      dY[r * LDF + f] = v;
      if (dXt) dXt[f * LDM + r] = v;
    }
  }
}

__global__ __launch_bounds__(TAIL_THREADS) void mlp_tail_bwd_kernel(TailBwd a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *G = sm, *P = sm + TM * LDF, *R = sm + 2 * TM * LDF, *Q = sm + 3 * TM * LDF, *Gt = sm + 4 * TM * LDF, *W = Gt + TH * LDM;
  const int Mp = (a.M + 3) & ~3;
  // fc_z_out: G = dz [m][l], Gt = dz^T [l][m], P = y2 [m][h2], W = Wz [l][h2]
  load_rows(G, LDF, a.dz, a.M, a.L, TM);
  load_cols(Gt, LDM, a.dz, a.M, a.L, TM);
  load_rows(P, LDF, a.y2, a.M, a.H2, TM);
  load_rows(Q, LDF, a.lin2, a.M, a.H2, a.M);
  load_rows(W, LDF, a.wz, a.L, a.H2, a.L);
  __syncthreads();
  gemm_kmajor(G, LDF, P, LDF, a.L, a.H2, a.M, [&](int l0, int k0, const f4 (&acc)[4]) {      // dWz[l][k] = sum_m dz[m][l] y2[m][k]
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) *(f4*)(a.dwz + (l0 + ii) * a.H2 + k0) = acc[ii];
  });
  for (int l = threadIdx.x; l < a.L; l += TAIL_THREADS) {
    float s = 0.f;
    for (int m = 0; m < a.M; ++m) s += G[m * LDF + l];
    a.dbz[l] = s;
  }
  gemm_kmajor(Gt, LDM, W, LDF, Mp, a.H2, a.L, [&](int m0, int k0, const f4 (&acc)[4]) {      // dy2[m][k] = sum_l dz[m][l] Wz[l][k]
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) *(f4*)(R + (m0 + ii) * LDF + k0) = acc[ii];
  });
  __syncthreads();
  bn_relu_drop_bwd(R, Gt, P, Q, a.M, a.H2, a.g2, a.keep2, a.mean2, a.inv2, a.rm2, a.rv2, a.dg2, a.db2, a.eps2, a.scale2,
                   a.training);      // R = dlin2 [m][n], Gt = dlin2^T [n][m]
  load_rows(W, LDF, a.w2, a.H2, a.H1, a.H2);
  __syncthreads();
  load_rows(P, LDF, a.y1, a.M, a.H1, TM);
  load_rows(Q, LDF, a.lin1, a.M, a.H1, a.M);
  __syncthreads();
  gemm_kmajor(R, LDF, P, LDF, a.H2, a.H1, a.M, [&](int n0, int k0, const f4 (&acc)[4]) {     // dW2[n][k] = sum_m dlin2[m][n] y1[m][k]
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) *(f4*)(a.dw2 + (n0 + ii) * a.H1 + k0) = acc[ii];
  });
  for (int n = threadIdx.x; n < a.H2; n += TAIL_THREADS) {
    float s = 0.f;
    for (int m = 0; m < a.M; ++m) s += R[m * LDF + n];
    a.dbias2[n] = s;
  }
  gemm_kmajor(Gt, LDM, W, LDF, Mp, a.H1, a.H2, [&](int m0, int k0, const f4 (&acc)[4]) {     // dy1[m][k] = sum_n dlin2[m][n] W2[n][k]
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) *(f4*)(G + (m0 + ii) * LDF + k0) = acc[ii];
  });
  __syncthreads();
  bn_relu_drop_bwd(G, nullptr, P, Q, a.M, a.H1, a.g1, a.keep1, a.mean1, a.inv1, a.rm1, a.rv1, a.dg1, a.db1, a.eps1,
                   a.scale1, a.training);
  __syncthreads();
  for (int i = threadIdx.x; i < a.M * a.H1; i += TAIL_THREADS) a.dlin1[i] = G[(i / a.H1) * LDF + (i % a.H1)];
}

int check_dims(const char* who, int M, int H1, int H2, int L) {
  DD_REQUIRE(M > 0 && H1 > 0 && H2 > 0 && L > 0, DD_ERR_BAD_ARG, "%s: non-positive size", who);
  DD_REQUIRE(M <= TM && H1 <= TH && H2 <= TH && L <= TH && H1 % 4 == 0 && H2 % 4 == 0 && L % 4 == 0, DD_ERR_UNSUPPORTED,
             "%s: needs M <= %d, widths <= %d and multiples of 4; got M=%d H1=%d H2=%d L=%d", who, TM, TH, M, H1, H2, L);
  return 0;
}

template <typename K>
int allow(K k, size_t bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return dd_fail(DD_ERR_LAUNCH, "mlp_tail: cannot reserve %zu bytes of LDS: %s", bytes, hipGetErrorString(e));
  return 0;
}

}  // namespace

extern "C" {

int dd_mlp_tail_supported(int32_t m, int32_t h1, int32_t h2, int32_t l) {
  return (m > 0 && m <= TM && h1 > 0 && h1 <= TH && h2 > 0 && h2 <= TH && l > 0 && l <= TH && h1 % 4 == 0 && h2 % 4 == 0 && l % 4 == 0) ? 1 : 0;
}

int dd_mlp_tail_fwd(const float* lin1, const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                    int64_t* num_batches_tracked1, const float* keep1, const float* w2, const float* bias2, const float* gamma2,
                    const float* beta2, float* running_mean2, float* running_var2, int64_t* num_batches_tracked2, const float* keep2,
                    const float* wz, const float* bz, float* y1, float* lin2, float* y2, float* z, float* save_mean1,
                    float* save_invstd1, float* save_mean2, float* save_invstd2, int32_t m, int32_t h1, int32_t h2, int32_t l,
                    float eps1, float eps2, float momentum1, float momentum2, float scale1, float scale2, int32_t training,
                    void* stream) {
  if (int rc = check_dims("mlp_tail_fwd", m, h1, h2, l)) return rc;
  DD_REQUIRE(lin1 && gamma1 && beta1 && running_mean1 && running_var1 && w2 && bias2 && gamma2 && beta2 && running_mean2 && running_var2 &&
                 wz && bz && y1 && lin2 && y2 && z && save_mean1 && save_invstd1 && save_mean2 && save_invstd2,
             DD_ERR_BAD_ARG, "mlp_tail_fwd: NULL pointer");
  DD_REQUIRE(!training || m > 1, DD_ERR_UNSUPPORTED, "mlp_tail_fwd: batch statistics need more than 1 row (torch raises too)");
  TailFwd a;
  a.lin1 = lin1; a.g1 = gamma1; a.b1 = beta1; a.keep1 = keep1; a.w2 = w2; a.bias2 = bias2; a.g2 = gamma2; a.b2 = beta2; a.keep2 = keep2;
  a.wz = wz; a.bz = bz; a.rm1 = running_mean1; a.rv1 = running_var1; a.rm2 = running_mean2; a.rv2 = running_var2;
  a.nbt1 = (long long*)(training ? num_batches_tracked1 : nullptr); a.nbt2 = (long long*)(training ? num_batches_tracked2 : nullptr);
  a.y1 = y1; a.lin2 = lin2; a.y2 = y2; a.z = z; a.mean1 = save_mean1; a.inv1 = save_invstd1; a.mean2 = save_mean2; a.inv2 = save_invstd2;
  a.M = m; a.H1 = h1; a.H2 = h2; a.L = l; a.training = training;
  a.eps1 = eps1; a.eps2 = eps2; a.mom1 = momentum1; a.mom2 = momentum2; a.scale1 = scale1; a.scale2 = scale2;
  const size_t lds = ((size_t)2 * TH * LDM + (size_t)TH * LDF + (size_t)TM * LDF) * 4;
  if (int rc = allow(mlp_tail_fwd_kernel, lds)) return rc;
  hipLaunchKernelGGL(mlp_tail_fwd_kernel, dim3(1), dim3(TAIL_THREADS), lds, (hipStream_t)stream, a);
  DD_LAUNCH_CHECK("mlp_tail_fwd");
  return 0;
}

int dd_mlp_tail_bwd(const float* dz, const float* lin1, const float* y1, const float* lin2, const float* y2, const float* gamma1,
                    const float* gamma2, const float* keep1, const float* keep2, const float* w2, const float* wz,
                    const float* save_mean1, const float* save_invstd1, const float* save_mean2, const float* save_invstd2,
                    const float* running_mean1, const float* running_var1, const float* running_mean2, const float* running_var2,
                    float* dlin1, float* dgamma1, float* dbeta1, float* dw2, float* dbias2, float* dgamma2, float* dbeta2, float* dwz,
                    float* dbz, int32_t m, int32_t h1, int32_t h2, int32_t l, float eps1, float eps2, float scale1, float scale2,
                    int32_t training, void* stream) {
  if (int rc = check_dims("mlp_tail_bwd", m, h1, h2, l)) return rc;
  DD_REQUIRE(dz && lin1 && y1 && lin2 && y2 && gamma1 && gamma2 && w2 && wz && dlin1 && dgamma1 && dbeta1 && dw2 && dbias2 && dgamma2 &&
                 dbeta2 && dwz && dbz,
             DD_ERR_BAD_ARG, "mlp_tail_bwd: NULL pointer");
  DD_REQUIRE(training ? (save_mean1 && save_invstd1 && save_mean2 && save_invstd2)
                      : (running_mean1 && running_var1 && running_mean2 && running_var2),
             DD_ERR_BAD_ARG, "mlp_tail_bwd: missing statistics");
  TailBwd a;
  a.dz = dz; a.lin1 = lin1; a.y1 = y1; a.lin2 = lin2; a.y2 = y2; a.g1 = gamma1; a.g2 = gamma2; a.keep1 = keep1; a.keep2 = keep2;
  a.w2 = w2; a.wz = wz; a.mean1 = save_mean1; a.inv1 = save_invstd1; a.mean2 = save_mean2; a.inv2 = save_invstd2;
  a.rm1 = running_mean1; a.rv1 = running_var1; a.rm2 = running_mean2; a.rv2 = running_var2;
  a.dlin1 = dlin1; a.dg1 = dgamma1; a.db1 = dbeta1; a.dw2 = dw2; a.dbias2 = dbias2; a.dg2 = dgamma2; a.db2 = dbeta2; a.dwz = dwz; a.dbz = dbz;
  a.M = m; a.H1 = h1; a.H2 = h2; a.L = l; a.training = training; a.eps1 = eps1; a.eps2 = eps2; a.scale1 = scale1; a.scale2 = scale2;
  const size_t lds = ((size_t)4 * TM * LDF + (size_t)TH * LDM + (size_t)TH * LDF) * 4;
  if (int rc = allow(mlp_tail_bwd_kernel, lds)) return rc;
  hipLaunchKernelGGL(mlp_tail_bwd_kernel, dim3(1), dim3(TAIL_THREADS), lds, (hipStream_t)stream, a);
  DD_LAUNCH_CHECK("mlp_tail_bwd");
  return 0;
}

}  // extern "C"
