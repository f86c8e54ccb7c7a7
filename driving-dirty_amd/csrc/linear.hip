// Skinny fp32 GEMMs of the dense head (kernel family K7) on the fp32 matrix cores.
//
// The reference's Linear layers have a tiny batch dimension (M = batch = 32) and one enormous dimension:
//   encoder fc1   : Linear(940032 -> 128)   reference components.py:27,104-105   (weight 481 MB)
//   roadmap head  : Linear(64 -> 640000)    reference roadmap_bce_v2.py:50,75    (weight 164 MB)
// so every pass is one stream over the weight: HBM-bound (16 flop per weight byte at M = 32 needs the
// matrix cores just to keep up with HBM, a VALU loop could not).  All three passes keep nn.Linear's
// [out, in] weight layout; only the operand roles of the MFMA change:
//   fwd    Y[M,N]  = X[M,K] W[N,K]^T + b     both operands K-contiguous  -> LDS-staged tiles, split-K
//   dgrad  dX[M,K] = dY[M,N] W[N,K]          W rows contiguous along the output -> LDS tile, split-N
//   wgrad  dW[N,K] = dY[M,N]^T X[M,K]        both operands lane-contiguous -> straight from HBM, no LDS
#include "dd_common.h"

namespace {

constexpr int KT = 64;          // k-chunk of the forward kernel
constexpr int WROW = KT + 4;    // LDS row stride in floats: 272 B keeps 16 rows x 16 B on 64 distinct banks
constexpr int NTD = 32;         // n-chunk of the dgrad kernel

// ---------------------------------------------------------------------------------------------- forward
// grid (ceil(N/128), ksplit); 4 waves, wave w owns output columns [nb*128 + 32w, +32); MT M-tiles of 32 rows.
template <int MT>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         float* __restrict__ part, int M, int N, int K,
                                                         int tiles_per_split) {
  __shared__ __attribute__((aligned(16))) float wl[128 * WROW];
  __shared__ __attribute__((aligned(16))) float xl[MT * 32 * WROW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int n0 = blockIdx.x * 128;
  const int ntiles = (K + KT - 1) / KT;
  const int t0 = blockIdx.y * tiles_per_split;
  const int t1 = min(t0 + tiles_per_split, ntiles);
  const int lrow = tid >> 4, lch = tid & 15;    // staging: 16 threads cover one 256-byte row segment

  f32x4 wreg[8], xreg[2 * MT];
  auto fetch = [&](int t) {
    const int kc = t * KT + lch * 4;
    const bool kok = kc < K;                    // K % 4 == 0 is required, so a chunk is all-in or all-out
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int n = n0 + p * 16 + lrow;
      const f32x4 v = __builtin_nontemporal_load((const f32x4*)(Wt + (long)min(n, N - 1) * K + min(kc, K - 4)));      // the weight streams through once
      wreg[p] = (kok && n < N) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 2 * MT; ++p) {
      const int m = p * 16 + lrow;
      const f32x4 v = *(const f32x4*)(X + (long)min(m, M - 1) * K + min(kc, K - 4));
      xreg[p] = (kok && m < M) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

  if (t0 < t1) fetch(t0);
  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int p = 0; p < 8; ++p) *(f32x4*)(wl + (p * 16 + lrow) * WROW + lch * 4) = wreg[p];
#pragma unroll
    for (int p = 0; p < 2 * MT; ++p) *(f32x4*)(xl + (p * 16 + lrow) * WROW + lch * 4) = xreg[p];
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);               // next tile's HBM loads fly under this tile's MFMAs
#pragma unroll
    for (int s = 0; s < KT / 8; ++s) {
      const f32x4 b4 = *(const f32x4*)(wl + (wave * 32 + r) * WROW + 8 * s + 4 * h);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 a4 = *(const f32x4*)(xl + (mt * 32 + r) * WROW + 8 * s + 4 * h);
        acc[mt] = DD_MFMA(a4.x, b4.x, acc[mt]);
        acc[mt] = DD_MFMA(a4.y, b4.y, acc[mt]);
        acc[mt] = DD_MFMA(a4.z, b4.z, acc[mt]);
        acc[mt] = DD_MFMA(a4.w, b4.w, acc[mt]);
      }
    }
    __syncthreads();
  }

  const int n = n0 + wave * 32 + r;
  const bool direct = gridDim.y == 1;
  const float bv = (direct && bias && n < N) ? bias[n] : 0.f;
  float* dst = direct ? Y : part + (long)blockIdx.y * M * N;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = mt * 32 + dd_acc_row(i, lane);
      if (m < M && n < N) dst[(long)m * N + n] = acc[mt][i] + bv;
    }
}

// out[e] = sum_s part[s][e] (+ bias[e % ncols]); fixed order -> deterministic.
// A block owns 8 consecutive elements; its 32 thread groups each sum the splits s = g, g + 32, ... (four independent
// chains), LDS adds the groups in order.  (One thread per element walking all the splits alone was a pure latency
// chain: 41 us for the head's 2048 x 1024 partials.)
__global__ __launch_bounds__(256) void splits_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                            float* __restrict__ out, long elems, int nsplit, int ncols) {
  __shared__ float red[32][8];
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const long e = (long)blockIdx.x * 8 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < elems) {
    int s = g;
    for (; s + 96 < nsplit; s += 128) {
      s0 += part[(long)(s + 0) * elems + e];
      s1 += part[(long)(s + 32) * elems + e];
      s2 += part[(long)(s + 64) * elems + e];
      s3 += part[(long)(s + 96) * elems + e];
    }
    for (; s < nsplit; s += 32) s0 += part[(long)s * elems + e];
  }
  red[g][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0 || e >= elems) return;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) v += red[i][el];
  if (bias) v += bias[e % ncols];
  out[e] = v;
}

// The same for a handful of splits (one thread per element is then the better shape).
__global__ __launch_bounds__(256) void splits_reduce_few_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                float* __restrict__ out, long elems, int nsplit, int ncols) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float v = 0.f;
  for (int s = 0; s < nsplit; ++s) v += part[(long)s * elems + e];
  if (bias) v += bias[e % ncols];
  out[e] = v;
}

static inline void launch_splits_reduce(const float* part, const float* bias, float* out, long elems, int nsplit, int ncols, hipStream_t st) {
  if (nsplit >= 16)
    hipLaunchKernelGGL(splits_reduce_kernel, dim3((unsigned)((elems + 7) / 8)), dim3(256), 0, st, part, bias, out, elems, nsplit, ncols);
  else
    hipLaunchKernelGGL(splits_reduce_few_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, part, bias, out, elems, nsplit, ncols);
}

// ---------------------------------------------------------------------------------------------- dgrad
// grid (ceil(K/128), nsplit); wave w owns output columns [kb*128 + 32w, +32); contraction over n.
template <int MT>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ dY, const float* __restrict__ Wt,
                                                           float* __restrict__ dX, float* __restrict__ part, int M,
                                                           int N, int K, int tiles_per_split) {
  __shared__ __attribute__((aligned(16))) float wl[NTD * 128];            // [n][col], read as 32 consecutive dwords
  __shared__ __attribute__((aligned(16))) float gl[MT * 32 * (NTD + 4)];  // [m][n], 144-byte rows for ds_read_b128
  constexpr int GROW = NTD + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int c0 = blockIdx.x * 128;
  const int ntiles = (N + NTD - 1) / NTD;
  const int t0 = blockIdx.y * tiles_per_split;
  const int t1 = min(t0 + tiles_per_split, ntiles);
  const int wrow = tid >> 5, wch = tid & 31;   // W tile: 32 threads cover one 512-byte row
  const int grow = tid >> 3, gch = tid & 7;    // dY tile: 8 threads cover one 128-byte row

  f32x4 wreg[4], greg[MT];
  auto fetch = [&](int t) {
    const int nb = t * NTD;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int n = nb + p * 8 + wrow, c = c0 + wch * 4;
      const f32x4 v = __builtin_nontemporal_load((const f32x4*)(Wt + (long)min(n, N - 1) * K + min(c, K - 4)));
      wreg[p] = (n < N && c < K) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < MT; ++p) {
      const int m = p * 32 + grow, nn = nb + gch * 4;
      const f32x4 v = *(const f32x4*)(dY + (long)min(m, M - 1) * N + min(nn, N - 4));
      greg[p] = (m < M && nn < N) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

  if (t0 < t1) fetch(t0);
  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *(f32x4*)(wl + (p * 8 + wrow) * 128 + wch * 4) = wreg[p];
#pragma unroll
    for (int p = 0; p < MT; ++p) *(f32x4*)(gl + (p * 32 + grow) * GROW + gch * 4) = greg[p];
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);
#pragma unroll
    for (int s = 0; s < NTD / 8; ++s) {
      float b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = wl[(8 * s + 4 * h + i) * 128 + wave * 32 + r];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 a4 = *(const f32x4*)(gl + (mt * 32 + r) * GROW + 8 * s + 4 * h);
        acc[mt] = DD_MFMA(a4.x, b[0], acc[mt]);
        acc[mt] = DD_MFMA(a4.y, b[1], acc[mt]);
        acc[mt] = DD_MFMA(a4.z, b[2], acc[mt]);
        acc[mt] = DD_MFMA(a4.w, b[3], acc[mt]);
      }
    }
    __syncthreads();
  }

  const int c = c0 + wave * 32 + r;
  float* dst = gridDim.y == 1 ? dX : part + (long)blockIdx.y * M * K;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = mt * 32 + dd_acc_row(i, lane);
      if (m < M && c < K) dst[(long)m * K + c] = acc[mt][i];
    }
}

// ---------------------------------------------------------------------------------------------- wgrad
// One wave = TN n-tiles x TK k-tiles of dW, contraction over the batch rows; operands straight from HBM.
template <int TN, int TK>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                           float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                           int K, int ngroups_k, long total) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wt = (long)blockIdx.x * 4 + wave;
  if (wt >= total) return;
  const int h = lane >> 5, r = lane & 31;
  const int kg = (int)(wt % ngroups_k);
  const long ng = wt / ngroups_k;
  const int nbase = (int)(ng * 32 * TN), kbase = kg * 32 * TK;

  f32x16 acc[TN][TK];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  float bsum[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) bsum[a] = 0.f;

  for (int m0 = 0; m0 < M; m0 += 32) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m = m0 + 2 * s + h;
      const bool mok = m < M;
      const int mc = min(m, M - 1);
      float av[TN], bv[TK];
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int n = nbase + 32 * a + r;
        const float v = dY[(long)mc * N + min(n, N - 1)];
        av[a] = (mok && n < N) ? v : 0.f;
        bsum[a] += av[a];
      }
#pragma unroll
      for (int b = 0; b < TK; ++b) {
        const int k = kbase + 32 * b + r;
        const float v = X[(long)mc * K + min(k, K - 1)];
        bv[b] = (mok && k < K) ? v : 0.f;
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b) acc[a][b] = DD_MFMA(av[a], bv[b], acc[a][b]);
    }
  }

#pragma unroll
  for (int a = 0; a < TN; ++a) {
#pragma unroll
    for (int b = 0; b < TK; ++b) {
      const int k = kbase + 32 * b + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = nbase + 32 * a + dd_acc_row(i, lane);
        if (n < N && k < K) __builtin_nontemporal_store(acc[a][b][i], dW + (long)n * K + k);
      }
    }
    if (db && kg == 0) {
      const float tot = bsum[a] + __shfl_xor(bsum[a], 32);   // the two half-waves hold even / odd batch rows
      const int n = nbase + 32 * a + r;
      if (h == 0 && n < N) db[n] = tot;
    }
  }
}

// Wide variant for long rows (encoder fc1: K = 940032): one wave = TN n-tiles x 128 consecutive k-columns.
// Lane r loads 16 bytes = columns 4r..4r+3 of a batch row (512 contiguous bytes per half-wave), which makes
// four B operands whose "column j" is the strided set {4j + c}; the matching four accumulators hold, per
// lane, four CONSECUTIVE output columns, so dW leaves as 16-byte stores, 512 contiguous bytes per row.
template <int TN>
__global__ __launch_bounds__(256) void linear_wgrad_wide_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                float* __restrict__ dW, float* __restrict__ db, int M,
                                                                int N, int K, int ngroups_k, long total) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wt = (long)blockIdx.x * 4 + wave;
  if (wt >= total) return;
  const int h = lane >> 5, r = lane & 31;
  const int kg = (int)(wt % ngroups_k);
  const int nbase = (int)(wt / ngroups_k) * 32 * TN, kb = kg * 128;

  f32x16 acc[TN][4];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][c][i] = 0.f;
  float bsum[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) bsum[a] = 0.f;
  const int kcol = kb + 4 * r;
  const bool kok = kcol < K;

  for (int m0 = 0; m0 < M; m0 += 32) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m = m0 + 2 * s + h;
      const bool mok = m < M;
      const int mc = min(m, M - 1);
      const f32x4 xv = *(const f32x4*)(X + (long)mc * K + min(kcol, K - 4));
      const f32x4 bv = (mok && kok) ? xv : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int n = nbase + 32 * a + r;
        const float v = dY[(long)mc * N + min(n, N - 1)];
        const float av = (mok && n < N) ? v : 0.f;
        bsum[a] += av;
        acc[a][0] = DD_MFMA(av, bv.x, acc[a][0]);
        acc[a][1] = DD_MFMA(av, bv.y, acc[a][1]);
        acc[a][2] = DD_MFMA(av, bv.z, acc[a][2]);
        acc[a][3] = DD_MFMA(av, bv.w, acc[a][3]);
      }
    }
  }

#pragma unroll
  for (int a = 0; a < TN; ++a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = nbase + 32 * a + dd_acc_row(i, lane);
      if (n < N && kok) __builtin_nontemporal_store(f32x4{acc[a][0][i], acc[a][1][i], acc[a][2][i], acc[a][3][i]}, (f32x4*)(dW + (long)n * K + kcol));
    }
    if (db && kg == 0) {
      const float tot = bsum[a] + __shfl_xor(bsum[a], 32);
      const int n = nbase + 32 * a + r;
      if (h == 0 && n < N) db[n] = tot;
    }
  }
}

// db[n] = sum_m dY[m][n]: the bias gradient alone, for a layer whose weight gradient is formed elsewhere (dd_adam_step_rankb without a
// registered bias, ddp factor mode without rank-B).  One thread per 4 columns, rows in order: deterministic.
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ dY, float* __restrict__ db, int M, int N) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= N) return;
  if (c + 4 <= N && N % 4 == 0) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < M; ++m) s += *(const f32x4*)(dY + (long)m * N + c);
    *(f32x4*)(db + c) = s;
  } else {
    for (int j = c; j < min(c + 4, N); ++j) {
      float s = 0.f;
      for (int m = 0; m < M; ++m) s += dY[(long)m * N + j];
      db[j] = s;
    }
  }
}

int pick_mt(int M) { return M <= 32 ? 1 : (M <= 64 ? 2 : 0); }   // 64 KB of static LDS caps the batch tile at 64 rows

int pick_split(int blocks_other, int ntiles) {
  if (ntiles <= 8) return 1;      // a handful of tiles: one block walks them faster than a second launch can add the splits up
  int want = (2 * DD_NUM_CU + blocks_other - 1) / blocks_other;   // aim at ~2 blocks per CU
  want = max(1, min(want, ntiles));
  return min(want, 512);
}

int check_linear(const char* who, int M, int N, int K) {
  DD_REQUIRE(M > 0 && N > 0 && K > 0, DD_ERR_BAD_ARG, "%s: non-positive size", who);
  DD_REQUIRE(pick_mt(M) != 0, DD_ERR_UNSUPPORTED, "%s: batch rows M = %d > 64", who, M);
  DD_REQUIRE(K % 4 == 0 && N % 4 == 0, DD_ERR_UNSUPPORTED, "%s: N = %d and K = %d must be multiples of 4", who, N, K);
  return 0;
}

}  // namespace

extern "C" {

int64_t dd_linear_workspace_bytes(int32_t m, int32_t n, int32_t k) {
  // worst case of the forward (split-K partials [512][M][N]) and dgrad (split-N partials [512][M][K]) passes,
  // only when that pass actually splits; never more than 512 slabs of the SMALLER output
  const int64_t fwd = ((n + 127) / 128 >= 2 * DD_NUM_CU) ? 0 : (int64_t)512 * m * n * 4;
  const int64_t dgr = ((k + 127) / 128 >= 2 * DD_NUM_CU) ? 0 : (int64_t)512 * m * k * 4;
  return (fwd > dgr ? fwd : dgr) + 16;
}

int dd_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t m, int32_t n, int32_t k,
                  void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_linear("linear_fwd", m, n, k)) return rc;
  DD_REQUIRE(x && w && y, DD_ERR_BAD_ARG, "linear_fwd: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (n + 127) / 128, ntiles = (k + KT - 1) / KT;
  const int split = pick_split(nblk, ntiles);
  const int tps = (ntiles + split - 1) / split;
  const int nsplit = (ntiles + tps - 1) / tps;
  float* part = (float*)workspace;
  if (nsplit > 1)
    DD_REQUIRE(workspace && workspace_bytes >= (int64_t)nsplit * m * n * 4, DD_ERR_WORKSPACE, "linear_fwd: workspace too small");
  const dim3 grid(nblk, nsplit);
  switch (pick_mt(m)) {
    case 1: hipLaunchKernelGGL(linear_fwd_kernel<1>, grid, dim3(256), 0, st, x, w, bias, y, part, m, n, k, tps); break;
    default: hipLaunchKernelGGL(linear_fwd_kernel<2>, grid, dim3(256), 0, st, x, w, bias, y, part, m, n, k, tps); break;
  }
  DD_LAUNCH_CHECK("linear_fwd");
  if (nsplit > 1) {
    const long elems = (long)m * n;
    launch_splits_reduce(part, bias, y, elems, nsplit, n, st);
    DD_LAUNCH_CHECK("linear_fwd reduce");
  }
  return 0;
}

int dd_linear_dgrad(const float* dy, const float* w, float* dx, int32_t m, int32_t n, int32_t k, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  if (int rc = check_linear("linear_dgrad", m, n, k)) return rc;
  DD_REQUIRE(dy && w && dx, DD_ERR_BAD_ARG, "linear_dgrad: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const int kblk = (k + 127) / 128, ntiles = (n + NTD - 1) / NTD;
  const int split = pick_split(kblk, ntiles);
  const int tps = (ntiles + split - 1) / split;
  const int nsplit = (ntiles + tps - 1) / tps;
  float* part = (float*)workspace;
  if (nsplit > 1)
    DD_REQUIRE(workspace && workspace_bytes >= (int64_t)nsplit * m * k * 4, DD_ERR_WORKSPACE, "linear_dgrad: workspace too small");
  const dim3 grid(kblk, nsplit);
  switch (pick_mt(m)) {
    case 1: hipLaunchKernelGGL(linear_dgrad_kernel<1>, grid, dim3(256), 0, st, dy, w, dx, part, m, n, k, tps); break;
    default: hipLaunchKernelGGL(linear_dgrad_kernel<2>, grid, dim3(256), 0, st, dy, w, dx, part, m, n, k, tps); break;
  }
  DD_LAUNCH_CHECK("linear_dgrad");
  if (nsplit > 1) {
    const long elems = (long)m * k;
    launch_splits_reduce(part, nullptr, dx, elems, nsplit, k, st);
    DD_LAUNCH_CHECK("linear_dgrad reduce");
  }
  return 0;
}

int dd_column_sum(const float* dy, float* dbias, int32_t m, int32_t n, void* stream) {
  DD_REQUIRE(dy && dbias && m > 0 && n > 0, DD_ERR_BAD_ARG, "column_sum: bad argument");
  DD_REQUIRE(n % 4 != 0 || ((uintptr_t)dy | (uintptr_t)dbias) % 16 == 0, DD_ERR_BAD_ARG, "column_sum: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(column_sum_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, dy, dbias, m, n);
  DD_LAUNCH_CHECK("column_sum");
  return 0;
}

int dd_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, int32_t m, int32_t n, int32_t k,
                    void* stream) {
  DD_REQUIRE(m > 0 && n > 0 && k > 0, DD_ERR_BAD_ARG, "linear_wgrad: non-positive size");
  DD_REQUIRE(dy && x && dw, DD_ERR_BAD_ARG, "linear_wgrad: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  if (k >= 512 && k % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dw % 16 == 0)) {   // long rows: 16-byte path
    const int gk = (k + 127) / 128;
    // 32 x 128 tiles (TN = 1): four times the waves of the 64 x 128 form re-read x from L2 but keep far more of the
    // 481 MB write stream in flight (measured on fc1: 0.254 -> 0.19 ms)
    const long total = (long)((n + 31) / 32) * gk;
    hipLaunchKernelGGL((linear_wgrad_wide_kernel<1>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, dy, x, dw,
                       dbias, m, n, k, gk, total);
  } else if (n <= 256 || k > n) {   // few output rows: one k-tile x up to 4 n-tiles per wave
    const int gk = (k + 31) / 32;
    const long total = (long)((n + 127) / 128) * gk;
    hipLaunchKernelGGL((linear_wgrad_kernel<4, 1>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, dy, x, dw, dbias,
                       m, n, k, gk, total);
  } else {                   // many output rows, short rows (roadmap head: K = 64): one n-tile x 2 k-tiles per wave
    const int gk = (k + 63) / 64;
    const long total = (long)((n + 31) / 32) * gk;
    hipLaunchKernelGGL((linear_wgrad_kernel<1, 2>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, dy, x, dw, dbias,
                       m, n, k, gk, total);
  }
  DD_LAUNCH_CHECK("linear_wgrad");
  return 0;
}

}  // extern "C"
