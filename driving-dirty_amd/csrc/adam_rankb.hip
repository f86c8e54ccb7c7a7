// dd_adam_step_rankb: the Adam pass of a big Linear weight with its weight gradient formed INSIDE the pass.
//
// The weight gradient of a Linear layer over an M-row batch is a rank-M product, dW[n][k] = sum_b dY[b][n] X[b][k] with M <= 32
// per GPU (M = world x batch when the factors were all-gathered: ddp.GradSync factor mode).  For the three tensors that ARE the
// optimizer's traffic -- the encoder's fc1 940032 -> 128 (reference components.py:27,105; 481 MB), the road-map head 64 -> 640000
// (roadmap_bce_v2.py:50,75; 164 MB), the decoder's fc2 128 -> 1253376 (components.py:70; 642 MB) -- writing dW (dd_linear_wgrad) and
// reading it back (dd_adam_step) are two full passes over the largest tensors of the step that nothing needs: here every gradient
// element is produced in MFMA accumulators from M products and consumed by the Adam update of the same lane.  HBM traffic per element:
// p, m, v read + written = 6 passes (24 B) instead of 8 (wgrad's write + Adam's 7), plus the factors (X: 120 MB for fc1, dY: 82 MB for
// the head), which the weight-gradient kernel read too.
//
// Shape.  The pass runs BESIDE the conv backward (optim.HipAdam, side stream), whose one-wave-per-SIMD kernels leave 48 of a SIMD's
// 512 registers: so one persistent workgroup per CU and a wave tile small enough for that budget -- 16 output rows x
// 64 columns in four v_mfma_f32_16x16x4_f32 accumulators (16 registers).  Lane (r = lane & 15, q = lane >> 4) loads 16 bytes
// X[4s + q][k0 + 4r .. 4r + 3] per contraction step: four B operands whose "column j" is the strided set {4j + c}, so accumulator c
// register i holds dW[n0 + 4q + i][k0 + 4r + c] and the four accumulators give each lane FOUR CONSECUTIVE columns of a row:
// p / m / v move as 16-byte accesses, 256 contiguous bytes per quarter-wave.  Tiles are numbered n-tile fastest, so the waves of a
// workgroup (and of its neighbour) share one X tile through the caches while each streams its own rows of p, m, v.
//
// The bias of the layer (optional): its gradient is the column sum of dY, which the k0 = 0 tile of every n-tile has in registers
// anyway; that wave applies the same Adam update to bias / its moments -- no second pass over dY (82 MB for the head), no ATen sum.
#include <math.h>
#include <stdlib.h>

#include "dd_common.h"
#include "dd_adam.h"

namespace {

// D(16x16) += A(16x4) B(4x16), exact fp32.  This file is compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py): the accumulators stay in
// ordinary vector registers -- left to the default the four accumulators go to the accumulation registers and are COPIED to vector
// registers for the epilogue: 32 registers for 16, and the whole kernel has 72.
__device__ __forceinline__ void mfma16(f32x4& acc, float a, float b) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }

struct RankbArgs {
  float* p;
  float* m;
  float* v;
  const float* dy;
  const float* x;
  float* bp;
  float* bm;
  float* bv;
  int M, N, K;
  int ntile_n, ntile_k;
  int per;                 // group-tiles per workgroup: workgroup b takes [b * per, (b + 1) * per) of the (n-group, k-tile) list, k fastest
  int total;               // ceil(ntile_n / 4) * ntile_k group-tiles
  float b1, b2, omb1, omb2, eps, step_size, inv_bc2, bc2_sqrt, gscale;
};

__device__ __forceinline__ f32x4 nt_load4(__amdgpu_buffer_rsrc_t r, int off, int soff) {      // streaming: nt
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 2));
}
__device__ __forceinline__ void nt_store4(__amdgpu_buffer_rsrc_t r, int off, int soff, f32x4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, soff, 2);
}

// ---- the LDS form: the shipped one ------------------------------------------------------------------------------------------------
// Same tile, same order, same arithmetic; what changes is how often a wave waits for memory.  Beside the conv backward a round trip to
// HBM takes 2-3 us, and the form above makes four per tile (two batches of factor loads, two of p / m / v): 1.5 TB/s in the step where
// the plain Adam kernel moves 2.3.  Here the workgroup's X tile (32 batch rows x 64 columns, 8 KB, shared by its four waves) is loaded
// ONE TILE AHEAD into 8 registers per lane while the current tile's p / m / v are on their way, and passes through LDS; the workgroup's
// dY tile (batch rows x 64 outputs) is loaded once per n-group -- the waves stay on their rows while they walk along k -- and stays in
// LDS.  A tile then costs the two round trips of its p / m / v and nothing else.  MC = 32-row chunks of the batch (1: rows <= 32; 2:
// rows <= 64, the gathered factors of two ranks); LDS 8 KB (X) + MC x 8 KB (dY): fits beside the c2 weight gradient's 133 KB.
// The tile epilogue shared by the two LDS forms: accumulator c, register i of lane (r, q) is dW[16 nt + 4q + i][64 kt + 4r + c]; p / m / v of
// RG rows per lane are in flight together (one round trip per RG rows, 12 registers per row).
template <int RG>
__device__ __forceinline__ void rankb_epilogue(const RankbArgs& a, int nt, int kt, const f32x4& acc0, const f32x4& acc1, const f32x4& acc2,
                                               const f32x4& acc3) {
  const int lane = dd_fresh_lane();
  const int r = lane & 15, q = lane >> 4;
  const long tile = (long)nt * 16 * a.K + kt * 64;
  const long left = ((long)a.N * a.K - tile) * 4;          // bytes from the tile's first element to the end of the tensor
  const int span = (int)min(left, (long)16 * a.K * 4);    // 16 rows: < 2^31 (host check); rows past N: out of range, dropped
  const __amdgpu_buffer_rsrc_t ps = dd_rsrc(a.p + tile, span), ms = dd_rsrc(a.m + tile, span), vs = dd_rsrc(a.v + tile, span);
  if (kt * 64 + 4 * r >= a.K) return;                     // K % 4 == 0: a 16-byte group is all in or all out
  const int off = (4 * q * a.K + 4 * r) * 4;
#pragma unroll
  for (int i0 = 0; i0 < 4; i0 += RG) {
    f32x4 pa[RG], ma[RG], va[RG];
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int so = (i0 + j) * a.K * 4;
      pa[j] = nt_load4(ps, off, so);
      ma[j] = nt_load4(ms, off, so);
      va[j] = nt_load4(vs, off, so);
    }
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int i = i0 + j, so = i * a.K * 4;
      const f32x4 g = {acc0[i], acc1[i], acc2[i], acc3[i]};
#pragma unroll
      for (int cc = 0; cc < 4; cc += 2) {
        f32x2a pe = {pa[j][cc], pa[j][cc + 1]}, me = {ma[j][cc], ma[j][cc + 1]}, ve = {va[j][cc], va[j][cc + 1]};
        adam_elem2(pe, me, ve, f32x2a{g[cc], g[cc + 1]}, a.gscale, a.b1, a.b2, a.omb1, a.omb2, a.eps, a.step_size, a.inv_bc2);
        pa[j][cc] = pe.x; pa[j][cc + 1] = pe.y; ma[j][cc] = me.x; ma[j][cc + 1] = me.y; va[j][cc] = ve.x; va[j][cc + 1] = ve.y;
      }
      nt_store4(ps, off, so, pa[j]);
      nt_store4(ms, off, so, ma[j]);
      nt_store4(vs, off, so, va[j]);
    }
  }
}

// bias[16 nt .. 16 nt + 15] from the column sums of the dY tile in LDS (the k-tile-0 wave of an n-tile owns them)
__device__ __forceinline__ void rankb_bias(const RankbArgs& a, const float* yl, int nt, int wave) {
  const int lane = dd_fresh_lane();
  const int r = lane & 15, nn = nt * 16 + r;
  if ((lane >> 4) != 0 || nn >= a.N) return;
  float tot = 0.f;
  for (int m = 0; m < a.M; ++m) tot += yl[m * 64 + ((16 * wave + r) ^ (16 * (m & 3)))];
  float pe = a.bp[nn], me = a.bm[nn], ve = a.bv[nn];
  adam_elem(pe, me, ve, tot, a.gscale, a.b1, a.b2, a.eps, a.step_size, a.bc2_sqrt);
  a.bp[nn] = pe; a.bm[nn] = me; a.bv[nn] = ve;
}

// contraction over one 32-row chunk: A = dY (yp: this lane's column, swizzled), B = X (xp: this lane's 4 columns), both from LDS
template <int XS = 64>      // XS: floats of a batch row of the X tile in LDS
__device__ __forceinline__ void rankb_mfma_chunk(const float* yp, const float* xp, int rows_left, f32x4& acc0, f32x4& acc1, f32x4& acc2,
                                                 f32x4& acc3) {
  const int pairs = min(4, (rows_left + 7) / 8);          // two contraction steps per trip (rows past M are zeros in LDS): the second
  for (int s = 0; s < 2 * pairs; s += 2) {                // pair of LDS reads is issued under the first four MFMAs
    const float av0 = yp[s * 256], av1 = yp[s * 256 + 256];
    const f32x4 bv0 = *(const f32x4*)(xp + s * 4 * XS), bv1 = *(const f32x4*)(xp + (s + 1) * 4 * XS);
    mfma16(acc0, av0, bv0.x);
    mfma16(acc1, av0, bv0.y);
    mfma16(acc2, av0, bv0.z);
    mfma16(acc3, av0, bv0.w);
    mfma16(acc0, av1, bv1.x);
    mfma16(acc1, av1, bv1.y);
    mfma16(acc2, av1, bv1.z);
    mfma16(acc3, av1, bv1.w);
  }
}

// LONG rows (many k-tiles per n-group: the encoder's fc1).  MC = 32-row chunks of the batch (1: rows <= 32; 2: rows <= 64, the gathered
// factors of two ranks); LDS 8 KB (X) + MC x 8 KB (dY): fits beside the c2 weight gradient's 133 KB.
template <int MC, int RG>
__global__ __launch_bounds__(256) void adam_rankb_lds_kernel(const RankbArgs a) {
  __shared__ __attribute__((aligned(16))) float xl[32 * 64];
  __shared__ __attribute__((aligned(16))) float yl[MC * 32 * 64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t xs = dd_rsrc(a.x, a.M * a.K * 4), ys = dd_rsrc(a.dy, a.M * a.N * 4);
  const int lo = min((int)blockIdx.x * a.per, a.total), hi = min(lo + a.per, a.total);
  if (lo >= hi) return;
  int gq = lo / a.ntile_k;
  int kt = lo - gq * a.ntile_k;
  int group_in_lds = -1;
  const int nfill = (hi - lo) * MC;                       // fills of xl: (tile, chunk) pairs, chunk fastest

  // Register discipline (the budget is 72, of which 16 are accumulators, 8 the X tile in flight and 24 the p / m / v in flight):
  // nothing lane-dependent is kept across the loop that one or two instructions can rebuild, so every phase starts from a FRESH lane id
  // (dd_fresh_lane: opaque to the optimiser, which otherwise hoists a dozen addresses and masks out of the loop and keeps them).
  // Staging role of a thread: rows srow = tid / 16 and srow + 16 of a 32-row chunk, 16 bytes at column 4 (tid % 16).
  f32x4 xr0, xr1;
  auto fetch_x = [&](int tile_kt, int c) {                // X of a fill -> xr0 / xr1 (zeros past M by the descriptor, past K by the select)
    const int t = wave * 64 + dd_fresh_lane();
    const bool ok = tile_kt * 64 + 4 * (t & 15) < a.K;
    const int off = ((t >> 4) * a.K + (ok ? 4 * (t & 15) : 0)) * 4;
    const int so = (c * 32 * a.K + tile_kt * 64) * 4;
    const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, off, so, 0));
    const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, off, so + 16 * a.K * 4, 0));
    xr0 = ok ? v0 : f32x4{0.f, 0.f, 0.f, 0.f};
    xr1 = ok ? v1 : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  fetch_x(kt, 0);

  f32x4 acc0, acc1, acc2, acc3;
  int f = 0;
  for (int it = lo; it < hi; ++it) {
    const int nt = gq * 4 + wave;
    for (int c = 0; c < MC; ++c, ++f) {
      __syncthreads();                                    // every wave is done with the previous fill's xl (and the previous group's yl)
      {
        const int t = wave * 64 + dd_fresh_lane();
        *(f32x4*)(xl + t * 4) = xr0;                      // row t / 16, column 4 (t % 16): [row][64]
        *(f32x4*)(xl + 16 * 64 + t * 4) = xr1;
        if (gq != group_in_lds) {                         // workgroup-uniform: first tile of an n-group in this workgroup's piece
          const int col = gq * 64 + 4 * (t & 15);
          const bool ok = col < a.N;                      // N % 4 == 0 (host check)
#pragma unroll
          for (int j = 0; j < 2 * MC; ++j) {
            const int row = (t >> 4) + 16 * j;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ys, (row * a.N + (ok ? col : 0)) * 4, 0, 0));
            // columns swizzled by the row's low bits: the four batch rows a matrix instruction reads together land in four different
            // bank groups (plain [row][64] puts all four on the same 16 banks)
            *(f32x4*)(yl + row * 64 + ((4 * (t & 15)) ^ (16 * (row & 3)))) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          }
          group_in_lds = gq;
        }
      }
      __syncthreads();
      if (f + 1 < nfill) {                                // the next fill's X: in flight through this fill's MFMAs and the tile's epilogue
        const bool wrap = (c + 1 == MC);
        fetch_x(wrap ? (kt + 1 == a.ntile_k ? 0 : kt + 1) : kt, wrap ? 0 : c + 1);
      }
      if (c == 0) acc0 = acc1 = acc2 = acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nt < a.ntile_n) {
        const int lane = dd_fresh_lane();
        const int r = lane & 15, q = lane >> 4;           // batch row 4 s + q of the chunk: its low bits are q
        rankb_mfma_chunk(yl + (c * 32 + q) * 64 + ((16 * wave + r) ^ (16 * q)), xl + q * 64 + 4 * r, a.M - c * 32, acc0, acc1, acc2, acc3);
      }
    }
    if (nt < a.ntile_n) {
      rankb_epilogue<RG>(a, nt, kt, acc0, acc1, acc2, acc3);
      if (a.bp && kt == 0) rankb_bias(a, yl, nt, wave);
    }
    gq += (kt + 1 == a.ntile_k);
    kt = (kt + 1 == a.ntile_k) ? 0 : kt + 1;
  }
}

// The same pass when it runs BY ITSELF (bf16 models: after the backward, dd_set_adam_blocks_per_cu > 1; no register or LDS budget): a tile
// of ONE n-tile (16 weight rows) x 256 columns, the four waves on four adjacent 64-column slabs of the same rows.  Above, a workgroup's
// tile is 64 rows x 64 columns: 256 bytes of 64 different rows (16 MB apart for the 2x-resolution fc1) per tensor and visit -- 49 k DRAM
// streams advancing 256 bytes at a time across the chip, which some boxes of the pool serve at 5.2 TB/s where a contiguous stream gets
// 5.7-6.1.  Here a visit is 1 KB of 16 rows.  X tile 32 x 256 (32 KB), one tile ahead in 32 registers; dY tile as above (its first n-tile).
__global__ __launch_bounds__(256) void adam_rankb_wide_kernel(const RankbArgs a) {
  __shared__ __attribute__((aligned(16))) float xl[32 * 256];
  __shared__ __attribute__((aligned(16))) float yl[32 * 64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t xs = dd_rsrc(a.x, a.M * a.K * 4), ys = dd_rsrc(a.dy, a.M * a.N * 4);
  const int ntk = (a.K + 255) / 256;                       // 256-column tiles per n-tile; the list is (n-tile, tile), tile fastest
  const int lo = min((int)blockIdx.x * a.per, a.total), hi = min(lo + a.per, a.total);
  if (lo >= hi) return;
  int nt = lo / ntk;
  int kt = lo - nt * ntk;
  int nt_in_lds = -1;
  f32x4 xr[8];                                              // thread t: batch rows t / 16 and t / 16 + 16, columns 64 j + 4 (t % 16), j = 0 .. 3
  auto fetch_x = [&](int tile) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = tile * 256 + 64 * j + 4 * (t & 15);
      const bool ok = col < a.K;
      const int off = ((t >> 4) * a.K + (ok ? col : 0)) * 4;
      const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, off, 0, 0));
      const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, off, 16 * a.K * 4, 0));
      xr[2 * j] = ok ? v0 : f32x4{0.f, 0.f, 0.f, 0.f};
      xr[2 * j + 1] = ok ? v1 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch_x(kt);
  for (int it = lo; it < hi; ++it) {
    __syncthreads();                                        // every wave is done with the previous tile's xl (and the previous n-tile's yl)
    {
      const int t = threadIdx.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *(f32x4*)(xl + (t >> 4) * 256 + 64 * j + 4 * (t & 15)) = xr[2 * j];
        *(f32x4*)(xl + ((t >> 4) + 16) * 256 + 64 * j + 4 * (t & 15)) = xr[2 * j + 1];
      }
      if (nt != nt_in_lds) {                                // dY columns 16 nt .. 16 nt + 63 (the first 16 are this n-tile's), swizzled as above
        const int col = nt * 16 + 4 * (t & 15);
        const bool ok = col < a.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = (t >> 4) + 16 * j;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ys, (row * a.N + (ok ? col : 0)) * 4, 0, 0));
          *(f32x4*)(yl + row * 64 + ((4 * (t & 15)) ^ (16 * (row & 3)))) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        nt_in_lds = nt;
      }
    }
    __syncthreads();
    const bool last_of_row = kt + 1 == ntk;
    if (it + 1 < hi) fetch_x(last_of_row ? 0 : kt + 1);     // the next tile's X: in flight through this tile's MFMAs and epilogue
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    {
      const int lane = dd_fresh_lane();
      const int r = lane & 15, q = lane >> 4;
      rankb_mfma_chunk<256>(yl + q * 64 + (r ^ (16 * q)), xl + q * 256 + 64 * wave + 4 * r, a.M, acc0, acc1, acc2, acc3);
    }
    rankb_epilogue<2>(a, nt, 4 * kt + wave, acc0, acc1, acc2, acc3);      // (returns at once for a slab past K)
    if (a.bp && kt == 0 && wave == 0) rankb_bias(a, yl, nt, 0);
    nt += last_of_row;
    kt = last_of_row ? 0 : kt + 1;
  }
}

// SHORT rows (one or two k-tiles per n-group: the road-map head K = 64, the decoder's fc2 K = 128; rows <= 32).  Here the roles swap: X
// (32 x K, at most 16 KB) is the same for every tile and stays in LDS for the whole launch, and it is the dY tile that changes -- with
// every n-group, i.e. every one or two tiles -- so dY is what travels one group ahead through the 8 prefetch registers.  A tile again
// costs the two round trips of its p / m / v only (the long-row form run on these shapes reloads dY behind a barrier for every tile:
// 0.77 ms for the head beside the c2 weight gradient against 0.2 alone).  NK = k-tiles per group (1 or 2); LDS NK x 8 KB + 8 KB.
template <int NK, int RG>
__global__ __launch_bounds__(256) void adam_rankb_short_kernel(const RankbArgs a) {
  __shared__ __attribute__((aligned(16))) float xl[NK * 32 * 64];
  __shared__ __attribute__((aligned(16))) float yl[32 * 64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t xs = dd_rsrc(a.x, a.M * a.K * 4), ys = dd_rsrc(a.dy, a.M * a.N * 4);
  const int lo = min((int)blockIdx.x * a.per, a.total), hi = min(lo + a.per, a.total);      // here the list is of n-GROUPS (host: total, per)
  if (lo >= hi) return;
  {                                                       // X, once: [k-tile][row][64]
    const int t = wave * 64 + dd_fresh_lane();
#pragma unroll
    for (int k2 = 0; k2 < NK; ++k2) {
      const bool ok = k2 * 64 + 4 * (t & 15) < a.K;
      const int off = ((t >> 4) * a.K + (ok ? k2 * 64 + 4 * (t & 15) : 0)) * 4;
      const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, off, 0, 0));
      const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, off, 16 * a.K * 4, 0));
      *(f32x4*)(xl + k2 * 2048 + t * 4) = ok ? v0 : f32x4{0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(xl + k2 * 2048 + 1024 + t * 4) = ok ? v1 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 yr0, yr1;                                         // the dY tile of an n-group in flight: rows t / 16 and t / 16 + 16, 4 columns
  auto fetch_y = [&](int group) {
    const int t = wave * 64 + dd_fresh_lane();
    const int col = group * 64 + 4 * (t & 15);
    const bool ok = col < a.N;                            // N % 4 == 0 (host check); a group past the last one: all zeros, never used
    const int off = ((t >> 4) * a.N + (ok ? col : 0)) * 4;
    const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ys, off, 0, 0));
    const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ys, off, 16 * a.N * 4, 0));
    yr0 = ok ? v0 : f32x4{0.f, 0.f, 0.f, 0.f};
    yr1 = ok ? v1 : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  fetch_y(lo);
  for (int gq = lo; gq < hi; ++gq) {
    const int nt = gq * 4 + wave;
    __syncthreads();                                      // every wave is done with the previous group's yl (first trip: X is written)
    {
      const int t = wave * 64 + dd_fresh_lane();
      const int row = t >> 4;                             // (row + 16) & 3 == row & 3: one swizzle for both rows
      *(f32x4*)(yl + row * 64 + ((4 * (t & 15)) ^ (16 * (row & 3)))) = yr0;
      *(f32x4*)(yl + (row + 16) * 64 + ((4 * (t & 15)) ^ (16 * (row & 3)))) = yr1;
    }
    __syncthreads();
    fetch_y(gq + 1);                                      // in flight through this group's MFMAs and epilogues
    if (nt < a.ntile_n) {
#pragma nounroll
      for (int kt = 0; kt < NK; ++kt) {                   // one tile at a time: two accumulator sets alive at once would not fit the budget
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        const int lane = dd_fresh_lane();
        const int r = lane & 15, q = lane >> 4;
        rankb_mfma_chunk(yl + q * 64 + ((16 * wave + r) ^ (16 * q)), xl + kt * 2048 + q * 64 + 4 * r, a.M, acc0, acc1, acc2, acc3);
        rankb_epilogue<RG>(a, nt, kt, acc0, acc1, acc2, acc3);
      }
      if (a.bp) rankb_bias(a, yl, nt, wave);
    }
  }
}

}  // namespace

extern "C" {

int dd_adam_step_rankb(float* p, float* m, float* v, const float* dy, const float* x, int32_t rows, int32_t n, int32_t k,
                       float* bias_p, float* bias_m, float* bias_v, float lr, float beta1, float beta2, float eps, int32_t step,
                       float grad_scale, void* stream) {
  DD_REQUIRE(p && m && v && dy && x && step >= 1, DD_ERR_BAD_ARG, "adam_rankb: bad argument");
  DD_REQUIRE(rows > 0 && n > 0 && k > 0, DD_ERR_BAD_ARG, "adam_rankb: non-positive size");
  DD_REQUIRE(rows <= 64, DD_ERR_UNSUPPORTED, "adam_rankb: %d batch rows > 64", rows);
  DD_REQUIRE(k % 4 == 0 && n % 4 == 0, DD_ERR_UNSUPPORTED, "adam_rankb: N = %d and K = %d must be multiples of 4", n, k);
  DD_REQUIRE(((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)x | (uintptr_t)dy) % 16 == 0, DD_ERR_BAD_ARG, "adam_rankb: buffers must be 16-byte aligned");
  DD_REQUIRE((bias_p != nullptr) == (bias_m != nullptr) && (bias_p != nullptr) == (bias_v != nullptr), DD_ERR_BAD_ARG,
             "adam_rankb: bias, its exp_avg and exp_avg_sq come together or not at all");
  DD_REQUIRE((int64_t)(rows + 4) * n < ((int64_t)1 << 29) && (int64_t)(rows + 4) * k < ((int64_t)1 << 29) && (int64_t)16 * k < ((int64_t)1 << 29),
             DD_ERR_UNSUPPORTED, "adam_rankb: factors of 2 GB or more");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  RankbArgs a;
  a.p = p; a.m = m; a.v = v; a.dy = dy; a.x = x; a.bp = bias_p; a.bm = bias_m; a.bv = bias_v;
  a.M = rows; a.N = n; a.K = k;
  a.ntile_n = (n + 15) / 16;
  a.ntile_k = (k + 63) / 64;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  a.omb1 = 1.f - beta1; a.omb2 = 1.f - beta2;
  a.step_size = lr / (float)bc1;
  a.bc2_sqrt = (float)sqrt(bc2);
  a.inv_bc2 = 1.f / a.bc2_sqrt;
  a.gscale = grad_scale;
  const bool short_rows = rows <= 32 && a.ntile_k == 1;      // adam_rankb_short_kernel: its work list is of n-groups
  // the pass by itself (more than one workgroup per CU: nothing to fit beside) on long rows: 16-row x 256-column tiles
  const bool wide = rows <= 32 && !short_rows && a.ntile_k >= 64 && dd_adam_blocks_internal() > 1;
  const long total = wide ? (long)a.ntile_n * ((k + 255) / 256) : (long)((a.ntile_n + 3) / 4) * (short_rows ? 1 : a.ntile_k);
  DD_REQUIRE(total < ((long)1 << 31), DD_ERR_UNSUPPORTED, "adam_rankb: too many tiles");
  a.total = (int)total;
  // one persistent workgroup per CU, as dd_adam_step (dense.hip): nothing of this launch is ever queued ahead of a conv kernel
  const int per_cu = dd_adam_blocks_internal();
  const int grid = (int)min(total, (long)max(DD_NUM_CU - dd_adam_spare_internal(), 1) * per_cu);
  a.per = (int)((total + grid - 1) / grid);
  hipStream_t st = (hipStream_t)stream;
  // (two k-tiles per group -- the decoder's fc2, K = 128 -- ran on a <2, RG = 1> build of the short form (its RG = 2 build needs 76 registers):
  // 2.98 ms beside the conv backward for 0.80 alone; the long-row form below keeps two rows of p / m / v in flight for it)
  if (wide) hipLaunchKernelGGL(adam_rankb_wide_kernel, dim3(grid), dim3(256), 0, st, a);
  else if (short_rows) hipLaunchKernelGGL((adam_rankb_short_kernel<1, 2>), dim3(grid), dim3(256), 0, st, a);
  else if (rows <= 32) hipLaunchKernelGGL((adam_rankb_lds_kernel<1, 2>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((adam_rankb_lds_kernel<2, 1>), dim3(grid), dim3(256), 0, st, a);
  DD_LAUNCH_CHECK("adam_rankb");
  return 0;
}

}  // extern "C"
