// Dilated stride-1 convolutions of the box heads on the fp32 matrix cores, phase-decomposed and LDS-staged:
//
//   RoadMapBoxesMergingCNN  up_conv_1..4  ConvTranspose2d 96->64, 64->32, 32->16 (k7 d7), 16->8 (k7 d3)   spatial_bb/components.py:135-138
//   BoxesMergingCNN         up_conv_1..3  ConvTranspose2d 64->32, 32->16 (k8 d8), 16->8 (k6 d6, output_padding 2)      :90-92
//
// forward (flipped-tap gather, pad d(k-1)) and data gradient (plain gather, pad 0) of each: 31.5 of the 33.6 G MAC of
// the head.  The generic engine (gconv.hip) gathers the A operand of every MFMA from L1/L2 and streams the weight image
// per 32-pixel row tile: 256 B of operand traffic per MFMA, 46-55 % of the matrix peak.  Here:
//
//   * rows of equal residue  oy mod d  share their input rows (taps are d rows apart): a workgroup owns RA output rows of
//     ONE residue class ("phase rows") x XT output columns and needs only RA + k - 1 input rows for all k tap rows;
//   * the input patch of one 8-channel chunk -- (RA + k - 1) rows x (XT + d(k-1)) columns x 32 B -- sits in LDS (double
//     buffered: chunk q + 1 is fetched while chunk q is multiplied); every tap of every wave reads it with one
//     ds_read_b128 per 4 MFMAs at an address = per-lane constant + tap offset (1 KB contiguous per wave: conflict-free);
//   * a wave owns 64 pixels (two 32-pixel m-tiles) of one phase row x all Cout (NT column tiles): the weight fragment of a
//     (chunk, tap) is loaded once per 8*NT MFMAs with lane-linear 16-byte loads at a SCALAR offset (no address VALU);
//   * taps that touch no input pixel for the whole wave (border rows / columns of the flipped form) are skipped.
//
// GEMM per (wave, chunk, tap): M = 64 pixels, N = Cout, K = 8 channels; k index pairing of v_mfma_f32_32x32x2_f32:
// lane (h = l >> 5, n = l & 31) holds channels 4h..4h+3 of pixel n (A) and of output column n (B); MFMA i multiplies
// channels (i, 4 + i).
// Cout <= 16 (up_conv_3 forward, up_conv_4 both ways) would leave half or more of a 32-wide column tile empty: those run on
// v_mfma_f32_16x16x4_f32 (NT = 0 below: same flop rate, N = 16): four 16-pixel m-tiles per wave, lane (g = l >> 4, m = l & 15)
// holds channels 2g, 2g+1 of pixel m (A, one ds_read_b64) and of column m (B); MFMA j multiplies channels (j, 2+j, 4+j, 6+j).
#define DD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#include <stdlib.h>

#include "dd_common.h"

namespace {

constexpr int DC_THREADS = 512;

template <int K, int D>
struct DcGeom {
  static constexpr int HALO = D * (K - 1);
  static constexpr int TWA = 128 + HALO, THA = 4 + K - 1;      // shape A: 4 phase rows x 128 columns
  static constexpr int TWB = 64 + HALO, THB = 8 + K - 1;       // shape B: 8 phase rows x 64 columns
  static constexpr int PXA = TWA * THA, PXB = TWB * THB;
  static constexpr int PX = PXA > PXB ? PXA : PXB;              // pixels of one LDS buffer (32 B each)
  static constexpr int NP = (2 * PX + DC_THREADS - 1) / DC_THREADS;      // 16-byte pieces per thread and fill
};

struct DcTile {
  int b, r, a0, x0, ra, xt;      // image, residue, first phase row, first column, rows and columns of the tile
};

// Tiles are numbered image-major, then residue, then shape-A tiles (x tile, row group), then shape-B tiles.
struct DcPlan {
  int rows_max;          // phase rows of the longest residue class: ceil(out_h / D)
  int nxa, nxb;          // x tiles of shape A (128 wide) and of shape B (64 wide, at most one, the last)
  int ga, gb;            // row groups of 4 / of 8
  int per_res, total;
};

__host__ __device__ inline DcPlan dc_plan(const dd_gconv_desc& d, int D) {
  DcPlan p;
  p.rows_max = (d.out_h + D - 1) / D;
  p.nxa = d.out_w / 128;
  const int rem = d.out_w - p.nxa * 128;
  p.nxb = 0;
  if (rem > 64) p.nxa += 1; else if (rem > 0) p.nxb = 1;
  p.ga = (p.rows_max + 3) / 4;
  p.gb = (p.rows_max + 7) / 8;
  p.per_res = p.nxa * p.ga + p.nxb * p.gb;
  p.total = p.per_res * D * d.batch;
  return p;
}

__device__ __forceinline__ DcTile dc_tile(const DcPlan& p, int D, int t) {
  DcTile o;
  const int per_img = p.per_res * D;
  o.b = t / per_img;
  t -= o.b * per_img;
  o.r = t / p.per_res;
  t -= o.r * p.per_res;
  if (t < p.nxa * p.ga) {
    const int xt = t / p.ga;
    o.ra = 4; o.xt = 128; o.x0 = xt * 128; o.a0 = (t - xt * p.ga) * 4;
  } else {
    t -= p.nxa * p.ga;
    o.ra = 8; o.xt = 64; o.x0 = p.nxa * 128; o.a0 = t * 8;
  }
  return o;
}

template <int K, int D, int NT>
__global__ __launch_bounds__(DC_THREADS) void dconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, const float* __restrict__ msk,
                                                               float* __restrict__ y, const dd_gconv_desc d, int epi, int wp_bytes, int dbg_repeat) {
  using G = DcGeom<K, D>;
  constexpr int T = K * K;
  constexpr bool N16 = NT == 0;                 // 16-wide column tile on the 16x16x4 MFMA
  constexpr int NTR = N16 ? 1 : NT;             // column tiles held in registers
  __shared__ __attribute__((aligned(16))) float lds[2][G::PX * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31;
  const DcPlan plan = dc_plan(d, D);
  const int NC = d.cin >> 3;
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  const __amdgpu_buffer_rsrc_t ws = dd_rsrc(wp, wp_bytes);

  float bv[NTR];
#pragma unroll
  for (int nt = 0; nt < NTR; ++nt) {
    const int ch = N16 ? (lane & 15) : nt * 32 + n;
    bv[nt] = (bias && (epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) && ch < d.cout) ? bias[ch] : 0.f;
  }

  // ---- fill state: the 16-byte pieces this thread fetches for a tile (piece p = pixel p >> 1, channel half p & 1).  The
  // offsets live in LDS, each thread's own column: they are computed once per tile and read once per chunk, and as NP
  // registers held across the whole tap loop they were what the register allocator spilled to scratch (ISA: 9 scratch
  // reloads per chunk in dconv_fwd_kernel<7,7,2>).
  __shared__ int voff[G::NP * DC_THREADS];
  auto plan_fill = [&](const DcTile& tl) {
    const int tid = dd_fresh_lane() + 64 * wave;
    const int tw = tl.xt + G::HALO, npc = (tl.ra + K - 1) * tw * 2;
    const int iy0 = tl.r + D * tl.a0 - d.pad_h, ix0 = tl.x0 - d.pad_w;
#pragma unroll
    for (int i = 0; i < G::NP; ++i) {
      const int p = tid + DC_THREADS * i, px = p >> 1;
      const int l = px / tw, lam = px - l * tw;
      const int iy = iy0 + D * l, ix = ix0 + lam;
      const bool ok = p < npc && (unsigned)iy < (unsigned)d.in_h && (unsigned)ix < (unsigned)d.in_w;
      voff[i * DC_THREADS + tid] = ok ? ((iy * d.in_w + ix) * d.in_cstore + d.in_coff + 4 * (p & 1)) * 4 : -16;
    }
  };

  int t = blockIdx.x;
  if (t >= plan.total) return;
  DcTile tile = dc_tile(plan, D, t);
  plan_fill(tile);
  {   // first chunk of the first tile
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)tile.b * d.in_h * d.in_w * d.in_cstore, in_bytes);
#pragma unroll
    for (int i = 0; i < G::NP; ++i) {
      const f32x4 v = dd_bload4(xs, voff[i * DC_THREADS + tid]);
      if (tid + DC_THREADS * i < 2 * G::PX) *(f32x4*)&lds[0][(tid + DC_THREADS * i) * 4] = v;      // pieces past the tile: zeros into the slack
    }
  }
  __syncthreads();
  int par = 0;
  for (; t < plan.total; t += gridDim.x) {
    const int tnext = t + gridDim.x;
    // ---- this wave's share of the tile
    const int lane = dd_fresh_lane(), h = lane >> 5, n = lane & 31;      // per tile: see dd_fresh_lane
    const int tw = tile.xt + G::HALO;
    const int row = tile.ra == 4 ? (wave & 3) : wave, xh = tile.ra == 4 ? (wave >> 2) : 0;
    const int a = tile.a0 + row, oy = tile.r + D * a;
    const int xw = tile.x0 + 64 * xh;
    const bool row_ok = oy < d.out_h && xw < d.out_w;
    const bool mt1 = xw + 32 < d.out_w;
    // valid tap rows / columns (the whole wave): input row oy + D*ky - pad_h in [0, in_h), some column of the wave's
    // 64 pixels xw + [0, 63] + D*kx - pad_w in [0, in_w)
    int ky0 = 0, ky1 = -1, kx0 = 0, kx1 = -1;
    if (row_ok) {
      const int ry = oy - d.pad_h;
      ky0 = ry >= 0 ? 0 : (-ry + D - 1) / D;
      ky1 = min(K - 1, (d.in_h - 1 - ry) >= 0 ? (d.in_h - 1 - ry) / D : -1);
      const int xlo = xw - d.pad_w, xhi = min(xw + 63, d.out_w - 1) - d.pad_w;
      kx0 = xhi >= 0 ? 0 : (-xhi + D - 1) / D;
      kx1 = min(K - 1, (d.in_w - 1 - xlo) >= 0 ? (d.in_w - 1 - xlo) / D : -1);
    }
    const int nky = max(ky1 - ky0 + 1, 0), nkx = max(kx1 - kx0 + 1, 0), ntaps = nky * nkx;
    // byte offset of this lane's pixel / channel group in an LDS buffer
    const int abase = N16 ? ((row * tw + 64 * xh + (lane & 15)) * 8 + 2 * (lane >> 4)) * 4 : ((row * tw + 64 * xh + n) * 8 + 4 * h) * 4;

    f32x16 acc[2][NTR];
    f32x4 acc16[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int nt = 0; nt < NTR; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) acc16[t4] = f32x4{0.f, 0.f, 0.f, 0.f};

    DcTile next = tile;
    for (int q = 0; q < NC; ++q) {
      // ---- request the next fill (next chunk of this tile, or chunk 0 of the workgroup's next tile)
      const bool last = q + 1 == NC;
      bool have_next = true;
      if (last) {
        have_next = tnext < plan.total;
        if (have_next) {
          next = dc_tile(plan, D, tnext);
          plan_fill(next);
        }
      }
      f32x4 stage[G::NP];
      if (have_next) {
        const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)next.b * d.in_h * d.in_w * d.in_cstore, in_bytes);
        const int soff = last ? 0 : 32 * (q + 1);
#pragma unroll
        for (int i = 0; i < G::NP; ++i)
          stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, voff[i * DC_THREADS + tid], soff, 0));
      }
      // ---- multiply chunk q out of lds[par]
      if (ntaps > 0) {
        const char* lbase = (const char*)&lds[par][0] + abase;
        int ky = ky0, kx = kx0;
        f32x4 A0[2], A1[2], B[2][NTR];
        f32x2 P[2][4], Q[2];                  // N16: four 16-pixel A fragments and the weight fragment of a tap
        auto load_tap = [&](int slot, int cky, int ckx) {
          const char* p = lbase + (cky * tw + ckx * D) * 32;
          if constexpr (N16) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) P[slot][t4] = *(const f32x2*)(p + 512 * t4);
            Q[slot] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ws, lane * 8, (q * T + cky * K + ckx) * 512, 0));
          } else {
            A0[slot] = *(const f32x4*)p;
            A1[slot] = *(const f32x4*)(p + 1024);
            const int soff = ((q * T + cky * K + ckx) * NTR) * 1024;
#pragma unroll
            for (int nt = 0; nt < NTR; ++nt)
              B[slot][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, lane * 16, soff + nt * 1024, 0));
          }
        };
        auto mul_tap = [&](int slot) {
          if constexpr (N16) {
            acc16[0] = DD_MFMA16(P[slot][0].x, Q[slot].x, acc16[0]);
            acc16[1] = DD_MFMA16(P[slot][1].x, Q[slot].x, acc16[1]);
            if (mt1) {
              acc16[2] = DD_MFMA16(P[slot][2].x, Q[slot].x, acc16[2]);
              acc16[3] = DD_MFMA16(P[slot][3].x, Q[slot].x, acc16[3]);
            }
            acc16[0] = DD_MFMA16(P[slot][0].y, Q[slot].y, acc16[0]);
            acc16[1] = DD_MFMA16(P[slot][1].y, Q[slot].y, acc16[1]);
            if (mt1) {
              acc16[2] = DD_MFMA16(P[slot][2].y, Q[slot].y, acc16[2]);
              acc16[3] = DD_MFMA16(P[slot][3].y, Q[slot].y, acc16[3]);
            }
          } else {
#pragma unroll
            for (int nt = 0; nt < NTR; ++nt) {
              acc[0][nt] = DD_MFMA(A0[slot].x, B[slot][nt].x, acc[0][nt]);
              acc[0][nt] = DD_MFMA(A0[slot].y, B[slot][nt].y, acc[0][nt]);
              acc[0][nt] = DD_MFMA(A0[slot].z, B[slot][nt].z, acc[0][nt]);
              acc[0][nt] = DD_MFMA(A0[slot].w, B[slot][nt].w, acc[0][nt]);
            }
            if (mt1) {
#pragma unroll
              for (int nt = 0; nt < NTR; ++nt) {
                acc[1][nt] = DD_MFMA(A1[slot].x, B[slot][nt].x, acc[1][nt]);
                acc[1][nt] = DD_MFMA(A1[slot].y, B[slot][nt].y, acc[1][nt]);
                acc[1][nt] = DD_MFMA(A1[slot].z, B[slot][nt].z, acc[1][nt]);
                acc[1][nt] = DD_MFMA(A1[slot].w, B[slot][nt].w, acc[1][nt]);
              }
            }
          }
        };
        // The operand loads of tap i+1 are issued before tap i is multiplied, UNCONDITIONALLY: a load inside an `if` makes the
        // compiler's s_waitcnt at the join assume it was not issued, i.e. wait for everything (measured in the ISA: vmcnt(0)
        // in front of the third MFMA of every tap, the prefetch's whole L2 latency exposed).  Past the last tap the iterator
        // stays on the last tap: one redundant, harmless reload.
        int left = ntaps - 1;                                      // taps after the one (ky, kx) points at
        auto advance = [&]() {
          const bool more = left > 0;
          left -= more ? 1 : 0;
          const bool wrap = kx >= kx1;
          kx = more ? (wrap ? kx0 : kx + 1) : kx;
          ky = (more && wrap) ? ky + 1 : ky;
        };
        // Every tap valid for this wave (data gradients; interior waves of the padded layers) and at most two column tiles: tap
        // columns unrolled, the operands of tap (ky + 1, kx) requested into the registers tap (ky, kx) has just been multiplied
        // from -- no (ky, kx) iterator, no operand-address arithmetic per tap (T(r) of the iterator loop: 17 % over its MFMAs)
        bool done = false;
        // (a whole tap row of operands is K * (8 + 4 * NTR) registers: 112 for K = 7 with two column tiles, 128 for K = 8 -- the
        // latter, beside 64 accumulators and the staged fill, spilled to scratch INSIDE the tap loop (448 B per lane): K = 8 with
        // two column tiles -- BoxesMergingCNN's up_conv_1 data gradient -- stays on the iterator loop below)
        if constexpr (!N16 && NTR <= 2 && K * (8 + 4 * NTR) <= 112) {
          if (nky == K && nkx == K && dbg_repeat == 1) {
            done = true;
            f32x4 Aq0[K], Aq1[K], Bq[K][NTR];
#pragma unroll
            for (int c = 0; c < K; ++c) {
              const char* p = lbase + (c * D) * 32;
              Aq0[c] = *(const f32x4*)p;
              Aq1[c] = *(const f32x4*)(p + 1024);
#pragma unroll
              for (int nt = 0; nt < NTR; ++nt)
                Bq[c][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, lane * 16, ((q * T + c) * NTR + nt) * 1024, 0));
            }
            for (int r = 0; r < K; ++r) {
              const int rn = min(r + 1, K - 1);                  // past the last tap row: harmless reloads of it
              const char* prow = lbase + (rn * tw) * 32;
              const int srow = (q * T + rn * K) * NTR * 1024;
#pragma unroll
              for (int c = 0; c < K; ++c) {
#pragma unroll
                for (int nt = 0; nt < NTR; ++nt) {
                  acc[0][nt] = DD_MFMA(Aq0[c].x, Bq[c][nt].x, acc[0][nt]);
                  acc[0][nt] = DD_MFMA(Aq0[c].y, Bq[c][nt].y, acc[0][nt]);
                  acc[0][nt] = DD_MFMA(Aq0[c].z, Bq[c][nt].z, acc[0][nt]);
                  acc[0][nt] = DD_MFMA(Aq0[c].w, Bq[c][nt].w, acc[0][nt]);
                }
                if (mt1) {
#pragma unroll
                  for (int nt = 0; nt < NTR; ++nt) {
                    acc[1][nt] = DD_MFMA(Aq1[c].x, Bq[c][nt].x, acc[1][nt]);
                    acc[1][nt] = DD_MFMA(Aq1[c].y, Bq[c][nt].y, acc[1][nt]);
                    acc[1][nt] = DD_MFMA(Aq1[c].z, Bq[c][nt].z, acc[1][nt]);
                    acc[1][nt] = DD_MFMA(Aq1[c].w, Bq[c][nt].w, acc[1][nt]);
                  }
                }
                Aq0[c] = *(const f32x4*)(prow + (c * D) * 32);
                Aq1[c] = *(const f32x4*)(prow + (c * D) * 32 + 1024);
#pragma unroll
                for (int nt = 0; nt < NTR; ++nt)
                  Bq[c][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, lane * 16, srow + (c * NTR + nt) * 1024, 0));
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        }
        if constexpr (N16) {      // the same for the 16-wide form: four 16-pixel A fragments and one weight fragment per tap
          if (nky == K && nkx == K && dbg_repeat == 1) {
            done = true;
            f32x2 Pq[K][4], Qq[K];
#pragma unroll
            for (int c = 0; c < K; ++c) {
              const char* p = lbase + (c * D) * 32;
#pragma unroll
              for (int t4 = 0; t4 < 4; ++t4) Pq[c][t4] = *(const f32x2*)(p + 512 * t4);
              Qq[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ws, lane * 8, (q * T + c) * 512, 0));
            }
            for (int r = 0; r < K; ++r) {
              const int rn = min(r + 1, K - 1);
              const char* prow = lbase + (rn * tw) * 32;
              const int srow = (q * T + rn * K) * 512;
#pragma unroll
              for (int c = 0; c < K; ++c) {
                acc16[0] = DD_MFMA16(Pq[c][0].x, Qq[c].x, acc16[0]);
                acc16[1] = DD_MFMA16(Pq[c][1].x, Qq[c].x, acc16[1]);
                if (mt1) {
                  acc16[2] = DD_MFMA16(Pq[c][2].x, Qq[c].x, acc16[2]);
                  acc16[3] = DD_MFMA16(Pq[c][3].x, Qq[c].x, acc16[3]);
                }
                acc16[0] = DD_MFMA16(Pq[c][0].y, Qq[c].y, acc16[0]);
                acc16[1] = DD_MFMA16(Pq[c][1].y, Qq[c].y, acc16[1]);
                if (mt1) {
                  acc16[2] = DD_MFMA16(Pq[c][2].y, Qq[c].y, acc16[2]);
                  acc16[3] = DD_MFMA16(Pq[c][3].y, Qq[c].y, acc16[3]);
                }
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) Pq[c][t4] = *(const f32x2*)(prow + (c * D) * 32 + 512 * t4);
                Qq[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ws, lane * 8, srow + c * 512, 0));
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        }
        if (!done)
        for (int rep = 0; rep < dbg_repeat; ++rep) {      // dbg_repeat = 1 (DD_DCONV_REPEAT: timing diagnostic only, results are then wrong)
        ky = ky0; kx = kx0; left = ntaps - 1;
        load_tap(0, ky, kx);
        advance();
        for (int i = 0; i < ntaps; i += 2) {
          load_tap(1, ky, kx);
          advance();
          __builtin_amdgcn_sched_barrier(0);
          mul_tap(0);
          load_tap(0, ky, kx);
          advance();
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < ntaps) mul_tap(1);
        }
        }
      }
      // ---- retire the staged pieces into the other buffer (read last one step ago, a barrier since)
      if (have_next) {
        const int tid = dd_fresh_lane() + 64 * wave;
#pragma unroll
        for (int i = 0; i < G::NP; ++i)
          if (tid + DC_THREADS * i < 2 * G::PX) *(f32x4*)&lds[par ^ 1][(tid + DC_THREADS * i) * 4] = stage[i];
      }
      __syncthreads();
      par ^= 1;
    }

    // ---- epilogue of the tile
    if (row_ok) {
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)tile.b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const __amdgpu_buffer_rsrc_t ms = dd_rsrc(msk ? msk + (long)tile.b * d.omem_h * d.omem_w * d.out_cstore : y, msk ? out_bytes : 0);
      // All mask values of a 32x32 (16x16) tile are requested before any of them is used: one conditional load per element
      // (`epi` is a run-time value) made the compiler wait for each load in turn -- 64..96 serial L2 round trips per tile,
      // 1.0 .. 1.2 ms of a 5.7 .. 11 ms data-gradient launch (measured by repeating the tap loop, DD_DCONV_REPEAT).
      const bool masked = epi == DD_EPI_RELU_MASK;
      const int base = ((oy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff;
      const int lane = dd_fresh_lane(), h = lane >> 5, n = lane & 31;      // shadows the kernel's: nothing lane-derived crosses the tap loop
      if constexpr (N16) {
        const int ch = lane & 15;
        const bool pass = d.out_coff + ch >= d.mask_pass_lo && d.out_coff + ch < d.mask_pass_hi;
        int off[16];
        float mv[16];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int xo = xw + 16 * t4 + 4 * (lane >> 4) + r;      // D row of the 16x16 tile: 4*(lane >> 4) + r
            const bool ok = xo < d.out_w && ch < d.cout && (t4 < 2 || mt1);
            off[4 * t4 + r] = ok ? (base + xo * d.out_cstore + ch) * 4 : -16;
            mv[4 * t4 + r] = 1.f;
          }
        if (masked && !pass) {
#pragma unroll
          for (int e = 0; e < 16; ++e) mv[e] = dd_bload1(ms, off[e]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc16[e >> 2][e & 3] + bv[0];
          if (epi == DD_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
          v = mv[e] > 0.f ? v : 0.f;
          dd_bstore1(ys, off[e], v);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j == 1 && !mt1) break;
#pragma unroll
          for (int nt = 0; nt < NTR; ++nt) {
            const int ch = nt * 32 + n;
            const bool pass = d.out_coff + ch >= d.mask_pass_lo && d.out_coff + ch < d.mask_pass_hi;
            int off[16];
            float mv[16];
            // Element r of the 32 x 32 tile sits at pixel xw + 32j + 4h + c_r, c_r = (r & 3) + 8 (r >> 2): its byte offset is a
            // per-lane part (one register) plus a wave-uniform part (scalar registers).  Written as one expression the
            // lane-variant, tile-invariant terms ((4h + c_r) * cstore + ch) * 4 of all 16 elements were hoisted out of the
            // persistent tile loop and lived across it: 16 registers the tap loop does not have (scratch spills).  The asm
            // makes the per-lane part opaque to that hoisting.
            int lane_off = ((4 * h) * d.out_cstore + ch) * 4;
            int lim = d.out_w - (xw + 32 * j + 4 * h);                    // element r is inside the row iff c_r < lim
            asm volatile("" : "+v"(lane_off), "+v"(lim));
            lim = ch < d.cout ? lim : 0;
            const int tile_off = (base + (xw + 32 * j) * d.out_cstore) * 4;      // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cr = (r & 3) + 8 * (r >> 2);
              off[r] = cr < lim ? lane_off + (tile_off + cr * d.out_cstore * 4) : -16;
              mv[r] = 1.f;
            }
            if (masked) {      // lanes of an exempt channel ask a zero-size resource (no memory request) and keep 1
              const __amdgpu_buffer_rsrc_t mr = ms;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float m = dd_bload1(mr, pass ? -16 : off[r]);
                mv[r] = pass ? 1.f : m;
              }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float v = acc[j][nt][r] + bv[nt];
              if (epi == DD_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
              v = mv[r] > 0.f ? v : 0.f;
              dd_bstore1(ys, off[r], v);
            }
          }
        }
      }
    }
    tile = next;
  }
}

// packed[(((q*T + tap)*NT + nt)*64 + lane)*4 + i] = W(n = nt*32 + (lane & 31), c = 8q + 4*(lane >> 5) + i, tap);
// Cout <= 16 (nt_count == 0): packed[((q*T + tap)*64 + lane)*2 + j] = W(n = lane & 15, c = 8q + 2*(lane >> 4) + j, tap)
__global__ void dconv_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int nchunks, int nt_count, int T, long w_off,
                                  long sn, long sc, int flip, int n_real, int c_real) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (nt_count == 0) {
    if (idx >= (long)nchunks * T * 128) return;
    const int j = idx & 1, lane = (idx >> 1) & 63;
    const long g = idx >> 7;
    const int tap = (int)(g % T), q = (int)(g / T);
    const int c = 8 * q + 2 * (lane >> 4) + j, n = lane & 15;
    p[idx] = (n < n_real && c < c_real) ? w[w_off + n * sn + c * sc + (flip ? T - 1 - tap : tap)] : 0.f;
    return;
  }
  const long total = (long)nchunks * T * nt_count * 256;
  if (idx >= total) return;
  const int i = idx & 3, lane = (idx >> 2) & 63;
  long g = idx >> 8;
  const int nt = (int)(g % nt_count);
  g /= nt_count;
  const int tap = (int)(g % T), q = (int)(g / T);
  const int c = 8 * q + 4 * (lane >> 5) + i, n = nt * 32 + (lane & 31);
  float v = 0.f;
  if (n < n_real && c < c_real) v = w[w_off + n * sn + c * sc + (flip ? T - 1 - tap : tap)];
  p[idx] = v;
}

int dc_variant(const dd_gconv_desc* d) {      // index of the (K, D) instantiation, -1 if this is not one of the box heads' layers
  if (d->kh != d->kw || d->dil_h != d->dil_w) return -1;
  const int k = d->kh, dl = d->dil_h;
  if (k == 7 && dl == 7) return 0;
  if (k == 7 && dl == 3) return 1;
  if (k == 8 && dl == 8) return 2;
  if (k == 6 && dl == 6) return 3;
  if (k == 3 && dl == 1) return 4;      // out_conv, the decoder's dc1 / dc2 (one residue class: plain LDS-tiled 3x3)
  if (k == 3 && dl == 3) return 5;      // rm_conv_2
  return -1;
}

bool dc_supported(const dd_gconv_desc* d) {
  return d && dc_variant(d) >= 0 && d->stride_h == 1 && d->stride_w == 1 && d->div_h == 1 && d->div_w == 1 && d->ostride_h == 1 &&
         d->ostride_w == 1 && d->cin > 0 && d->cin % 8 == 0 && d->in_coff % 4 == 0 && d->in_cstore % 4 == 0 &&
         d->in_coff + d->cin <= d->in_cstore && d->cout > 0 && d->cout <= 96 && d->out_coff >= 0 && d->out_coff + d->cout <= d->out_cstore &&
         d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->out_h > 0 && d->out_w > 0 && d->pad_h >= 0 && d->pad_w >= 0 &&
         d->ooff_h >= 0 && d->ooff_w >= 0 && d->out_h + d->ooff_h <= d->omem_h && d->out_w + d->ooff_w <= d->omem_w &&
         (long)d->in_h * d->in_w * d->in_cstore * 4 < (1L << 31) && (long)d->omem_h * d->omem_w * d->out_cstore * 4 < (1L << 31);
}

}  // namespace

bool dd_dconv_desc_ok(const dd_gconv_desc* d) { return dc_supported(d); }      // for the launchers in dconv_t.hip

extern "C" {

int32_t dd_dconv_supported(const dd_gconv_desc* d) { return dc_supported(d) ? 1 : 0; }

int64_t dd_dconv_packed_floats(const dd_gconv_desc* d) {
  if (!dc_supported(d)) {
    dd_fail(DD_ERR_UNSUPPORTED, "dconv: not a stride-1 k7d7 / k7d3 / k8d8 / k6d6 / k3d1 / k3d3 layer with Cin %% 8 == 0 and Cout <= 96");
    return -1;
  }
  if (d->cout <= 16) return (int64_t)(d->cin / 8) * d->kh * d->kw * 128;
  return (int64_t)(d->cin / 8) * d->kh * d->kw * ((d->cout + 31) / 32) * 256;
}

int dd_dconv_pack(const float* w, float* packed, const dd_gconv_desc* d, int64_t w_off, int64_t sn, int64_t sc, int32_t flip,
                  int32_t n_real, int32_t c_real, void* stream) {
  DD_REQUIRE(dc_supported(d), DD_ERR_UNSUPPORTED, "dconv_pack: unsupported layer");
  DD_REQUIRE(w && packed, DD_ERR_BAD_ARG, "dconv_pack: NULL pointer");
  DD_REQUIRE(n_real > 0 && n_real <= d->cout && c_real > 0 && c_real <= d->cin, DD_ERR_BAD_ARG, "dconv_pack: n_real/c_real");
  const int nt = d->cout <= 16 ? 0 : (d->cout + 31) / 32, T = d->kh * d->kw;
  const long total = dd_dconv_packed_floats(d);
  hipLaunchKernelGGL(dconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, d->cin / 8, nt,
                     T, (long)w_off, (long)sn, (long)sc, flip, n_real, c_real);
  DD_LAUNCH_CHECK("dconv_pack");
  return 0;
}

int dd_dconv_fwd(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                 int32_t epilogue, void* stream) {
  DD_REQUIRE(dc_supported(d), DD_ERR_UNSUPPORTED, "dconv_fwd: unsupported layer");
  DD_REQUIRE(x && packed && y, DD_ERR_BAD_ARG, "dconv_fwd: NULL pointer");
  DD_REQUIRE(epilogue == DD_EPI_NONE || epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU || epilogue == DD_EPI_RELU_MASK,
             DD_ERR_BAD_ARG, "dconv_fwd: epilogue %d", epilogue);
  DD_REQUIRE(epilogue != DD_EPI_RELU_MASK || mask, DD_ERR_BAD_ARG, "dconv_fwd: RELU_MASK needs a mask");
  DD_REQUIRE(!(epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU) || bias, DD_ERR_BAD_ARG, "dconv_fwd: bias epilogue needs a bias");
  hipStream_t st = (hipStream_t)stream;
  const int nt = d->cout <= 16 ? 0 : (d->cout + 31) / 32;
  const int wp_bytes = (int)(dd_dconv_packed_floats(d) * 4);
#ifdef DD_TIMING_DIAG      // diagnostic builds only (hipcc -DDD_TIMING_DIAG): repeating the tap loop times it, the results are then wrong
  static const int dbg_repeat = getenv("DD_DCONV_REPEAT") ? atoi(getenv("DD_DCONV_REPEAT")) : 1;
#else
  constexpr int dbg_repeat = 1;
#endif
  // rows that are not a whole number of 8 m-tiles (up_conv_2 both ways, up_conv_3's data gradient): several rows per workgroup (dconv_m.hip)
  if (dd_dconv_mfwd_launch(x, packed, bias, mask, y, d, epilogue, wp_bytes, st)) {
    DD_LAUNCH_CHECK("dconv_mfwd");
    return 0;
  }
  // the full transposed forward of the wide layers: input-aligned tiles, no border zero multiplied (dconv_t.hip)
  if (dd_dconv_tfwd_launch(x, packed, bias, y, d, epilogue, wp_bytes, st)) {
    DD_LAUNCH_CHECK("dconv_tfwd");
    return 0;
  }
  // up_conv_4's forward (16 -> 8, k7 d3): input-aligned, four tap columns packed into one column tile (dconv_t.hip)
  if (!mask && dd_dconv_tfwd8_launch(x, packed, bias, y, d, epilogue, st)) {
    DD_LAUNCH_CHECK("dconv_tfwd8");
    return 0;
  }
  // the data gradient of the 96->64 / 64->32 layers: one output row per workgroup, unrolled tap columns (dconv_t.hip)
  if (dd_dconv_gfwd_launch(x, packed, bias, mask, y, d, epilogue, wp_bytes, st)) {
    DD_LAUNCH_CHECK("dconv_gfwd");
    return 0;
  }
  const DcPlan plan = dc_plan(*d, d->dil_h);
  // one 8-wave workgroup per CU, all resident -- TWO for the stride-1 3x3 layers on at most one column tile (out_conv, rm_conv_2, the decoder's
  // dc2: 58 KB of LDS and <= 118 registers fit twice; their short tap loops (9 taps) leave a single workgroup waiting on its fills)
  const int per_cu = (d->kh == 3 && nt <= 1) ? 2 : 1;
  const int grid = (int)max(1, min(dd_cu_budget_internal() * per_cu, plan.total));
#define DD_DC(KK, DD_, NTT) hipLaunchKernelGGL((dconv_fwd_kernel<KK, DD_, NTT>), dim3(grid), dim3(DC_THREADS), 0, st, x, packed, bias, mask, y, *d, epilogue, wp_bytes, dbg_repeat)
#define DD_DC_NT(KK, DD_) do { if (nt == 0) DD_DC(KK, DD_, 0); else if (nt == 1) DD_DC(KK, DD_, 1); else if (nt == 2) DD_DC(KK, DD_, 2); else DD_DC(KK, DD_, 3); } while (0)
  switch (dc_variant(d)) {
    case 0: DD_DC_NT(7, 7); break;
    case 1: DD_DC_NT(7, 3); break;
    case 2: DD_DC_NT(8, 8); break;
    case 3: DD_DC_NT(6, 6); break;
    case 4: DD_DC_NT(3, 1); break;
    default: DD_DC_NT(3, 3); break;
  }
#undef DD_DC_NT
#undef DD_DC
  DD_LAUNCH_CHECK("dconv_fwd");
  return 0;
}

// The same launch that ALSO leaves the per-channel sums of what it wrote (after the epilogue) in colsum[0 .. cout): the data gradient
// of up_conv_(k+1) is dL/dy of up_conv_k, and its channel sums are up_conv_k's bias gradient -- without this a separate pass re-reads
// the whole tensor.  Only the windowed multi-row kernel (csrc/dconv_m.hip) does it: ask dd_dconv_colsum_supported first.
int32_t dd_dconv_colsum_supported(const dd_gconv_desc* d, int32_t epilogue, int32_t has_mask) {
  return d && dd_dconv_mwin_takes(d, epilogue, has_mask != 0, false) ? 1 : 0;
}

int64_t dd_dconv_colsum_workspace_bytes(void) { return (int64_t)DD_NUM_CU * 3 * 64 * 4; }

int dd_dconv_fwd_colsum(const float* x, const float* packed, const float* mask, float* y, float* colsum, const dd_gconv_desc* d,
                        int32_t epilogue, void* workspace, int64_t workspace_bytes, void* stream) {
  DD_REQUIRE(dc_supported(d), DD_ERR_UNSUPPORTED, "dconv_fwd_colsum: unsupported layer");
  DD_REQUIRE(x && packed && y && colsum && workspace, DD_ERR_BAD_ARG, "dconv_fwd_colsum: NULL pointer");
  DD_REQUIRE(epilogue == DD_EPI_NONE || epilogue == DD_EPI_RELU_MASK, DD_ERR_BAD_ARG, "dconv_fwd_colsum: epilogue %d", epilogue);
  DD_REQUIRE(workspace_bytes >= dd_dconv_colsum_workspace_bytes(), DD_ERR_WORKSPACE, "dconv_fwd_colsum: workspace too small");
  DD_REQUIRE(dd_dconv_mwin_takes(d, epilogue, mask != nullptr, false), DD_ERR_UNSUPPORTED,
             "dconv_fwd_colsum: not a layer of the windowed multi-row kernel (dd_dconv_colsum_supported)");
  hipStream_t st = (hipStream_t)stream;
  const int wp_bytes = (int)(dd_dconv_packed_floats(d) * 4);
  const int grid = dd_cu_budget_internal();
  DD_REQUIRE(dd_dconv_mfwd_launch_colsum(x, packed, nullptr, mask, y, d, epilogue, wp_bytes, st, (float*)workspace), DD_ERR_UNSUPPORTED,
             "dconv_fwd_colsum: the launcher refused the layer");
  DD_LAUNCH_CHECK("dconv_fwd_colsum");
  dd_dconv_colsum_reduce_launch((const float*)workspace, colsum, grid, d->cout, st);
  DD_LAUNCH_CHECK("dconv_colsum_reduce");
  return 0;
}

}  // extern "C"

// =====================================================================================================================
// Weight gradient of the same layers (ConvTranspose2d, stride 1, dilation D, no padding):
//     dW[c][o][ky][kx] = sum over images and input pixels (iy, ix) of  x[iy][ix][c] * g[iy + D*ky][ix + D*kx][o]
// GEMM per tap: M = Cin, N = Cout, K = pixels (two per v_mfma_f32_32x32x2_f32: lane (h, m) supplies x[pixel h][c = m], lane
// (h, n) g[pixel h + tap shift][o = n]).  The generic engine (gconv_wgrad_kernel) loads both operands of every MFMA from
// L1/L2 (0.6 .. 1.3 loads per MFMA).  Here a workgroup owns ONE tap row ky and a range of input rows; per step an XT-pixel
// piece of an x row and the matching (XT + D(K-1))-pixel piece of the g row iy + D*ky sit in LDS (double buffered,
// contiguous 16-byte global loads), every tap column kx is an address offset, and the 8 waves split the tap row's
// (kx, Cin tile, Cout tile) accumulator tiles into NSETS sets and the piece's pixel pairs into 8 / NSETS parts: at most
// 8 tiles = 128 accumulator registers per wave, two ds_read_b32 per MFMA, no global operand loads in the loop.
// Partials per workgroup, then a fixed-order fp64 second stage (deterministic).

// NSETS_ = 0, "balanced": a tile count that 8 does not divide (96->64: 42) -- every wave owns OWN = PER_KY / 8 tiles over all
// pixel pairs of the piece, the NSH = PER_KY % 8 tiles left over are SHARED: each wave multiplies them for its eighth of the
// pixel pairs (its own partial sums, added by the second stage like those of the pixel parts).  42 tiles = 8 x 5.25: with sets
// of 6 one wave in eight multiplied nothing useful.
template <int K, int D, int C, int O, int XT_, int NSETS_>
struct DwGeom {
  static constexpr int HALO = D * (K - 1);
  static constexpr bool BAL = NSETS_ == 0;
  static constexpr int XT = XT_, NSETS = BAL ? 8 : NSETS_, PARTS = 8 / NSETS;
  static constexpr int MT = (C + 31) / 32, NTO = (O + 31) / 32;
  static constexpr int PER_KY = K * MT * NTO;                       // accumulator tiles per tap row
  static constexpr int OWN = BAL ? PER_KY / 8 : (PER_KY + NSETS - 1) / NSETS;      // tiles a wave owns
  static constexpr int NSH = BAL ? PER_KY % 8 : 0;                  // shared tiles
  static constexpr int TPW = OWN + NSH;                             // accumulator tiles per wave
  static constexpr int UNITS = XT / 2 / PARTS / 8;                  // 8-pixel-pair units per wave and step
  static constexpr int XB = XT * C;                                 // floats of the x piece
  static constexpr int GB = (XT + HALO) * O;                        // floats of the g piece
  static constexpr int BUF = XB + GB + 64;                          // + slack: lanes past Cout / Cin read (and ignore) the next pixel
  static constexpr int NPIECE = (BUF / 4 + DC_THREADS - 1) / DC_THREADS;
  static_assert(XT % (16 * PARTS) == 0 && UNITS >= 1 && TPW <= 8, "piece / tile split");
};

template <int K, int D, int C, int O, int XT_, int NSETS_>
__global__ __launch_bounds__(DC_THREADS) void dconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 float* __restrict__ part, int wg_per_ky, int B, int H, int W,
                                                                 int x_cstore, int x_coff, int gh, int gw, int g_cstore, int g_coff) {
  using G = DwGeom<K, D, C, O, XT_, NSETS_>;
  __shared__ __attribute__((aligned(16))) float lds[2][G::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, n = lane & 31;
  const int ky = blockIdx.x / wg_per_ky, wl = blockIdx.x - ky * wg_per_ky;
  const int set = wave % G::NSETS, prt = wave / G::NSETS;
  // rows of this workgroup: an even share of the B*H input rows
  const long rows = (long)B * H;
  const long r0 = rows * wl / wg_per_ky, r1 = rows * (wl + 1) / wg_per_ky;
  const int nxt = (W + G::XT - 1) / G::XT;
  const long nsteps = (r1 - r0) * nxt;

  // LDS byte offsets of this wave's tiles: A = x piece (+ Cin tile), B = g piece (+ tap column, Cout tile)
  // The piece's pixel pairs are dealt round-robin to the PARTS pixel parts: pair j = pp * PARTS + prt, pixels 2j, 2j + 1.  A row's
  // last piece is only partly inside the image (298 = 2 x 128 + 42): dealt this way every part holds an equal share of the VALID
  // pairs and the pair loop skips the rest (contiguous parts left the first ones full and the last ones empty: the step took a
  // full piece's time whatever it held).
  const int px0 = 2 * prt;
  int aoff[G::TPW], boff[G::TPW];
#pragma unroll
  for (int i = 0; i < G::TPW; ++i) {
    const int t = i < G::OWN ? set * G::OWN + i : 8 * G::OWN + (i - G::OWN);      // (balanced: set = wave)
    if (t < G::PER_KY) {
      const int nt = t % G::NTO, mt = (t / G::NTO) % G::MT, kx = t / (G::NTO * G::MT);
      aoff[i] = ((px0 + h) * C + mt * 32 + n) * 4;
      boff[i] = (G::XB + (px0 + h + kx * D) * O + nt * 32 + n) * 4;
    } else {      // no such tile: multiply something harmless (no branch in the loop), the reduce never reads this accumulator
      aoff[i] = n * 4;
      boff[i] = n * 4;
    }
  }
  f32x16 acc[G::TPW];
#pragma unroll
  for (int i = 0; i < G::TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // fill: piece p (16 bytes) of a step = x piece [0, XB/4), then the g piece.  The per-thread byte offsets inside the two row
  // pieces never change: they are computed once; a step only rebuilds two wave-uniform descriptors (row piece start, bytes left
  // in the row: the hardware range check zero-fills what lies past the row end) -- no vector instruction per load.
  int voff[G::NPIECE];
#pragma unroll
  for (int i = 0; i < G::NPIECE; ++i) {
    const int p = tid + DC_THREADS * i;
    if (p < G::XB / 4) {
      const int px = (4 * p) / C, c = 4 * p - px * C;
      voff[i] = (px * x_cstore + x_coff + c) * 4;
    } else {
      const int q = 4 * p - G::XB, pg = q / O, cg = q - pg * O;
      voff[i] = q < G::GB ? (pg * g_cstore + g_coff + cg) * 4 : -16;
    }
  }
  auto issue = [&](long s, f32x4* st) {
    const long row = r0 + s / nxt;
    const int xt = (int)(s % nxt), b = (int)(row / H), iy = (int)(row - (long)b * H);
    const int x0 = xt * G::XT, gy = iy + D * ky;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (((long)b * H + iy) * W + x0) * x_cstore, (W - x0) * x_cstore * 4);
    const __amdgpu_buffer_rsrc_t gs = dd_rsrc(g + (((long)b * gh + gy) * gw + x0) * g_cstore, (gy < gh && x0 < gw) ? (gw - x0) * g_cstore * 4 : 0);
#pragma unroll
    for (int i = 0; i < G::NPIECE; ++i) {
      if (DC_THREADS * (i + 1) <= G::XB / 4) st[i] = dd_bload4(xs, voff[i]);                 // the whole round lies in the x piece
      else if (DC_THREADS * i >= G::XB / 4) st[i] = dd_bload4(gs, voff[i]);                  // ... in the g piece
      else {                                                                                 // the round straddles both
        const bool isx = tid + DC_THREADS * i < G::XB / 4;
        const f32x4 vx = dd_bload4(xs, isx ? voff[i] : -16), vg = dd_bload4(gs, isx ? -16 : voff[i]);
        st[i] = isx ? vx : vg;
      }
    }
  };
  auto retire = [&](int buf, const f32x4* st) {
#pragma unroll
    for (int i = 0; i < G::NPIECE; ++i) {
      const int p = tid + DC_THREADS * i;
      if (p < G::BUF / 4) *(f32x4*)&lds[buf][4 * p] = st[i];
    }
  };

  if (nsteps > 0) {
    f32x4 st[G::NPIECE];
    issue(0, st);
    retire(0, st);
  }
  __syncthreads();
  int par = 0;
  for (long s = 0; s < nsteps; ++s) {
    f32x4 st[G::NPIECE];
    const bool more = s + 1 < nsteps;
    if (more) issue(s + 1, st);
    {
      // operands of pixel pair pp + 1 are read while pair pp is multiplied (left to itself the compiler emits read, read, wait,
      // MFMA: every MFMA exposed to the LDS latency)
      const char* lb = (const char*)&lds[par][0];
      constexpr int NP = G::UNITS * 8;
      constexpr int PSTRIDE = 2 * G::PARTS;                                   // pixels between two pairs of one part
      // valid pairs of this wave's part in this step's piece (pairs past the row end would multiply zeros)
      const int valid_px = min(G::XT, W - (int)(s % nxt) * G::XT);
      const int npp = G::BAL ? NP : max(0, min(NP, ((valid_px + 1) / 2 - prt + G::PARTS - 1) / G::PARTS));
      float av[2][G::OWN], bw[2][G::OWN];
      // shared tiles: this wave's eighth of the pixel pairs, all operands requested up front
      constexpr int NSP = G::BAL ? NP / 8 : 0;
      static_assert(!G::BAL || NP % 8 == 0, "pixel pairs of a piece split over the 8 waves");
      float sa[NSP > 0 ? NSP : 1][G::NSH > 0 ? G::NSH : 1], sb[NSP > 0 ? NSP : 1][G::NSH > 0 ? G::NSH : 1];
      if constexpr (G::BAL) {
        const char* la = lb + wave * (NSP * 2 * C * 4);
        const char* lg = lb + wave * (NSP * 2 * O * 4);
#pragma unroll
        for (int j = 0; j < NSP; ++j)
#pragma unroll
          for (int k = 0; k < G::NSH; ++k) {
            sa[j][k] = *(const float*)(la + aoff[G::OWN + k] + j * 2 * C * 4);
            sb[j][k] = *(const float*)(lg + boff[G::OWN + k] + j * 2 * O * 4);
          }
      }
#pragma unroll
      for (int i = 0; i < G::OWN; ++i) {
        av[0][i] = *(const float*)(lb + aoff[i]);
        bw[0][i] = *(const float*)(lb + boff[i]);
      }
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        if (pp >= npp) continue;                                              // wave-uniform (a `break` would defeat the unrolling)
        if (pp + 1 < NP) {
#pragma unroll
          for (int i = 0; i < G::OWN; ++i) {
            av[(pp + 1) & 1][i] = *(const float*)(lb + aoff[i] + (pp + 1) * PSTRIDE * C * 4);
            bw[(pp + 1) & 1][i] = *(const float*)(lb + boff[i] + (pp + 1) * PSTRIDE * O * 4);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G::OWN; ++i) acc[i] = DD_MFMA(av[pp & 1][i], bw[pp & 1][i], acc[i]);
        if constexpr (G::BAL) {
          if (pp < NSP) {      // the shared tiles' MFMAs ride along with the first pairs (their operands landed long ago)
#pragma unroll
            for (int k = 0; k < G::NSH; ++k) acc[G::OWN + k] = DD_MFMA(sa[pp < NSP ? pp : 0][k], sb[pp < NSP ? pp : 0][k], acc[G::OWN + k]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) retire(par ^ 1, st);
    __syncthreads();
    par ^= 1;
  }
#pragma unroll
  for (int i = 0; i < G::TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[((((long)blockIdx.x * 8 + wave) * G::TPW + i) * 16 + r) * 64 + lane] = acc[i][r];
}

// One block = one accumulator tile of one tap row (1024 threads = 16 registers x 64 lanes): sums the tap row's workgroups and
// the pixel parts in fp64, in a fixed order, and scatters to the ConvTranspose2d weight layout [Cin][Cout][K][K].
template <int K, int D, int C, int O, int XT_, int NSETS_>
__global__ __launch_bounds__(1024) void dconv_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int wg_per_ky,
                                                           int accumulate) {
  using G = DwGeom<K, D, C, O, XT_, NSETS_>;
  const int ky = blockIdx.x / G::PER_KY, t = blockIdx.x - ky * G::PER_KY;
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  if (G::BAL && t >= 8 * G::OWN) {      // a shared tile: every wave of every workgroup holds a partial sum
    const int i = G::OWN + (t - 8 * G::OWN);
    // (workgroup, wave) pairs in order: wave index (ky * wg_per_ky + w) * 8 + wave runs through consecutive values
    dd_sum_strided(s, part + ((((long)ky * wg_per_ky * 8) * G::TPW + i) * 16 + r) * 64 + lane, (long)G::TPW * 1024, wg_per_ky * 8);
  } else {
    const int set = t / G::OWN, i = t - set * G::OWN;
    // (workgroup, pixel part) pairs in order: wave = set + NSETS * prt, and NSETS * PARTS == 8 makes w * 8 + NSETS * prt = NSETS * (w * PARTS + prt)
    static_assert(G::NSETS * G::PARTS == 8, "the pairs are one strided sequence");
    dd_sum_strided(s, part + ((((long)ky * wg_per_ky * 8 + set) * G::TPW + i) * 16 + r) * 64 + lane, (long)G::NSETS * G::TPW * 1024,
                   wg_per_ky * G::PARTS);
  }
  const int nt = t % G::NTO, mt = (t / G::NTO) % G::MT, kx = t / (G::NTO * G::MT);
  const int c = mt * 32 + dd_acc_row(r, lane), o = nt * 32 + (lane & 31);
  if (c < C && o < O) {
    const long wi = (((long)c * O + o) * K + ky) * K + kx;
    dw[wi] = accumulate ? dw[wi] + (float)s : (float)s;
  }
}

// ---- the 32->16 / 16->8 layers: 16x16 accumulator tiles on v_mfma_f32_16x16x4_f32 (K = 4 pixels per MFMA), ALL tap rows in one
// workgroup: wave w owns tap row ky = w (waves past K idle) and its K x ceil(Cin/16) x ceil(Cout/16) tiles (4 registers each); per
// step an XT-pixel piece of one x row and the K matching g pieces sit in LDS, so x and g are read from memory exactly once.
// Lane (q = l >> 4, m = l & 15) supplies x[pixel q][c = m] and g[pixel q + tap shift][o = m].
template <int K, int D, int C, int O, int XT_>
struct Dw16Geom {
  static constexpr int HALO = D * (K - 1), XT = XT_;
  static constexpr int MT = (C + 15) / 16, NTO = (O + 15) / 16;
  // Cout <= 8 would leave half of every 16-wide column tile empty: two tap columns share a tile -- column n < 8 is
  // (kx = 2p, o = n), column n >= 8 is (kx = 2p + 1, o = n - 8); the B operand is a per-lane LDS address either way
  static constexpr bool PAIR = O <= 8;
  static constexpr int KXT = PAIR ? (K + 1) / 2 : K;               // column-tile groups along kx
  static constexpr int TPW = KXT * MT * NTO;                        // tiles of one tap row = tiles per wave
  static constexpr int XB = XT * C, GB = (XT + HALO) * O;
  static constexpr int BUF = XB + K * GB + 64;
  static constexpr int NPIECE = (BUF / 4 + DC_THREADS - 1) / DC_THREADS;
  static constexpr int NQ = XT / 4;                                 // pixel quads per step
  static_assert(XT % 4 == 0 && K <= 8 && C % 4 == 0 && O % 4 == 0, "piece split");
};

template <int K, int D, int C, int O, int XT_>
__global__ __launch_bounds__(DC_THREADS) void dconv_wgrad16_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                   float* __restrict__ part, int B, int H, int W, int x_cstore,
                                                                   int x_coff, int gh, int gw, int g_cstore, int g_coff) {
  using G = Dw16Geom<K, D, C, O, XT_>;
  __shared__ __attribute__((aligned(16))) float lds[2][G::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q4 = lane >> 4, m = lane & 15;
  const bool active = wave < K;
  const int ky = active ? wave : 0;
  const long rows = (long)B * H;
  const long r0 = rows * blockIdx.x / gridDim.x, r1 = rows * (blockIdx.x + 1) / gridDim.x;
  const int nxt = (W + G::XT - 1) / G::XT;
  const long nsteps = (r1 - r0) * nxt;
  const int abase = (q4 * C + m) * 4;                                               // + mt*64 + quad*4*C*4
  const int bbase = G::PAIR ? (G::XB + ky * G::GB + (q4 + (m >> 3) * D) * O + (m & 7)) * 4      // + (2p*D*O)*4 + quad*4*O*4
                            : (G::XB + ky * G::GB + q4 * O + m) * 4;                          // + (kx*D*O + nt*16)*4 + quad*4*O*4

  f32x4 acc[G::TPW];
#pragma unroll
  for (int i = 0; i < G::TPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int voff[G::NPIECE];
#pragma unroll
  for (int i = 0; i < G::NPIECE; ++i) {
    const int p = tid + DC_THREADS * i;
    if (p < G::XB / 4) {
      const int px = (4 * p) / C, c = 4 * p - px * C;
      voff[i] = (px * x_cstore + x_coff + c) * 4;
    } else {
      const int e = 4 * p - G::XB, kyi = e / G::GB, r = e - kyi * G::GB, pg = r / O, cg = r - pg * O;
      voff[i] = kyi < K ? ((kyi * D * gw + pg) * g_cstore + g_coff + cg) * 4 : -16;
    }
  }
  auto issue = [&](long s, f32x4* st) {
    const long row = r0 + s / nxt;
    const int xt = (int)(s % nxt), b = (int)(row / H), iy = (int)(row - (long)b * H);
    const int x0 = xt * G::XT;
    // x: the rest of this row (pixels past the row end read zeros); g: from (iy, x0) to the end of the image -- tap row kyi is a
    // row offset inside voff.  g pixels past a row's end belong to the next row: they only ever meet x pixels that are zero.
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (((long)b * H + iy) * W + x0) * x_cstore, (W - x0) * x_cstore * 4);
    const __amdgpu_buffer_rsrc_t gs = dd_rsrc(g + (((long)b * gh + iy) * gw + x0) * g_cstore, (int)min((long)((gh - iy) * gw - x0) * g_cstore * 4, 0x7fffffffL));
#pragma unroll
    for (int i = 0; i < G::NPIECE; ++i) {
      if (DC_THREADS * (i + 1) <= G::XB / 4) st[i] = dd_bload4(xs, voff[i]);
      else if (DC_THREADS * i >= G::XB / 4) st[i] = dd_bload4(gs, voff[i]);
      else {
        const bool isx = tid + DC_THREADS * i < G::XB / 4;
        const f32x4 vx = dd_bload4(xs, isx ? voff[i] : -16), vg = dd_bload4(gs, isx ? -16 : voff[i]);
        st[i] = isx ? vx : vg;
      }
    }
  };
  auto retire = [&](int buf, const f32x4* st) {
#pragma unroll
    for (int i = 0; i < G::NPIECE; ++i) {
      const int p = tid + DC_THREADS * i;
      if (p < G::BUF / 4) *(f32x4*)&lds[buf][4 * p] = st[i];
    }
  };

  if (nsteps > 0) {
    f32x4 st[G::NPIECE];
    issue(0, st);
    retire(0, st);
  }
  __syncthreads();
  int par = 0;
  for (long s = 0; s < nsteps; ++s) {
    f32x4 st[G::NPIECE];
    const bool more = s + 1 < nsteps;
    if (more) issue(s + 1, st);
    if (active) {
      const char* lb = (const char*)&lds[par][0];
      float av[2][G::MT], bw[2][G::KXT * G::NTO];
      auto ld = [&](int slot, int quad) {
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) av[slot][mt] = *(const float*)(lb + abase + (quad * 4 * C + mt * 16) * 4);
#pragma unroll
        for (int kx = 0; kx < G::KXT; ++kx)
#pragma unroll
          for (int nt = 0; nt < G::NTO; ++nt)
            bw[slot][kx * G::NTO + nt] = *(const float*)(lb + bbase + ((quad * 4 + (G::PAIR ? 2 * kx : kx) * D) * O + nt * 16) * 4);
      };
      ld(0, 0);
#pragma unroll
      for (int quad = 0; quad < G::NQ; ++quad) {
        if (quad + 1 < G::NQ) ld((quad + 1) & 1, quad + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < G::KXT; ++kx)
#pragma unroll
          for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < G::NTO; ++nt)
              acc[(kx * G::MT + mt) * G::NTO + nt] = DD_MFMA16(av[quad & 1][mt], bw[quad & 1][kx * G::NTO + nt], acc[(kx * G::MT + mt) * G::NTO + nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) retire(par ^ 1, st);
    __syncthreads();
    par ^= 1;
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < G::TPW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[((((long)blockIdx.x * K + wave) * G::TPW + i) * 4 + r) * 64 + lane] = acc[i][r];
  }
}

// One block = one 16x16 tile of one tap row (256 threads = 4 registers x 64 lanes): fp64 sum over the workgroups, fixed order.
template <int K, int D, int C, int O, int XT_>
__global__ __launch_bounds__(256) void dconv_wgrad16_reduce(const float* __restrict__ part, float* __restrict__ dw, int nwg, int accumulate) {
  using G = Dw16Geom<K, D, C, O, XT_>;
  const int ky = blockIdx.x / G::TPW, t = blockIdx.x - ky * G::TPW;
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  dd_sum_strided(s, part + (((long)ky * G::TPW + t) * 4 + r) * 64 + lane, (long)K * G::TPW * 256, nwg);
  const int nt = t % G::NTO, mt = (t / G::NTO) % G::MT;
  const int kx = G::PAIR ? 2 * (t / (G::NTO * G::MT)) + ((lane & 15) >> 3) : t / (G::NTO * G::MT);
  const int c = mt * 16 + 4 * (lane >> 4) + r;                                  // D[row = 4*(lane >> 4) + r][col = lane & 15]
  const int o = G::PAIR ? (lane & 7) : nt * 16 + (lane & 15);
  if (c < C && o < O && kx < K) {
    const long wi = (((long)c * O + o) * K + ky) * K + kx;
    dw[wi] = accumulate ? dw[wi] + (float)s : (float)s;
  }
}

struct DwLayer { int k, d, c, o; };
const DwLayer kDwLayers[] = {{7, 7, 96, 64}, {7, 7, 64, 32}, {7, 7, 32, 16}, {7, 3, 16, 8}, {8, 8, 64, 32}, {8, 8, 32, 16}, {6, 6, 16, 8}};

int dw_variant(int k, int d, int c, int o) {
  for (int i = 0; i < (int)(sizeof(kDwLayers) / sizeof(kDwLayers[0])); ++i)
    if (kDwLayers[i].k == k && kDwLayers[i].d == d && kDwLayers[i].c == c && kDwLayers[i].o == o) return i;
  return -1;
}

// (K, D, Cin, Cout, piece width, tile sets): 42 tiles per tap row -> balanced (0: 5 own + 2 shared per wave), 14 -> 2 sets of 7; the 32->16 / 16->8 layers run
// on the 16-wide kernel (F16: piece width = a divisor of the layer's input width where there is one)
#define DD_DW_DISPATCH(V, F, F16)                 \
  switch (V) {                                    \
    case 0: F(7, 7, 96, 64, 64, 0); break;        \
    case 1: F(7, 7, 64, 32, 128, 2); break;       \
    case 2: F16(7, 7, 32, 16, 68); break;         \
    case 3: F16(7, 3, 16, 8, 128); break;         \
    case 4: F(8, 8, 64, 32, 128, 2); break;       \
    case 5: F16(8, 8, 32, 16, 52); break;         \
    default: F16(6, 6, 16, 8, 92); break;         \
  }

extern "C" {

/* x [B,H,W,x_cstore] (channels [x_coff, +cin)), g = dL/dy [B,gh,gw,g_cstore] (channels [g_coff, +cout)), gh >= H + D(K-1). */
int32_t dd_dconv_wgrad_supported(int32_t k, int32_t dil, int32_t cin, int32_t cout) {
  // Measured at bs 32 against the generic weight-gradient kernel: 96->64 11.39 -> 10.36 ms, 64->32 6.25 -> 5.66 ms on the 32-wide
  // kernel; the 32->16 / 16->8 layers would fill half / a quarter of its MFMA columns (4.2 / 4.5 ms there against 2.73 / 1.88 on
  // the generic kernel) and run on the 16-wide kernel instead.
  const int v = dw_variant(k, dil, cin, cout);
  return v >= 0 ? 1 : 0;
}

int64_t dd_dconv_wgrad_workspace_bytes(int32_t k, int32_t dil, int32_t cin, int32_t cout) {
  const int v = dw_variant(k, dil, cin, cout);
  if (v < 0) return -1;
  int64_t per_wg = 0;
#define DD_F(KK, DD_, CC, OO, XX, SS) per_wg = 8L * DwGeom<KK, DD_, CC, OO, XX, SS>::TPW * 1024 * 4
#define DD_F16(KK, DD_, CC, OO, XX) per_wg = (int64_t)KK * Dw16Geom<KK, DD_, CC, OO, XX>::TPW * 256 * 4
  DD_DW_DISPATCH(v, DD_F, DD_F16)
#undef DD_F
#undef DD_F16
  return (int64_t)DD_NUM_CU * per_wg;
}

int dd_dconv_wgrad(const float* x, const float* g, float* dw, int32_t batch, int32_t h, int32_t w, int32_t x_cstore, int32_t x_coff,
                   int32_t cin, int32_t gh, int32_t gw, int32_t g_cstore, int32_t g_coff, int32_t cout, int32_t k, int32_t dil,
                   int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  const int v = dw_variant(k, dil, cin, cout);
  DD_REQUIRE(v >= 0, DD_ERR_UNSUPPORTED, "dconv_wgrad: k%d d%d %d->%d is not one of the box heads' up-convs", k, dil, cin, cout);
  DD_REQUIRE(x && g && dw && workspace, DD_ERR_BAD_ARG, "dconv_wgrad: NULL pointer");
  DD_REQUIRE(batch > 0 && h > 0 && w > 0 && gh >= h + dil * (k - 1) && gw >= w + dil * (k - 1), DD_ERR_BAD_ARG,
             "dconv_wgrad: the gradient image %dx%d is smaller than %dx%d + %d", gh, gw, h, w, dil * (k - 1));
  DD_REQUIRE(x_cstore % 4 == 0 && x_coff % 4 == 0 && x_coff + cin <= x_cstore && g_cstore % 4 == 0 && g_coff % 4 == 0 && g_coff + cout <= g_cstore,
             DD_ERR_UNSUPPORTED, "dconv_wgrad: channel slices must be 4-aligned");
  DD_REQUIRE(((uintptr_t)x | (uintptr_t)g) % 16 == 0, DD_ERR_BAD_ARG, "dconv_wgrad: buffers must be 16-byte aligned");
  DD_REQUIRE((long)h * w * x_cstore * 4 < (1L << 31) && (long)gh * gw * g_cstore * 4 < (1L << 31), DD_ERR_UNSUPPORTED, "dconv_wgrad: one image exceeds 2 GB");
  DD_REQUIRE(workspace_bytes >= dd_dconv_wgrad_workspace_bytes(k, dil, cin, cout), DD_ERR_WORKSPACE, "dconv_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int wg_per_ky = max(1, dd_cu_budget_internal() / k);          // every tap row gets the same share of the workgroups
#define DD_F(KK, DD_, CC, OO, XX, SS)                                                                                                        \
  {                                                                                                                                          \
    hipLaunchKernelGGL((dconv_wgrad_kernel<KK, DD_, CC, OO, XX, SS>), dim3(wg_per_ky * KK), dim3(DC_THREADS), 0, st, x, g, (float*)workspace, \
                       wg_per_ky, batch, h, w, x_cstore, x_coff, gh, gw, g_cstore, g_coff);                                                  \
    hipLaunchKernelGGL((dconv_wgrad_reduce<KK, DD_, CC, OO, XX, SS>), dim3(KK * DwGeom<KK, DD_, CC, OO, XX, SS>::PER_KY), dim3(1024), 0, st, \
                       (const float*)workspace, dw, wg_per_ky, accumulate);                                                                  \
  }
#define DD_F16(KK, DD_, CC, OO, XX)                                                                                                         \
  {                                                                                                                                         \
    const int nwg = dd_cu_budget_internal();                                                                                                \
    hipLaunchKernelGGL((dconv_wgrad16_kernel<KK, DD_, CC, OO, XX>), dim3(nwg), dim3(DC_THREADS), 0, st, x, g, (float*)workspace, batch, h, w, \
                       x_cstore, x_coff, gh, gw, g_cstore, g_coff);                                                                         \
    hipLaunchKernelGGL((dconv_wgrad16_reduce<KK, DD_, CC, OO, XX>), dim3(KK * Dw16Geom<KK, DD_, CC, OO, XX>::TPW), dim3(256), 0, st,         \
                       (const float*)workspace, dw, nwg, accumulate);                                                                       \
  }
  DD_DW_DISPATCH(v, DD_F, DD_F16)
#undef DD_F
#undef DD_F16
  DD_LAUNCH_CHECK("dconv_wgrad");
  return 0;
}

}  // extern "C"
