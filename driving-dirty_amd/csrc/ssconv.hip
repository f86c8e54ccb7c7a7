// Data gradient of ss_conv -- Conv2d(32, 32, (1, 24), stride (1, 7)), spatial_bb/components.py:88,129 -- in one launch.
//
//     dx[b][y][7m + r][c] = sum over j, n of  g[b][y][m - j][n] * w[n][c][0][r + 7j]          (r = x mod 7; taps r + 7j < 24: 4 for r < 3, else 3)
//
// An input pixel only ever meets the taps of its own phase r, so the gather with a divisibility test multiplied 24 taps for the 3.4
// that contribute; round 2 cut it into 7 phase launches of the generic engine (exactly the useful work, 3.6 -> 0.5 ms at the time),
// but each of those is a reduction of 64 MFMAs per tile behind operand gathers from L1/L2: 0.13-0.15 ms per launch, 1.07 ms for the
// layer at bs 32 against 0.1 ms of HBM time for its 0.55 GB.  Here:
//
//   * a workgroup has SEVEN waves, wave = phase r: the phase's weights -- (4 taps x 32 x 32) as 16 B-fragments of the fp32 MFMA -- are
//     loaded ONCE into 64 registers and stay there for the whole launch;
//   * a task is 4 rows of one image: their g rows (4 x 128 pixels x 32 channels) sit in LDS at a pixel pitch of 144 B (16-byte reads
//     of 32 consecutive pixels touch every bank once) with zeroed halos (m - j < 0; m >= 128), so every tap is an address offset;
//   * the 4 x 132 values of m of a phase are numbered row-major and cut into 32-pixel m-tiles (a tile may straddle rows: the pixel
//     enters only through the lane's LDS address): 16.5 tiles instead of 4 x 5 with 4.1 used;
//   * per tile 3-4 taps x 4 chunks x 4 MFMAs, one ds_read_b128 per 4 MFMAs, then 16 stores of 128 contiguous bytes per pixel;
//   * two workgroups per CU run out of step: one's fill and stores under the other's MFMAs (no double buffering).
#include "dd_common.h"

namespace {

constexpr int SS_K = 24, SS_S = 7, SS_C = 32;
#ifndef SS_ROWS_N
#define SS_ROWS_N 4
#endif
constexpr int SS_ROWS = SS_ROWS_N;                       // rows of a task
constexpr int SS_GW = 128;                       // widest g row
constexpr int SS_HL = 3, SS_HR = 5;              // zeroed halo pixels left / right of a row image
constexpr int SS_RP = SS_HL + SS_GW + SS_HR;     // pixels of a row image
constexpr int SS_PP = 36;                        // floats per pixel (32 channels + 4 of padding: 144 B)
constexpr int SS_THREADS = 7 * 64;

__global__ __launch_bounds__(SS_THREADS) void ssconv_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int batch, int h, int gw, int xw) {
  __shared__ __attribute__((aligned(16))) float img[SS_ROWS * SS_RP * SS_PP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = phase
  const int hh = lane >> 5, n = lane & 31;
  const int ntaps = (SS_K - r + SS_S - 1) / SS_S;
  const int mp = (xw + SS_S - 1) / SS_S;                        // values of m per row (phase 0; the others lose at most the last one)
  const int ntiles = (SS_ROWS * mp + 31) >> 5;

  // ---- this phase's weights: B[k = g channel 8q + 4hh + i][col = n] = w[8q + 4hh + i][n][0][r + 7j]
  f32x4 Bf[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kx = r + SS_S * j, ch = 8 * q + 4 * hh + i;
        Bf[j][q][i] = kx < SS_K ? w[(ch * SS_C + n) * SS_K + kx] : 0.f;
      }
  // ---- halos: zero once (the fills never touch them)
  for (int p = tid; p < SS_ROWS * (SS_HL + SS_HR) * (SS_PP / 4); p += SS_THREADS) {
    const int c4 = p % (SS_PP / 4), hp = (p / (SS_PP / 4)) % (SS_HL + SS_HR), row = p / ((SS_PP / 4) * (SS_HL + SS_HR));
    const int px = hp < SS_HL ? hp : SS_GW + hp;                // SS_HL + SS_GW + (hp - SS_HL)
    *(f32x4*)&img[(row * SS_RP + px) * SS_PP + 4 * c4] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // pixels gw .. SS_GW - 1 of a narrower g row are zeros as well
  for (int p = tid; p < SS_ROWS * (SS_GW - gw) * 8; p += SS_THREADS) {
    const int c4 = p & 7, px = gw + (p >> 3) % (SS_GW - gw), row = (p >> 3) / (SS_GW - gw);
    *(f32x4*)&img[(row * SS_RP + SS_HL + px) * SS_PP + 4 * c4] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int groups = (h + SS_ROWS - 1) / SS_ROWS;
  const long ntasks = (long)batch * groups;
  const int out_bytes = h * xw * SS_C * 4;                      // one image of dx (the launcher checked < 2 GB)
  for (long t = blockIdx.x; t < ntasks; t += gridDim.x) {
    const int b = (int)(t / groups), y0 = (int)(t - (long)b * groups) * SS_ROWS;
    const int rows = min(SS_ROWS, h - y0);
    __syncthreads();                                            // the previous task's reads are done
    // ---- fill: rows y0 .. y0 + rows - 1 of g, 8 sixteen-byte pieces per pixel (rows past the image: zeros)
    const float* gb = g + ((long)b * h + y0) * gw * SS_C;
    for (int p = tid; p < SS_ROWS * gw * 8; p += SS_THREADS) {
      const int c4 = p & 7, px = (p >> 3) % gw, row = (p >> 3) / gw;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < rows) v = *(const f32x4*)&gb[((long)row * gw + px) * SS_C + 4 * c4];
      *(f32x4*)&img[(row * SS_RP + SS_HL + px) * SS_PP + 4 * c4] = v;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t ys = dd_rsrc(dx + ((long)b * h) * xw * SS_C, out_bytes);
    for (int tile = 0; tile < ntiles; ++tile) {
      const int j = 32 * tile + n;                              // this lane's (row, m)
      const int jr = j / mp, m = j - jr * mp;
      const char* ap = (const char*)&img[(min(jr, SS_ROWS - 1) * SS_RP + SS_HL + m) * SS_PP + 4 * hh];      // past the task: discarded
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        if (jt < ntaps) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 A = *(const f32x4*)(ap - jt * (SS_PP * 4) + q * 32);
            acc = DD_MFMA(A.x, Bf[jt][q].x, acc);
            acc = DD_MFMA(A.y, Bf[jt][q].y, acc);
            acc = DD_MFMA(A.z, Bf[jt][q].z, acc);
            acc = DD_MFMA(A.w, Bf[jt][q].w, acc);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int je = 32 * tile + dd_acc_row(e, lane);
        const int er = je / mp, em = je - er * mp;
        const int x = SS_S * em + r;
        const bool ok = er < rows && x < xw;
        dd_bstore1(ys, ok ? (((y0 + er) * xw + x) * SS_C + n) * 4 : -16, acc[e]);
      }
    }
  }
}


// =====================================================================================================================
// The FORWARD of the same layer:  y[b][row][m][n] = relu(bias[n] + sum over kx < 24, c of x[b][row][7m + kx][c] * w[n][c][0][kx]).
// On the generic engine every lane gathers its A operand from L1/L2 (16 bytes at a 896-byte stride, each input pixel fetched by the
// 3.4 outputs it feeds): 0.44 ms at bs 32 for 0.16 ms of MFMAs and 0.06 ms of HBM time.  Here a task is one 32-output m-tile of one
// row: its 248 input pixels are ONE contiguous 31 KB piece of x, staged in LDS at a 144-byte pixel pitch (lanes 7 pixels apart then
// touch every bank once), the 24 taps are split over the 8 waves (wave w: taps 3w .. 3w + 2, their 48 B-fragments resident in
// registers for the whole launch), and the eight partial tiles are summed through LDS in a fixed order (wave w finishes accumulator
// rows 2w, 2w + 1: bias, ReLU, two stores of 128 contiguous bytes per pixel).  Two workgroups per CU run out of step.
constexpr int SF_TW = 32;                                // outputs of a task
constexpr int SF_PX = SS_S * (SF_TW - 1) + SS_K;         // input pixels of a task: 241
constexpr int SF_IMG = SF_PX * SS_PP;                    // floats of the pixel image
constexpr int SF_THREADS = 512;

__global__ __launch_bounds__(SF_THREADS) void ssconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, long rows, int xw,
                                                                int gw, int relu) {
  __shared__ __attribute__((aligned(16))) float img[SF_IMG];
  __shared__ __attribute__((aligned(16))) float part[8 * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, n = lane & 31;
  // ---- this wave's taps: B[k = channel 8q + 4hh + i][col = n] = w[n][8q + 4hh + i][0][3 wave + t]
  f32x4 Bf[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) Bf[t][q][i] = w[(n * SS_C + 8 * q + 4 * hh + i) * SS_K + 3 * wave + t];
  const float bv0 = bias ? bias[n] : 0.f;
  const int mtiles = (gw + SF_TW - 1) / SF_TW;
  const long ntasks = rows * mtiles;
  const int x_row_bytes = xw * SS_C * 4;
  for (long t = blockIdx.x; t < ntasks; t += gridDim.x) {
    const long row = t / mtiles;
    const int m0 = (int)(t - row * mtiles) * SF_TW;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + row * xw * SS_C, x_row_bytes);      // pixels past the row: zeros (outputs past gw are not stored)
    __syncthreads();                                            // the previous task's reads of img / part are done
    for (int p = tid; p < SF_PX * 8; p += SF_THREADS) {
      const int c4 = p & 7, px = p >> 3;
      const f32x4 v = dd_bload4(xs, ((SS_S * m0 + px) * SS_C + 4 * c4) * 4);
      *(f32x4*)&img[px * SS_PP + 4 * c4] = v;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const char* ap = (const char*)&img[(SS_S * n + 3 * wave) * SS_PP + 4 * hh];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 A = *(const f32x4*)(ap + tt * (SS_PP * 4) + q * 32);
        acc = DD_MFMA(A.x, Bf[tt][q].x, acc);
        acc = DD_MFMA(A.y, Bf[tt][q].y, acc);
        acc = DD_MFMA(A.z, Bf[tt][q].z, acc);
        acc = DD_MFMA(A.w, Bf[tt][q].w, acc);
      }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[(wave * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + row * gw * SS_C, gw * SS_C * 4);
#pragma unroll
    for (int k = 0; k < 2; ++k) {                               // accumulator rows 2 wave, 2 wave + 1: the eight partials in wave order
      const int e = 2 * wave + k;
      float v = bv0;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) v += part[(ww * 16 + e) * 64 + lane];
      if (relu) v = fmaxf(v, 0.f);
      const int m = m0 + dd_acc_row(e, lane);
      dd_bstore1(ys, m < gw ? (m * SS_C + n) * 4 : -16, v);
    }
  }
}

}  // namespace

extern "C" {

int32_t dd_ssconv_dgrad_supported(int32_t h, int32_t gw, int32_t xw) {
  return h > 0 && gw > 0 && gw <= SS_GW && xw >= SS_K && (xw - SS_K) / SS_S + 1 == gw && (long)h * xw * SS_C * 4 < (1L << 31) ? 1 : 0;
}

int dd_ssconv_dgrad(const float* g, const float* w, float* dx, int32_t batch, int32_t h, int32_t gw, int32_t xw, void* stream) {
  DD_REQUIRE(g && w && dx && batch > 0, DD_ERR_BAD_ARG, "ssconv_dgrad: bad argument");
  DD_REQUIRE(dd_ssconv_dgrad_supported(h, gw, xw), DD_ERR_UNSUPPORTED,
             "ssconv_dgrad: g rows of at most %d pixels with gw = (xw - 24) / 7 + 1, an image of dx below 2 GB", SS_GW);
  DD_REQUIRE((((uintptr_t)g | (uintptr_t)dx) & 15) == 0, DD_ERR_BAD_ARG, "ssconv_dgrad: 16-byte aligned tensors");
  const long ntasks = (long)batch * ((h + SS_ROWS - 1) / SS_ROWS);
  const int grid = (int)min((long)2 * dd_cu_budget_internal(), ntasks);
  hipLaunchKernelGGL(ssconv_dgrad_kernel, dim3(grid), dim3(SS_THREADS), 0, (hipStream_t)stream, g, w, dx, batch, h, gw, xw);
  DD_LAUNCH_CHECK("ssconv_dgrad");
  return 0;
}

int dd_ssconv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t batch, int32_t h, int32_t xw, int32_t gw, int32_t relu,
                  void* stream) {
  DD_REQUIRE(x && w && y && batch > 0 && h > 0, DD_ERR_BAD_ARG, "ssconv_fwd: bad argument");
  DD_REQUIRE(xw >= SS_K && (xw - SS_K) / SS_S + 1 == gw && (long)xw * SS_C * 4 < (1L << 30), DD_ERR_UNSUPPORTED,
             "ssconv_fwd: gw = (xw - 24) / 7 + 1");
  DD_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, DD_ERR_BAD_ARG, "ssconv_fwd: 16-byte aligned tensors");
  const long rows = (long)batch * h, ntasks = rows * ((gw + SF_TW - 1) / SF_TW);
  const int grid = (int)min((long)2 * dd_cu_budget_internal(), ntasks);
  hipLaunchKernelGGL(ssconv_fwd_kernel, dim3(grid), dim3(SF_THREADS), 0, (hipStream_t)stream, x, w, bias, y, rows, xw, gw, relu);
  DD_LAUNCH_CHECK("ssconv_fwd");
  return 0;
}

}  // extern "C"
