// Dilated stride-1 convolutions of the box heads in gather form, SEVERAL OUTPUT ROWS PER WORKGROUP ("multi-row"):
//
//   RoadMapBoxesMergingCNN  up_conv_2 (64->32, k7 d7)   data gradient (a conv 32->64 over rows of 298 pixels)
//                           up_conv_3 (32->16, k7 d7)   data gradient (a conv 16->32 over rows of 340)        spatial_bb/components.py:136-137
//
// The one-row-per-workgroup kernels of dconv_t.hip give wave w the m-tile w of the row: 298 pixels are 9.3 tiles of 32 (7 % of the
// tenth multiplies nothing, and ten tiles do not deal evenly to the 8 waves of a workgroup), so these rows stayed on the 64-pixel wave
// tiles of dconv_fwd_kernel: 0.71-0.77 of the fp32 matrix peak.  In GATHER form the pixel a lane works on enters only through its LDS
// address -- tap (ky, kx) is a wave-uniform offset on top of it -- so an m-tile may straddle a row boundary for free.  Here a task is R
// consecutive output rows of one image, its R x out_w pixels are numbered row-major and cut into 32-pixel m-tiles (3 x 298 = 894 =
// 27.9 tiles -> 28: 0.2 % padding; 3 x 340 = 1020 -> 32 tiles: 0.4 %), and wave w owns tiles w, w + 8, ...: NTW accumulator tiles x NTC
// column tiles per wave (4 x 2 x 16 = 128 registers); 28 tiles are 7 per SIMD (waves w and w + 4 share one).
//
// One step = (tap row ky, 8-channel chunk q): the R input rows oy0 + jr - pad + D*ky (out_w + D(K-1) pixels x 32 B each) AND the step's
// K*NTC weight fragments (1 KB each, the images dd_dconv_pack writes) sit in LDS, double buffered, filled by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, nothing but the fill on the vector-memory counter) a whole step ahead -- across
// task boundaries too.  Per step a wave issues NTW x K x NTC x 4 MFMAs; a weight fragment is read once per NTW tiles, an A fragment
// once per 4*NTC MFMAs: one ds_read_b128 per 8 (NTC = 2) or 4 MFMAs, all of it LDS traffic (no global operand load in the loop).
// Tap rows are the OUTER loop: the chunks of a pixel share its 128-byte lines, and with the chunks outside every 32-byte piece cost
// its whole line again (13 GB fetched per launch for up_conv_2's data gradient; 10 GB and 4.44 -> 4.28 ms with the chunks inside).
//
// Rows are dealt to the workgroups as contiguous ranges of the (image, row) list -- 298 x 32 rows over 256 workgroups = 37.25 rows:
// twelve full tasks and a remainder of one or two rows, 2 % of imbalance instead of the 8 % of whole tasks.
//
// Measured (bs 32, one box, tools/ab_mfwd.py): up_conv_2 data gradient 4.71 -> 4.28 ms (121 -> 133 TF, 0.85 of the fp32 matrix
// peak), up_conv_3's 1.64 -> 1.51 ms (113 -> 123 TF).  What is left, by ablation builds (tools/build_variant.sh -DMF_ABL_*): the fills
// 0.22 ms, the mask loads of the write-out 0.11 ms.  This kernel fetches every input row once per tap row (10 GB per launch for
// up_conv_2); dconv_mwin_kernel below -- the one the launcher picks for the data gradients -- removes that (5.9 GB) at the same time.  Tried and dropped: 6-row tasks with 8 tiles per wave for up_conv_3 (1.58 against 1.52 ms); two 4-wave workgroups per CU
// on 2-row tasks, out of step with each other (4.75 against 4.61 ms at the time); requesting a tile's mask values a tile ahead of their
// use (no change once the write-out was a request / retire pair per tile).
//
// PADDED (the forward of a transposed layer: flipped taps, pad D(K-1)) is implemented -- tap rows that touch no input row of the task
// are skipped for the whole workgroup, tap columns that touch no input pixel of a tile for that tile (a 7-bit mask per tile slot),
// the zero border comes from the buffer range check of the fill -- and correct, but SLOWER than the input-aligned forward of
// dconv_t.hip (up_conv_2: 6.04 against 5.36 ms: a tile that straddles a row needs nearly every tap column): DD_DCONV_MFWD_PADDED=1
// turns it on for experiments, nothing dispatches to it by default.
#include <stdlib.h>

#include "dd_common.h"

namespace {


// LDS-DMA fills must have landed (vmcnt) before the barrier publishes the buffer; LDS reads of this step are done (lgkmcnt).
__device__ __forceinline__ void mf_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct MfTask {
  int b, oy, rows, ky0, ky1;
};

template <int K, int D, int NTC, int NTW, int R, int PXP, bool PADDED, int NW>
__global__ __launch_bounds__(NW * 64, 2) void dconv_mfwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, const float* __restrict__ msk,
                                                                float* __restrict__ y, const dd_gconv_desc d, int epi, int wp_bytes) {
  constexpr int T = K * K, HALO = D * (K - 1);
  constexpr int PATCH = PXP * 32;                         // bytes of the pixel image of a step (PXP pixels, a multiple of 32)
  constexpr int WB = K * NTC * 1024;                      // bytes of the step's weight fragments
  constexpr int STAGE = PATCH + WB;
  constexpr int NPI = PATCH / 1024;                       // wave-instructions that fill the pixel image
  constexpr int NPW = (NPI + NW - 1) / NW, NWW = (K * NTC + NW - 1) / NW;
  static_assert(PXP % 32 == 0 && 2 * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = d.cin >> 3;
  const int W = d.out_w, RW = W + HALO;                   // pixels of an image row in LDS: input columns [-pad_w, out_w - pad_w + HALO)
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  const __amdgpu_buffer_rsrc_t ws = dd_rsrc(wp, wp_bytes);

  // ---- fill plan of this thread: pixel-image instruction j = wave + 8i covers the 16-byte pieces 64j .. 64j + 63 (piece p = pixel
  // p >> 1, channel half p & 1; pixel = jr * RW + lam).  Offsets are relative to input row oy0 - pad_h + D*ky of the image.
  int poff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = 64 * (wave + NW * i) + lane, px = p >> 1;
    const int jr = px / RW, ix = px - jr * RW - d.pad_w;
    poff[i] = (jr < R && (unsigned)ix < (unsigned)d.in_w) ? ((jr * d.in_w + ix) * d.in_cstore + d.in_coff + 4 * (p & 1)) * 4 : (int)0xC0000000;
  }
  // ---- this wave's tile slots: tile t = wave + 8i covers pixels 32t .. 32t + 31 of the task's row-major pixel list
  int aoff[NTW];                                          // byte offset of this lane's A fragment (tap column 0) in the pixel image
  int kxmask[NTW];                                        // PADDED: tap columns that touch an input pixel of the tile (wave-uniform)
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int j = 32 * (wave + NW * i) + (lane & 31);
    const int jr = (j >= W) + (j >= 2 * W) + (R > 3 ? (j >= 3 * W) + (j >= 4 * W) + (j >= 5 * W) : 0);
    aoff[i] = ((jr * RW + (j - jr * W)) * 8 + 4 * (lane >> 5)) * 4;
    int m = 0;
    if (PADDED) {
      const int j0 = 32 * (wave + NW * i), j1 = j0 + 31;
      for (int r = 0; r < R; ++r) {                       // the tile's column range inside row r
        const int c0 = max(j0 - r * W, 0), c1 = min(j1 - r * W, W - 1);
        if (c0 > c1) continue;
        for (int kx = 0; kx < K; ++kx)
          if (c1 - d.pad_w + D * kx >= 0 && c0 - d.pad_w + D * kx < d.in_w) m |= 1 << kx;
      }
    } else {
      m = (1 << K) - 1;
    }
    kxmask[i] = __builtin_amdgcn_readfirstlane(m);
  }

  // ---- tasks: this workgroup's contiguous range of the (image, output row) list, R rows at a time, never across an image
  const long units = (long)d.batch * d.out_h;
  const long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  auto decode = [&](long u, MfTask& k) {
    k.b = (int)(u / d.out_h);
    k.oy = (int)(u - (long)k.b * d.out_h);
    k.rows = (int)min((long)R, min(u1 - u, (long)(d.out_h - k.oy)));
    k.ky0 = 0;
    k.ky1 = K - 1;
    if (PADDED) {      // tap rows with an input row for at least one of the task's rows: iy = oy + jr - pad_h + D*ky in [0, in_h)
      const int lo = d.pad_h - k.oy - (k.rows - 1), hi = d.in_h - 1 + d.pad_h - k.oy;
      k.ky0 = lo > 0 ? (lo + D - 1) / D : 0;
      k.ky1 = hi >= 0 ? min(K - 1, hi / D) : -1;
      if (k.ky1 < k.ky0) k.ky0 = k.ky1 = 0;               // no input row at all: one tap row of zeros (out-of-range fills)
    }
  };

  auto fill = [&](int buf, const MfTask& k, int q, int ky) {
    char* base = lds + buf * STAGE;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)k.b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const int rowoff = (k.oy - d.pad_h + D * ky) * d.in_w * d.in_cstore * 4;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int j = wave + NW * i;
      if (j < NPI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (__attribute__((address_space(3))) void*)(base + j * 1024), 16,
                                                 (int)((unsigned)poff[i] + (unsigned)rowoff), 32 * q, 0, 0);
    }
    const int wrow = (q * T + ky * K) * NTC * 1024;
#pragma unroll
    for (int i = 0; i < NWW; ++i) {
      const int j = ((wave + NW - (NPI % NW)) % NW) + NW * i;      // the waves with the fewest pixel-image instructions first
      if (j < K * NTC)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (__attribute__((address_space(3))) void*)(base + PATCH + j * 1024), 16, lane * 16,
                                                 wrow + j * 1024, 0, 0);
    }
  };

  if (u0 >= u1) return;
  MfTask cur, nxt;
  decode(u0, cur);
  fill(0, cur, 0, cur.ky0);
  mf_barrier();
  int par = 0;
  long u = u0;
  while (u < u1) {
    const long un = u + cur.rows;
    const bool have_next = un < u1;
    if (have_next) decode(un, nxt); else nxt = cur;
    const int npix = cur.rows * W;
    int nact = 0;                                          // active tile slots of this wave (a prefix: tiles ascend with the slot)
#pragma unroll
    for (int i = 0; i < NTW; ++i) nact += 32 * (wave + NW * i) < npix ? 1 : 0;

    f32x16 acc[NTW][NTC];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][nt][e] = 0.f;

    // Tap rows outside, channel chunks inside: the NC chunks of an input pixel share its 128-byte lines, and read in consecutive steps
    // the lines are still in the L2 / the memory-side cache for chunks 1 .. NC-1 (chunks outside: 13 GB fetched per launch for
    // up_conv_2's data gradient, 7 tap rows x 4 chunks x the 0.47 GB input -- every 32-byte piece cost its whole line again)
    for (int ky = cur.ky0; ky <= cur.ky1; ++ky) {
      for (int q = 0; q < NC; ++q) {
        // ---- the next step's images into the other buffer: next chunk, next tap row, or the next task's first step
        const bool lastk = ky == cur.ky1, lastq = q + 1 == NC;
#ifndef MF_ABL_NOFILL
        if (!lastq) fill(par ^ 1, cur, q + 1, ky);
        else if (!lastk) fill(par ^ 1, cur, 0, ky + 1);
        else if (have_next) fill(par ^ 1, nxt, 0, nxt.ky0);
#endif
        __builtin_amdgcn_sched_barrier(0);

        const char* pb = lds + par * STAGE;
        const char* wb = pb + PATCH + lane * 16;
        f32x4 Bc[NTC], Bn[NTC], Ac, An = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) Bn[nt] = Bc[nt] = *(const f32x4*)(wb + nt * 1024);
        Ac = *(const f32x4*)(pb + aoff[0]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          if (kx + 1 < K) {
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) Bn[nt] = *(const f32x4*)(wb + ((kx + 1) * NTC + nt) * 1024);
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (i + 1 < NTW) An = *(const f32x4*)(pb + aoff[i + 1] + kx * (D * 32));
            else if (kx + 1 < K) An = *(const f32x4*)(pb + aoff[0] + (kx + 1) * (D * 32));
            __builtin_amdgcn_sched_barrier(0);
            if (i < nact && (!PADDED || ((kxmask[i] >> kx) & 1))) {
#pragma unroll
              for (int nt = 0; nt < NTC; ++nt) {
                acc[i][nt] = DD_MFMA(Ac.x, Bc[nt].x, acc[i][nt]);
                acc[i][nt] = DD_MFMA(Ac.y, Bc[nt].y, acc[i][nt]);
                acc[i][nt] = DD_MFMA(Ac.z, Bc[nt].z, acc[i][nt]);
                acc[i][nt] = DD_MFMA(Ac.w, Bc[nt].w, acc[i][nt]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            Ac = An;
          }
#pragma unroll
          for (int nt = 0; nt < NTC; ++nt) Bc[nt] = Bn[nt];
        }
        mf_barrier();
        par ^= 1;
      }
    }

    // ---- write-out.  The mask values of a 32 x 32 tile are requested one tile AHEAD of their use (a ring of two), so that a tile's
    // stores go out while the next tile's mask is on its way: requested and used in the same tile, the eight tiles of a wave cost eight
    // serial memory round trips with the matrix pipe idle (the whole workgroup is in its epilogue at once).
    {
      const int fl = dd_fresh_lane();
      const int n = fl & 31;
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const __amdgpu_buffer_rsrc_t ms = dd_rsrc(msk ? msk + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore : y, msk ? out_bytes : 0);
#ifdef MF_ABL_NOMASK
      const bool masked = false;
#else
      const bool masked = epi == DD_EPI_RELU_MASK;
#endif
      const int base = ((cur.oy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff;
      // (two named register sets: an array indexed by `blk & 1` went to scratch memory)
      int offA[16], offB[16];
      float mvA[16], mvB[16];
      auto request = [&](int (&off)[16], float (&mv)[16], int blk) {      // offsets and mask values of block blk = (tile slot, column tile)
        const int i = blk / NTC, nt = blk % NTC;
        const int ch = nt * 32 + n;
        const bool pass = d.out_coff + ch >= d.mask_pass_lo && d.out_coff + ch < d.mask_pass_hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * (wave + NW * i) + dd_acc_row(r, fl);
          const int jr = (j >= W) + (j >= 2 * W) + (R > 3 ? (j >= 3 * W) + (j >= 4 * W) + (j >= 5 * W) : 0);
          const bool ok = i < nact && j < npix && ch < d.cout;
          off[r] = ok ? (base + (jr * d.omem_w + (j - jr * W)) * d.out_cstore + ch) * 4 : -16;
          mv[r] = 1.f;
        }
        if (masked) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float m = dd_bload1(ms, pass ? -16 : off[r]);
            mv[r] = pass ? 1.f : m;
          }
        }
      };
      auto retire = [&](const int (&off)[16], const float (&mv)[16], int blk) {
        const int i = blk / NTC, nt = blk % NTC;
        const int ch = nt * 32 + n;
        float bvn = 0.f;
        if ((epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) && ch < d.cout) bvn = bias[ch];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][nt][r] + bvn;
          if (epi == DD_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
          v = mv[r] > 0.f ? v : 0.f;
          dd_bstore1(ys, off[r], v);
        }
      };
      constexpr int NBLK = NTW * NTC;
      static_assert(NBLK % 2 == 0, "blocks are retired in pairs");
#ifdef MF_EPI_RING_OFF
#pragma unroll
      for (int blk = 0; blk < NBLK; ++blk) {
        request(offA, mvA, blk);
        retire(offA, mvA, blk);
      }
#else
      request(offA, mvA, 0);
#pragma unroll
      for (int blk = 0; blk < NBLK; blk += 2) {
        request(offB, mvB, blk + 1);
        __builtin_amdgcn_sched_barrier(0);
        retire(offA, mvA, blk);
        __builtin_amdgcn_sched_barrier(0);
        if (blk + 2 < NBLK) request(offA, mvA, blk + 2);
        __builtin_amdgcn_sched_barrier(0);
        retire(offB, mvB, blk + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    cur = nxt;
    u = un;
  }
}


// =====================================================================================================================
// The same tiles with PHASE ROWS and a SLIDING WINDOW (pad 0: the data gradients).  Above, a task's R rows are consecutive, so each
// of its K tap rows needs R rows of its own: every input row is fetched K times (10 GB per launch for up_conv_2's data gradient,
// 5.7 x its algorithmic bytes).  Rows of equal residue oy mod D share their input rows (taps are D rows apart): a task here is R
// consecutive rows of ONE residue class, oy = res + D(a0 + jr), and tap row ky of row jr reads input row res + D(a0 + jr + ky) --
// "sweep row" p = jr + ky.  Step ky multiplies the window p = ky .. ky + R - 1 and needs ONE new row for step ky + 1: K - 1 + R rows per
// sweep for R x K (row, tap row) pairs (9 instead of 21).  The rows live in a ring of 2R slots (slot = running row index mod 2R:
// R in use, one arriving, R - 1 of the next sweep -- the next chunk's, or the next task's -- arriving during the last R steps), the
// lane's A address is recomputed per step from its row jr (two conditional subtracts per tile), everything else is the kernel above.
struct MwTask {
  int b, res, a0, rows;
};

template <int K, int D, int NTC, int NTW, int R, int IWP, bool COLSUM>
__global__ __launch_bounds__(512, 2) void dconv_mwin_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, const float* __restrict__ msk,
                                                            float* __restrict__ y, const dd_gconv_desc d, int epi, int wp_bytes,
                                                            float* __restrict__ cpart) {
  constexpr int T = K * K;
  constexpr int NSLOT = 2 * R, SWEEP = K - 1 + R;
  constexpr int ROWB = IWP * 32;                          // bytes of a row image (IWP pixels x one 8-channel chunk)
  constexpr int WB = K * NTC * 1024;
  constexpr int NRI = ROWB / 1024, NRW = (NRI + 7) / 8, NWW = (K * NTC + 7) / 8;
  static_assert(IWP % 32 == 0 && NSLOT * ROWB + 2 * WB <= 160 * 1024, "LDS");
  static_assert(R == 3, "the row of a pixel index is found with two comparisons");
  __shared__ __attribute__((aligned(1024))) char lds[NSLOT * ROWB + 2 * WB];
  char* const ring = lds;
  char* const wbuf = lds + NSLOT * ROWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = d.cin >> 3;
  const int W = d.out_w;
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  const int row_bytes = d.in_w * d.in_cstore * 4;
  const __amdgpu_buffer_rsrc_t ws = dd_rsrc(wp, wp_bytes);

  // ---- fill plan: row-image instruction j = wave + 8i covers the 16-byte pieces 64j .. 64j + 63 (piece p = pixel p >> 1, channel half p & 1)
  int poff[NRW];
#pragma unroll
  for (int i = 0; i < NRW; ++i) {
    const int p = 64 * (wave + 8 * i) + lane, px = p >> 1;
    poff[i] = px < d.in_w ? (px * d.in_cstore + d.in_coff + 4 * (p & 1)) * 4 : (int)0xC0000000;
  }
  // ---- this wave's tile slots: tile t = wave + 8i covers pixels 32t .. 32t + 31 of the task's row-major pixel list
  int jrow[NTW], cb[NTW];                                 // this lane's row of the task and byte offset inside a row image (tap column 0)
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int j = 32 * (wave + 8 * i) + (lane & 31);
    jrow[i] = (j >= W) + (j >= 2 * W);
    cb[i] = ((j - jrow[i] * W) * 8 + 4 * (lane >> 5)) * 4;
  }

  // ---- tasks: this workgroup's contiguous range of the (image, residue class, phase row) list, R rows at a time inside a class
  const long units = (long)d.batch * d.out_h;
  const long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  auto decode = [&](long u, MwTask& k) {
    k.b = (int)(u / d.out_h);
    int rem = (int)(u - (long)k.b * d.out_h), res = 0, n = (d.out_h + D - 1) / D;
    while (rem >= n) {                                    // classes 0 .. D-1 hold ceil((out_h - res) / D) rows each
      rem -= n;
      ++res;
      n = (d.out_h - res + D - 1) / D;
    }
    k.res = res;
    k.a0 = rem;
    k.rows = (int)min((long)R, min(u1 - u, (long)(n - rem)));
  };
  auto fill_row = [&](int slot, const MwTask& k, int p, int q) {      // sweep row p of chunk q of task k into ring slot `slot`
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)k.b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    // (a sweep row past the image -- only rows the task does not have read it -- is out of the image's range: zeros, no fault)
    const int rowoff = (k.res + D * (k.a0 + p)) * row_bytes;
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
      const int j = wave + 8 * i;
      if (j < NRI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (__attribute__((address_space(3))) void*)(ring + slot * ROWB + j * 1024), 16,
                                                 (int)((unsigned)poff[i] + (unsigned)rowoff), 32 * q, 0, 0);
    }
  };
  auto fill_w = [&](int buf, int q, int ky) {
    const int wrow = (q * T + ky * K) * NTC * 1024;
#pragma unroll
    for (int i = 0; i < NWW; ++i) {
      const int j = ((wave + 8 - (NRI & 7)) & 7) + 8 * i;      // the waves with the fewest row-image instructions first
      if (j < K * NTC)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (__attribute__((address_space(3))) void*)(wbuf + buf * WB + j * 1024), 16, lane * 16,
                                                 wrow + j * 1024, 0, 0);
    }
  };
  auto slot_of = [&](int idx) { return idx % NSLOT; };

  // COLSUM: per-channel sums of everything this workgroup writes (after the mask) -- the bias gradient of the layer whose output
  // gradient this launch produces, which otherwise re-reads the whole tensor (channel_sum: 0.14 ms for up_conv_1's 727 MB)
  float bs[NTC];
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) bs[nt] = 0.f;
  if (u0 >= u1) {
    if (COLSUM && threadIdx.x < NTC * 64) cpart[(long)blockIdx.x * (NTC * 64) + threadIdx.x] = 0.f;
    return;
  }
  MwTask cur, nxt;
  decode(u0, cur);
  int sbase = 0;                                            // ring slot of sweep row 0 of the current sweep
#pragma unroll
  for (int p = 0; p < R; ++p) fill_row(p, cur, p, 0);
  fill_w(0, 0, 0);
  mf_barrier();
  int par = 0;
  long u = u0;
  while (u < u1) {
    const long un = u + cur.rows;
    const bool have_next = un < u1;
    if (have_next) decode(un, nxt); else nxt = cur;
    const int npix = cur.rows * W;
    int nact = 0;
#pragma unroll
    for (int i = 0; i < NTW; ++i) nact += 32 * (wave + 8 * i) < npix ? 1 : 0;

    f32x16 acc[NTW][NTC];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][nt][e] = 0.f;

    for (int q = 0; q < NC; ++q) {
      const bool lastq = q + 1 == NC;
      const bool next_sweep = !lastq || have_next;
      for (int ky = 0; ky < K; ++ky) {
        // ---- fills: the row step ky + 1 adds to the window; during the last R steps the first rows of the next sweep; the next weights
        if (ky + 1 < K) {
          fill_row(slot_of(sbase + ky + R), cur, ky + R, q);
          fill_w(par ^ 1, q, ky + 1);
        } else if (next_sweep) {
          fill_w(par ^ 1, lastq ? 0 : q + 1, 0);
        }
        if (ky >= K - R && next_sweep) {
          const int p2 = ky - (K - R);
          if (lastq) fill_row(slot_of(sbase + SWEEP + p2), nxt, p2, 0);
          else fill_row(slot_of(sbase + SWEEP + p2), cur, p2, q + 1);
        }
        __builtin_amdgcn_sched_barrier(0);

        int abase[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          int t = sbase + ky + jrow[i];                     // < 3 * NSLOT
          t -= t >= NSLOT ? NSLOT : 0;
          t -= t >= NSLOT ? NSLOT : 0;
          abase[i] = t * ROWB + cb[i];
        }
        const char* wb = wbuf + par * WB + lane * 16;
        f32x4 Bc[NTC], Bn[NTC], Ac, An = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) Bn[nt] = Bc[nt] = *(const f32x4*)(wb + nt * 1024);
        Ac = *(const f32x4*)(ring + abase[0]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          if (kx + 1 < K) {
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) Bn[nt] = *(const f32x4*)(wb + ((kx + 1) * NTC + nt) * 1024);
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (i + 1 < NTW) An = *(const f32x4*)(ring + abase[i + 1] + kx * (D * 32));
            else if (kx + 1 < K) An = *(const f32x4*)(ring + abase[0] + (kx + 1) * (D * 32));
            __builtin_amdgcn_sched_barrier(0);
            if (i < nact) {
#pragma unroll
              for (int nt = 0; nt < NTC; ++nt) {
                acc[i][nt] = DD_MFMA(Ac.x, Bc[nt].x, acc[i][nt]);
                acc[i][nt] = DD_MFMA(Ac.y, Bc[nt].y, acc[i][nt]);
                acc[i][nt] = DD_MFMA(Ac.z, Bc[nt].z, acc[i][nt]);
                acc[i][nt] = DD_MFMA(Ac.w, Bc[nt].w, acc[i][nt]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            Ac = An;
          }
#pragma unroll
          for (int nt = 0; nt < NTC; ++nt) Bc[nt] = Bn[nt];
        }
        mf_barrier();
        par ^= 1;
      }
      sbase = (sbase + SWEEP) % NSLOT;
    }

    // ---- write-out: a request / retire pair per tile (all offsets and mask values of a 32 x 32 tile before the first is used)
    {
      const int fl = dd_fresh_lane();
      const int n = fl & 31;
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const __amdgpu_buffer_rsrc_t ms = dd_rsrc(msk ? msk + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore : y, msk ? out_bytes : 0);
      const bool masked = epi == DD_EPI_RELU_MASK;
      const int row0 = cur.res + D * cur.a0 + d.ooff_h;
#pragma unroll
      for (int blk = 0; blk < NTW * NTC; ++blk) {
        const int i = blk / NTC, nt = blk % NTC;
        const int ch = nt * 32 + n;
        const bool pass = d.out_coff + ch >= d.mask_pass_lo && d.out_coff + ch < d.mask_pass_hi;
        int off[16];
        float mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * (wave + 8 * i) + dd_acc_row(r, fl);
          const int jr = (j >= W) + (j >= 2 * W);
          const bool ok = i < nact && j < npix && ch < d.cout;
          off[r] = ok ? (((row0 + D * jr) * d.omem_w + d.ooff_w + (j - jr * W)) * d.out_cstore + d.out_coff + ch) * 4 : -16;
          mv[r] = 1.f;
        }
        if (masked) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float m = dd_bload1(ms, pass ? -16 : off[r]);
            mv[r] = pass ? 1.f : m;
          }
        }
        float bvn = 0.f;
        if ((epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) && ch < d.cout) bvn = bias[ch];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][nt][r] + bvn;
          if (epi == DD_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
          v = mv[r] > 0.f ? v : 0.f;
          dd_bstore1(ys, off[r], v);
          if (COLSUM) {      // an opaque add: left to the compiler the 16 sums became a tree that kept 50 more registers alive (spills at NTC = 2)
            const float t = off[r] >= 0 ? v : 0.f;
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(bs[nt]) : "v"(t));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = nxt;
    u = un;
  }
  if (COLSUM) {      // the eight waves' sums in a fixed order (deterministic), one partial per workgroup; the second stage adds the halves
    float* red = (float*)lds;                               // the ring is idle: the last step's barrier is behind every wave
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) red[(wave * NTC + nt) * 64 + lane] = bs[nt];
    __syncthreads();
    if (threadIdx.x < NTC * 64) {
      float sum = 0.f;
      for (int w8 = 0; w8 < 8; ++w8) sum += red[w8 * NTC * 64 + threadIdx.x];
      cpart[(long)blockIdx.x * (NTC * 64) + threadIdx.x] = sum;
    }
  }
}

__global__ __launch_bounds__(64) void dconv_colsum_reduce(const float* __restrict__ cpart, float* __restrict__ out, int nblocks, int ntc, int cout) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cout) return;
  const int nt = c >> 5, n = c & 31;
  double sum = 0.0;
  const float* q0 = cpart + nt * 64 + n;
  for (int b0 = 0; b0 < nblocks; b0 += 16) {      // sixteen workgroups' pairs in flight; the sum runs in the order of the plain loop
    float lo[16], hi[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float* q = q0 + (long)min(b0 + j, nblocks - 1) * (ntc * 64);      // unconditional loads; past the end the last pair again, not added
      lo[j] = q[0];
      hi[j] = q[32];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (b0 + j < nblocks) sum += (double)lo[j] + (double)hi[j];
  }
  out[c] = (float)sum;
}

}  // namespace

// Whether the windowed kernel (the only one that can also sum its output per channel) takes this layer: the launcher's own conditions.
bool dd_dconv_mwin_takes(const dd_gconv_desc* d, int epilogue, bool has_mask, bool has_bias) {
  if (!dd_dconv_desc_ok(d)) return false;
  if (epilogue != DD_EPI_NONE && epilogue != DD_EPI_BIAS && epilogue != DD_EPI_BIAS_RELU && epilogue != DD_EPI_RELU_MASK) return false;
  if ((epilogue == DD_EPI_RELU_MASK && !has_mask) || ((epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU) && !has_bias)) return false;
  if (d->kh != 7 || d->kw != 7 || d->dil_h != 7 || d->dil_w != 7 || d->cout <= 16 || d->cout > 64 || d->cin % 8) return false;
  if ((long)d->in_h * d->in_w * d->in_cstore * 4 >= (1L << 30) || d->pad_h != 0 || d->pad_w != 0) return false;
  if (d->out_h > d->in_h - 42 || d->out_w > d->in_w - 42) return false;
  return d->in_w <= 384 && 3 * d->out_w <= 1024 && d->out_w + 42 <= 384 && d->out_h >= 7 && dd_cu_budget_internal() >= 1;
}

// Launches the multi-row gather kernel if the layer is one it is built for; false = nothing launched.
// ``colsum_part`` (NULL: none): one partial of (cout + 31) / 32 x 64 floats per workgroup of per-channel sums of the output; only the
// windowed kernel fills it -- false is returned, nothing launched, when the layer would go elsewhere.
bool dd_dconv_mfwd_launch_colsum(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                                 int epilogue, int wp_bytes, hipStream_t st, float* colsum_part) {
  // (the padded, transposed-forward form of this kernel measured slower than the input-aligned forward of dconv_t.hip -- up_conv_2: 6.04
  // against 5.36 ms, tiles that straddle a row need nearly every tap column -- and is not instantiated: gather form, pad 0 only)
  if (!dd_dconv_desc_ok(d)) return false;
  if (epilogue != DD_EPI_NONE && epilogue != DD_EPI_BIAS && epilogue != DD_EPI_BIAS_RELU && epilogue != DD_EPI_RELU_MASK) return false;
  if (epilogue == DD_EPI_RELU_MASK && !mask) return false;
  if ((epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU) && !bias) return false;
  if (d->kh != 7 || d->kw != 7 || d->dil_h != 7 || d->dil_w != 7) return false;
  if (d->cout <= 16 || d->cout > 64 || d->cin % 8) return false;
  if ((long)d->in_h * d->in_w * d->in_cstore * 4 >= (1L << 30)) return false;      // rejected fill offsets stay rejected with a row offset added
  if (d->pad_h != 0 || d->pad_w != 0) return false;
  const int halo = 42;
  if (d->out_h > d->in_h - halo || d->out_w > d->in_w - halo) return false;
  const int ntc = (d->cout + 31) / 32;
  const int rw = d->out_w + halo;
  const int grid = dd_cu_budget_internal();
  if (grid < 1) return false;
#define DD_MF(NTC_, NTW_, R_, PXP_, PAD_, NW_)                                                                                         \
  do {                                                                                                                               \
    if (R_ * rw > PXP_ || (R_ * d->out_w + 31) / 32 > NW_ * NTW_) return false;                                                      \
    hipLaunchKernelGGL((dconv_mfwd_kernel<7, 7, NTC_, NTW_, R_, PXP_, PAD_, NW_>), dim3(grid * (NW_ == 4 ? 2 : 1)), dim3(NW_ * 64),   \
                       0, st, x, packed, bias, mask, y, *d, epilogue, wp_bytes);                                                     \
    return true;                                                                                                                     \
  } while (0)
#define DD_MW(NTC_, IWP_)                                                                                                            \
  do {                                                                                                                               \
    if (d->in_w <= IWP_ && 3 * d->out_w <= 1024 && d->out_w + halo <= IWP_ && d->out_h >= 7) {                                        \
      if (colsum_part)                                                                                                               \
        hipLaunchKernelGGL((dconv_mwin_kernel<7, 7, NTC_, 4, 3, IWP_, true>), dim3(grid), dim3(512), 0, st, x, packed, bias, mask,    \
                           y, *d, epilogue, wp_bytes, colsum_part);                                                                  \
      else                                                                                                                           \
        hipLaunchKernelGGL((dconv_mwin_kernel<7, 7, NTC_, 4, 3, IWP_, false>), dim3(grid), dim3(512), 0, st, x, packed, bias, mask,   \
                           y, *d, epilogue, wp_bytes, (float*)nullptr);                                                              \
      return true;                                                                                                                   \
    }                                                                                                                                \
  } while (0)
  if (ntc == 2) DD_MW(2, 384);        // up_conv_2 data gradient: phase rows, one new input row per step
  if (ntc == 1) DD_MW(1, 384);        // up_conv_3 data gradient
#undef DD_MW
  if (colsum_part) return false;                           // only the windowed kernel sums its output
  // rows the windowed kernel does not take (wider than 384 pixels, fewer than 7 output rows): whole-row gather, no sliding window
  if (ntc == 2) DD_MF(2, 4, 3, 1024, false, 8);      // up_conv_2 data gradient: 3 x 340 = 1020 pixels
  if (ntc == 1) DD_MF(1, 4, 3, 1152, false, 8);      // up_conv_3 data gradient: 3 x 382 = 1146 (6-row tasks, 8 tiles per wave: 1.58 against 1.52 ms)
#undef DD_MF
  return false;
}

bool dd_dconv_mfwd_launch(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                          int epilogue, int wp_bytes, hipStream_t st) {
  return dd_dconv_mfwd_launch_colsum(x, packed, bias, mask, y, d, epilogue, wp_bytes, st, nullptr);
}

void dd_dconv_colsum_reduce_launch(const float* part, float* out, int nblocks, int cout, hipStream_t st) {
  hipLaunchKernelGGL(dconv_colsum_reduce, dim3((cout + 63) / 64), dim3(64), 0, st, part, out, nblocks, (cout + 31) / 32, cout);
}
