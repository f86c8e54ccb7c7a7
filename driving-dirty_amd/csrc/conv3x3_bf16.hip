// bf16 mixed-precision twin of conv3x3.hip (BASELINE config 5: 2x input resolution, bf16): the encoder's three 3x3
// convolutions (reference src/autoencoder/components.py:19-21,41-43) with bf16 operands on v_mfma_f32_32x32x16_bf16
// and fp32 accumulation.  Activations and activation gradients live in HBM as NHWC bf16 (64 bytes per 32-channel
// pixel, 8 bytes per stitched input pixel); weights, biases and their gradients stay fp32 (master copies), the
// operand images are re-rounded from them every step.
//
// Same "strip marching" decomposition as the fp32 kernels (one wave = 32 output pixels, private 3-slot LDS ring,
// register prefetch of the next row, no barrier in the loop, persistent equal ranges, range-checked buffer
// addressing) -- but at 16x the matrix rate these kernels are HBM-bound (read 64 B + write 64 B per pixel against
// 18 MFMAs of 32 cycles), so the design goal shifts from MFMA issue to bytes in flight:
//   * the matrix operand roles are swapped (A = weights, B = pixels): the accumulator then holds 16 CHANNELS of
//     one pixel per lane, which pack to 4 x 8-byte stores per lane instead of 16 two-byte ones;
//   * weights sit in registers (72 VGPRs), not LDS: the ring read (1 KB per MFMA) is then the only LDS traffic
//     and stays under the MFMA time; 2-3 workgroups share a CU so that a dozen rows per CU are in flight;
//   * the weight gradient needs both operands with PIXELS along k: it reads them out of row-major LDS images with
//     the hardware transpose read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group), where a tap's
//     pixel shift is just a row offset.
#include <stdlib.h>

#include "dd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

#define BF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {   // round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// Output streams carry the non-temporal hint (aux bit 1), as in conv3x3.hip; input streams carry it in the weight-gradient kernels only.
// A/B/A/B on one box, bs 16, 512 x 3672 (tools/ab_libs.sh over tools/bench_bf16.py): stores -- c1 forward 0.69 -> 0.59 ms, c2 forward
// 0.99 -> 0.87, c2 data gradient 0.89 -> 0.86; loads on top of that -- c3 / c1 / c2 weight gradient 0.43 -> 0.39, 0.44 -> 0.40, 0.79 -> 0.78,
// but c1 forward 0.45 -> 0.51 (its 8-byte pixels are re-read by the neighbouring strips), the others unchanged.  Not on the pool forward's
// stores: its 32-byte pieces want to be merged in the L2 first (0.21 -> 0.41 ms with the hint).
constexpr int BF_NT = 2;
template <int AUX = 0>
__device__ __forceinline__ u32x4 bload4(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX); }
template <int AUX = 0>
__device__ __forceinline__ u32x2 bload2(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, AUX); }
__device__ __forceinline__ unsigned bload1(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
constexpr int BF_ST = BF_NT;
__device__ __forceinline__ void bstore2(__amdgpu_buffer_rsrc_t r, int off, u32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, BF_ST); }
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, int off, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, BF_ST); }

// LDS geometry of one wave.  CIN == 32: 64-byte pixels in 16-byte chunks of 8 channels; CIN == 4 (3 real channels):
// 8-byte pixels, one "chunk" per pixel.
template <int CIN, int S>
struct BCfg {
  static constexpr int PXB = CIN * 2;
  static constexpr int CHB = (CIN == 32) ? 16 : 8;                 // bytes per load/store unit
  static constexpr int CHUNKS = PXB / CHB;
  static constexpr int NPX = (CIN == 32) ? 32 * S + 2 : 36;        // CIN 4: pixels n .. n+3 feed one k16 step
  static constexpr int SLOTB = NPX * PXB;
  static constexpr int NCH = NPX * CHUNKS;
  static constexpr int NLOAD = (NCH + 63) / 64;
  static constexpr int SPILLB = (NLOAD * 64 - NCH) * CHB;
  static constexpr int WAVEB = 3 * SLOTB + SPILLB;
  static constexpr int NW = (CIN == 32) ? 18 : 3;                  // operand registers of 8 bf16 per lane
};

template <int CIN>
struct RowRegs;
template <>
struct RowRegs<32> { typedef u32x4 T; };
template <>
struct RowRegs<4> { typedef u32x2 T; };

// chunk swizzle for the ds_read_b128 of 16 consecutive pixels (64-byte pitch): pixels p and p+4 would share banks
__device__ __forceinline__ int swz(int q) { return (q >> 2) & 3; }

template <int CIN, int S, int AUX = 0>
__device__ __forceinline__ void load_row(const unsigned short* __restrict__ img, int H, int W, int iy, int gx0, int lane,
                                         typename RowRegs<CIN>::T (&r)[BCfg<CIN, S>::NLOAD]) {
  using C = BCfg<CIN, S>;
  const bool rowok = (iy >= 0) && (iy < H);
  const __amdgpu_buffer_rsrc_t rs = rsrc(img + (long)(rowok ? iy : 0) * W * CIN, rowok ? W * C::PXB : 0);
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int c = lane + 64 * i;
    const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
    const int off = (c < C::NCH) ? ((gx0 + q) * C::PXB + ch * C::CHB) : -16;      // negative pixel -> huge offset -> zeros
    if constexpr (CIN == 32) r[i] = bload4<AUX>(rs, off);
    else r[i] = bload2<AUX>(rs, off);
  }
}

template <int CIN, int S, bool SWZ>
__device__ __forceinline__ void store_row(char* slot, char* spill, int lane, const typename RowRegs<CIN>::T (&r)[BCfg<CIN, S>::NLOAD]) {
  using C = BCfg<CIN, S>;
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int c = lane + 64 * i;
    const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
    char* dst = slot + q * C::PXB + ((CIN == 32 && SWZ) ? ((ch ^ swz(q)) << 4) : ch * C::CHB);
    if (64 * (i + 1) > C::NCH) dst = (c < C::NCH) ? dst : spill + (c - C::NCH) * C::CHB;
    *(typename RowRegs<CIN>::T*)dst = r[i];
  }
}

__device__ __forceinline__ void wave_range(long total, int gw, int nw, long& idx, long& end) {
  const long per = (total + nw - 1) / nw;
  idx = (long)gw * per;
  end = min(idx + per, total);
}

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // accumulator row -> channel

// Output staging.  The accumulator holds 16 channels of ONE pixel per lane; written straight to HBM that is 8-byte
// pieces scattered at a 64-byte pitch (4 partial requests per pixel -- measured: the store path, not HBM, then bounds
// the kernel).  Instead the tile goes through a per-wave LDS image (64-byte pixels, 16-byte chunks XOR-swizzled so
// that the four 8-byte writes of a wave are conflict-free) and leaves lane-linear: every store instruction writes
// 1 KB of contiguous memory, exactly like the loads.
__device__ __forceinline__ int stage_swz(int pix) { return (pix >> 1) & 3; }

__device__ __forceinline__ void stage_pixel16(char* tile, int pix, int h, const float (&v)[16]) {
  const int sw = stage_swz(pix);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x2 w;
    w.x = pack_bf16(v[4 * g], v[4 * g + 1]);
    w.y = pack_bf16(v[4 * g + 2], v[4 * g + 3]);
    *(u32x2*)(tile + pix * 64 + ((g ^ sw) << 4) + h * 8) = w;
  }
}

// NPIX staged pixels -> row bytes [base_off, base_off + 64*NPIX) of the descriptor (out-of-row pixels are dropped)
template <int NPIX>
__device__ __forceinline__ void flush_tile(const char* tile, __amdgpu_buffer_rsrc_t rs, int base_off, int lane) {
#pragma unroll
  for (int k = 0; k < NPIX / 16; ++k) {
    const int c = 64 * k + lane;
    const int pix = c >> 2, q = c & 3;
    const u32x4 v = *(const u32x4*)(tile + pix * 64 + ((q ^ stage_swz(pix)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, base_off + c * 16, 0, BF_ST);
  }
}

// ------------------------------------------------------------------------------------------------
// forward (EPI 0: y = bf16(relu(conv + bias)) + optional sign bits) and stride-1 data gradient
// (EPI 1: y = bf16(conv * bit(channel) of bits_in[pixel])).
//
// Memory pipeline.  vmcnt counts loads AND stores in issue order, and at 0.3 us of MFMA work per row there is no long
// compute phase for a store acknowledgement to hide behind: a wait that is even one operation too strict makes every
// row pay the full HBM write latency (measured: 2.4 us per row, 4x the roofline time).  So the loop is written for
// EXACT counted waits: two register slots (A, B) hold the rows in flight two iterations ahead, every iteration issues
// the same vector-memory sequence with no conditional operation in it (absent outputs and rows past the range go to a
// zero-size descriptor: the hardware drops them but still counts them), and the prologue issues the same sequence
// with dummy stores, so that the state at the loop header is identical from both predecessors.
// ------------------------------------------------------------------------------------------------
template <int CIN, int S>
struct InFlight {          // what one iteration prefetches: the new input rows of an output row + that row's sign word
  typename RowRegs<CIN>::T rows[S][BCfg<CIN, S>::NLOAD];
  unsigned mword;
};

template <int CIN, int S, int EPI, int WPB>
__global__ __launch_bounds__(WPB * 64) void bf_strip_fwd(const unsigned short* __restrict__ x, const bf16x8* __restrict__ wp,
                                                         const float* __restrict__ bias, const unsigned* __restrict__ bits_in,
                                                         unsigned short* __restrict__ y, unsigned* __restrict__ bits_out, int B,
                                                         int H, int W, int Ho, int Wo, int nstrips) {
  using C = BCfg<CIN, S>;
  using R = typename RowRegs<CIN>::T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int WAVEB = C::WAVEB + 2048;                  // ring + landing zone + output staging tile
  float* bias_lds = (float*)smem;                         // 32 floats (128 B) in front of the waves' regions
  char* ring = smem + 128 + wave * WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  char* tile = ring + C::WAVEB;
  const int h = lane >> 5, n = lane & 31;
  if (EPI == 0) {
    if (threadIdx.x < 32) bias_lds[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
  }

  bf16x8 wreg[C::NW];
#pragma unroll
  for (int i = 0; i < C::NW; ++i) wreg[i] = wp[i * 64 + lane];

  long idx, end;
  wave_range((long)B * nstrips * Ho, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / Ho;
    const int y0 = (int)(idx - col * Ho);
    const int y1 = (int)min((long)Ho, y0 + (end - idx));
    idx += y1 - y0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
#ifdef BF_ABL_NOREAD      // ablation builds (tools/build_variant.sh; wrong results): every image reads image 0 (cache-resident input)
    const unsigned short* xb = x;
#else
    const unsigned short* xb = x + (long)b * H * W * CIN;
#endif
    const int gx0 = S * x0 - 1;

    // group(t): the input rows output row t+1 adds to the window, and the sign word of output row t
    auto issue = [&](int t, InFlight<CIN, S>& f) {
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) load_row<CIN, S>(xb, H, W, S * t + 2 + s2, gx0, lane, f.rows[s2]);
      if (EPI == 1) {
        const bool ok = t < y1;
        const __amdgpu_buffer_rsrc_t ms = rsrc(bits_in + (long)(b * Ho + (ok ? t : 0)) * Wo, ok ? Wo * 4 : 0);
        f.mword = bload1(ms, (x0 + n) * 4);
      }
    };
    // one output row: compute from the ring, retire `cur` (rows for the next output row), write row t
    auto step = [&](int t, InFlight<CIN, S>& cur, InFlight<CIN, S>& nxt) {
      issue(t + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);      // loads stay above the MFMA chain

      f32x16 acc;
      if (EPI == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bq = *(const f32x4*)(bias_lds + 8 * g + 4 * h);
          acc[4 * g] = bq.x; acc[4 * g + 1] = bq.y; acc[4 * g + 2] = bq.z; acc[4 * g + 3] = bq.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      if constexpr (CIN == 32) {
        // operand reads run three taps ahead of the MFMAs that consume them (an LDS read is ~4 MFMAs of latency)
        const char* rowp[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rowp[dy] = ring + ((S * t + dy) % 3) * C::SLOTB;
        bf16x8 op[9][2];
        auto rd = [&](int tap) {
          const int q = S * n + tap % 3;
          const char* pa = rowp[tap / 3] + q * 64;
          const int sw = swz(q);
#pragma unroll
          for (int m = 0; m < 2; ++m) op[tap][m] = __builtin_bit_cast(bf16x8, *(const u32x4*)(pa + (((2 * m + h) ^ sw) << 4)));
        };
        rd(0); rd(1); rd(2);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 3 < 9) rd(tap + 3);
          __builtin_amdgcn_sched_barrier(0);
          acc = BF_MFMA(wreg[tap * 2], op[tap][0], acc);
          acc = BF_MFMA(wreg[tap * 2 + 1], op[tap][1], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {   // k16 step = the 4 channels of pixels n + 2h, n + 2h + 1 (dx = 2h, 2h+1; dx = 3 has zero weights)
        bf16x8 op[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const char* pa = ring + ((S * t + dy) % 3) * C::SLOTB + (n + 2 * h) * 8;
          u32x4 v;
          const u32x2 lo = *(const u32x2*)pa, hi = *(const u32x2*)(pa + 8);
          v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
          op[dy] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) acc = BF_MFMA(wreg[dy], op[dy], acc);
      }

      // retire the rows output row t+1 needs into the ring slots row t no longer uses
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        const int iy = S * t + 2 + s2;
        store_row<CIN, S, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, cur.rows[s2]);
      }

#ifdef BF_ABL_NOSTORE     // ablation: the output row goes to a zero-size descriptor (issued, counted, dropped)
      const bool live = false;
#else
      const bool live = t < y1;                 // the second half of the last pair may be a dummy row: stores dropped
#endif
      const long opix = (long)(b * Ho + (live ? t : 0)) * Wo;
      const __amdgpu_buffer_rsrc_t ys = rsrc(y + opix * 32, live ? Wo * 64 : 0);
      float v[16];
      unsigned mine = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (EPI == 0) {
          v[r] = fmaxf(acc[r], 0.f);
          mine |= (v[r] > 0.f ? 1u : 0u) << chan_of(r, h);
        } else {
          v[r] = ((cur.mword >> chan_of(r, h)) & 1u) ? acc[r] : 0.f;
        }
      }
      stage_pixel16(tile, n, h, v);
      flush_tile<32>(tile, ys, x0 * 64, lane);
      if (EPI == 0) {   // the two half-waves hold the two 16-channel halves of pixel n's word
        const unsigned word = mine | (unsigned)__shfl_xor((int)mine, 32);
#ifdef BF_ABL_NOBITS      // ablation: no sign words written
        const bool want = false;
#else
        const bool want = live && bits_out != nullptr;
#endif
        const __amdgpu_buffer_rsrc_t bs = rsrc(want ? bits_out + opix : (unsigned*)y, want ? Wo * 4 : 0);
        bstore1(bs, (h == 0) ? (x0 + n) * 4 : -16, word);
      }
      __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int d = 0; d < 3; ++d) {
      R t3[C::NLOAD];
      const int iy = S * y0 - 1 + d;
      load_row<CIN, S>(xb, H, W, iy, gx0, lane, t3);
      store_row<CIN, S, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t3);
    }
    InFlight<CIN, S> fa, fb;
    fa.mword = 0; fb.mword = 0;
    issue(y0, fa);
    asm volatile("" ::: "memory");            // keep the loads in front of the mirror stores at IR level too
    __builtin_amdgcn_sched_barrier(0);
    {   // mirror the loop body's stores so that the loop header sees the same queue from both predecessors
      const __amdgpu_buffer_rsrc_t none = rsrc(y, 0);
      const u32x4 z = {0u, 0u, 0u, 0u};
      __builtin_amdgcn_raw_buffer_store_b128(z, none, lane * 16, 0, BF_ST);          // distinct offsets: two stores, not one
      __builtin_amdgcn_raw_buffer_store_b128(z, none, 1024 + lane * 16, 0, BF_ST);
      if (EPI == 0) bstore1(none, lane * 4, 0u);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int t = y0;               // y0 < y1 always: a do-while keeps prologue and first iteration in one straight line
    do {
      step(t, fa, fb);
      step(t + 1, fb, fa);
      t += 2;
    } while (t < y1);
  }
}

// ------------------------------------------------------------------------------------------------
// stride-2 data gradient by output parity class (tap lists as in conv3x3.hip's conv_s2_dgrad).  The two column
// parities of an output row are staged into ONE 64-pixel image so the row leaves as contiguous 4 KB.
// ------------------------------------------------------------------------------------------------
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void bf_s2_dgrad(const unsigned short* __restrict__ dy, const bf16x8* __restrict__ wp,
                                                        const unsigned* __restrict__ bits_in, unsigned short* __restrict__ dx,
                                                        int B, int H, int W, int Ho, int Wo, int nstrips) {
  using C = BCfg<32, 1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int WAVEB = C::WAVEB + 4096;
  char* ring = smem + wave * WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  char* tile = ring + C::WAVEB;
  const int h = lane >> 5, n = lane & 31;
  const int nr = (H + 1) / 2;

  bf16x8 wreg[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) wreg[i] = wp[i * 64 + lane];

  long idx, end;
  wave_range((long)B * nstrips * nr, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / nr;
    const int r0 = (int)(idx - col * nr);
    const int r1 = (int)min((long)nr, r0 + (end - idx));
    idx += r1 - r0;
    const int b = (int)(col / nstrips), s0 = (int)(col % nstrips) * 32;
    const unsigned short* dyb = dy + (long)b * Ho * Wo * 32;

#pragma unroll
    for (int d = 0; d < 2; ++d) {
      u32x4 t[C::NLOAD];
      load_row<32, 1>(dyb, Ho, Wo, r0 + d, s0, lane, t);
      store_row<32, 1, true>(ring + ((r0 + d) % 3) * C::SLOTB, spill, lane, t);
    }

    for (int r = r0; r < r1; ++r) {
      u32x4 pre[C::NLOAD];
      load_row<32, 1>(dyb, Ho, Wo, r + 2, s0, lane, pre);
      // sign words of the four output pixels this lane finishes: rows 2r, 2r+1 x columns 2(s0+n), 2(s0+n)+1
      unsigned mw[2][2];
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int yi = 2 * r + py;
        const __amdgpu_buffer_rsrc_t ms = rsrc(bits_in + (long)(b * H + min(yi, H - 1)) * W, (yi < H) ? W * 4 : 0);
        mw[py][0] = bload1(ms, (2 * (s0 + n)) * 4);
        mw[py][1] = bload1(ms, (2 * (s0 + n) + 1) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);

      const char* row_r = ring + (r % 3) * C::SLOTB;
      const char* row_r1 = ring + ((r + 1) % 3) * C::SLOTB;
#define BF_TILE(PY, PX, NTAP, ...)                                                                     \
  {                                                                                                    \
    constexpr int taps[NTAP][3] = {__VA_ARGS__};                                                       \
    f32x16 acc;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[i] = 0.f;                                       \
    _Pragma("unroll") for (int t = 0; t < NTAP; ++t) {                                                 \
      const char* rowp = taps[t][0] ? row_r1 : row_r;                                                  \
      const int q = n + taps[t][1];                                                                    \
      const char* pa = rowp + q * 64;                                                                  \
      const int sw = swz(q);                                                                           \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                  \
        const bf16x8 px = __builtin_bit_cast(bf16x8, *(const u32x4*)(pa + (((2 * m + h) ^ sw) << 4))); \
        acc = BF_MFMA(wreg[taps[t][2] * 2 + m], px, acc);                                              \
      }                                                                                                \
    }                                                                                                  \
    float v[16];                                                                                       \
    _Pragma("unroll") for (int rr = 0; rr < 16; ++rr)                                                  \
      v[rr] = ((mw[PY][PX] >> chan_of(rr, h)) & 1u) ? acc[rr] : 0.f;                                   \
    stage_pixel16(tile, 2 * n + (PX), h, v);                                                           \
  }
#define BF_FLUSH(PY)                                                                                   \
  {                                                                                                    \
    const int yi = 2 * r + (PY);                                                                       \
    const __amdgpu_buffer_rsrc_t os = rsrc(dx + ((long)(b * H + min(yi, H - 1)) * W) * 32, (yi < H) ? W * 64 : 0); \
    flush_tile<64>(tile, os, 2 * s0 * 64, lane);                                                       \
  }
      BF_TILE(1, 1, 4, {1, 1, 0}, {1, 0, 2}, {0, 1, 6}, {0, 0, 8})
      store_row<32, 1, true>(ring + ((r + 2) % 3) * C::SLOTB, spill, lane, pre);
      BF_TILE(1, 0, 2, {1, 0, 1}, {0, 0, 7})
      BF_FLUSH(1)
      BF_TILE(0, 1, 2, {0, 1, 3}, {0, 0, 5})
      BF_TILE(0, 0, 1, {0, 0, 4})
      BF_FLUSH(0)
#undef BF_TILE
#undef BF_FLUSH
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight / bias gradient: D[ci][co] += sum over pixels of x[pixel + tap][ci] * dy[pixel][co], one accumulator per
// tap; both operands come out of row-major LDS images through the transpose read.
// ------------------------------------------------------------------------------------------------
// transpose read: per 16-lane group a block of 4 rows x 16 consecutive bf16; lane 4q+p of the group supplies the
// address of row q, elements 4p..4p+3; lane e receives element e of the four rows.
__device__ __forceinline__ s16x4 tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p); }
__device__ __forceinline__ bf16x8 join(s16x4 a, s16x4 b) {
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int CIN, int S>
struct WCfg {
  using C = BCfg<CIN, S>;
  static constexpr int DYB = 32 * 64;                   // one dy row of the strip
  static constexpr int WAVEB = C::WAVEB + DYB;
  static constexpr int NT = (CIN == 32) ? 9 : 2;        // accumulators
};

template <int CIN, int S, int WPB>
__global__ __launch_bounds__(WPB * 64) void bf_wgrad(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                     float* __restrict__ part, float* __restrict__ bpart, int B, int H,
                                                     int W, int Ho, int Wo, int nstrips) {
  using C = BCfg<CIN, S>;
  using WC = WCfg<CIN, S>;
  using R = typename RowRegs<CIN>::T;
  constexpr int NT = WC::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * WC::WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  char* dys = ring + C::WAVEB;
  const int gw = blockIdx.x * WPB + wave;
  // transpose-read coordinates of this lane
  const int grp = lane >> 4, e = lane & 15, q = e >> 2, p = e & 3;
  const int hk = grp >> 1;                              // k half (pixels 8hk .. 8hk+7 of a k16 step)
  const int c0 = 16 * (grp & 1);                        // channel half of the 32 rows / columns
  // dy image: 64-byte rows; the lane's address for pixel block P: (P + q) * 64 + (c0 + 4p) * 2
  const int dy_lane = q * 64 + (c0 + 4 * p) * 2 + 8 * hk * 64;
  // x image, CIN 32: pixel S*(P+q)+dx, same column arithmetic.  CIN 4: row q = pixel P+q, elements 4p.. = pixel P+q+p
  const int x_lane = (CIN == 32) ? (S * (q + 8 * hk)) * 64 + (c0 + 4 * p) * 2 : (q + p + 8 * hk) * 8;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  long idx, end;
  wave_range((long)B * nstrips * Ho, gw, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / Ho;
    const int y0 = (int)(idx - col * Ho);
    const int y1 = (int)min((long)Ho, y0 + (end - idx));
    idx += y1 - y0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const unsigned short* xb = x + (long)b * H * W * CIN;
    const unsigned short* dyb = dy + (long)b * Ho * Wo * 32;
    const int gx0 = S * x0 - 1;
    const int doff = x0 * 64 + lane * 16;                // the strip's dy row: 2 KB = 2 x (64 lanes x 16 B)

#pragma unroll
    for (int d = 0; d < 3; ++d) {
      R t[C::NLOAD];
      const int iy = S * y0 - 1 + d;
      load_row<CIN, S, BF_NT>(xb, H, W, iy, gx0, lane, t);
      store_row<CIN, S, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t);
    }
    u32x4 dcur[2];
    {
      const __amdgpu_buffer_rsrc_t ds = rsrc(dyb + (long)y0 * Wo * 32, Wo * 64);
      dcur[0] = bload4<BF_NT>(ds, doff);
      dcur[1] = bload4<BF_NT>(ds, doff + 1024);
    }

    for (int yy = y0; yy < y1; ++yy) {
      R pre[S][C::NLOAD];
#pragma unroll
      for (int s = 0; s < S; ++s) load_row<CIN, S, BF_NT>(xb, H, W, S * yy + 2 + s, gx0, lane, pre[s]);
      u32x4 dnext[2];
      {
        const bool ok = yy + 1 < Ho;
        const __amdgpu_buffer_rsrc_t ds = rsrc(dyb + (long)(ok ? yy + 1 : 0) * Wo * 32, ok ? Wo * 64 : 0);
        dnext[0] = bload4<BF_NT>(ds, doff);
        dnext[1] = bload4<BF_NT>(ds, doff + 1024);
      }
      __builtin_amdgcn_sched_barrier(0);

      // this row's dy -> LDS (plain rows), then the two B operands every tap shares
      *(u32x4*)(dys + lane * 16) = dcur[0];
      *(u32x4*)(dys + 1024 + lane * 16) = dcur[1];
      bf16x8 bm[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const s16x4 lo = tr_read(dys + dy_lane + (16 * m) * 64), hi = tr_read(dys + dy_lane + (16 * m + 4) * 64);
        bm[m] = join(lo, hi);
        const u32x4 raw = __builtin_bit_cast(u32x4, bm[m]);    // bias gradient: this lane's 8 pixels of channel lane&31
        bsum += (bf_lo(raw.x) + bf_hi(raw.x)) + (bf_lo(raw.y) + bf_hi(raw.y)) + (bf_lo(raw.z) + bf_hi(raw.z)) + (bf_lo(raw.w) + bf_hi(raw.w));
      }

      if constexpr (CIN == 32) {
        // 18 (tap, k16 half) steps of 2 transpose reads + 1 MFMA; one wave per SIMD, so the reads run four steps ahead
        const char* rowb[3];
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) rowb[dyy] = ring + ((S * yy + dyy) % 3) * C::SLOTB + x_lane;
        constexpr int AH = 4;
        s16x4 lo[AH + 1], hi[AH + 1];
        auto rd = [&](int it) {
          const int tap = it >> 1, m = it & 1;
          const char* pa = rowb[tap / 3] + (S * 16 * m + tap % 3) * 64;
          lo[it % (AH + 1)] = tr_read(pa);
          hi[it % (AH + 1)] = tr_read(pa + S * 4 * 64);
        };
#pragma unroll
        for (int it = 0; it < AH; ++it) rd(it);
#pragma unroll
        for (int it = 0; it < 18; ++it) {
          if (it + AH < 18) rd(it + AH);
          __builtin_amdgcn_sched_barrier(0);
          acc[it >> 1] = BF_MFMA(join(lo[it % (AH + 1)], hi[it % (AH + 1)]), bm[it & 1], acc[it >> 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // rows of D: i = 16*(ky select) + 4*shift + channel; accumulator 0 takes ky 0 (rows 0-15) and 1 (rows 16-31),
        // accumulator 1 takes ky 2 in rows 0-15 (rows 16-31 repeat it and are ignored)
        const char* rk0 = ring + ((yy + (grp & 1)) % 3) * C::SLOTB + x_lane;
        const char* rk2 = ring + ((yy + 2) % 3) * C::SLOTB + x_lane;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[0] = BF_MFMA(join(tr_read(rk0 + (16 * m) * 8), tr_read(rk0 + (16 * m + 4) * 8)), bm[m], acc[0]);
          acc[1] = BF_MFMA(join(tr_read(rk2 + (16 * m) * 8), tr_read(rk2 + (16 * m + 4) * 8)), bm[m], acc[1]);
        }
      }

#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int iy = S * yy + 2 + s;
        store_row<CIN, S, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, pre[s]);
      }
      dcur[0] = dnext[0];
      dcur[1] = dnext[1];
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(((long)gw * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
  bpart[(long)gw * 64 + lane] = bsum;
}

// Second stage (deterministic): per (accumulator, register) row of 64 lanes, sum the waves' partials in a fixed order
// and scatter to OIHW.  D[i][j]: lane = j = co, register r -> i = chan_of(r, lane>>5).
template <int CIN>
__global__ __launch_bounds__(1024) void bf_wgrad_reduce(const float* __restrict__ part, const float* __restrict__ bpart,
                                                        float* __restrict__ dw, float* __restrict__ db, int nw) {
  constexpr int NT = (CIN == 32) ? 9 : 2;
  constexpr int G = 16;
  __shared__ float red[G][64];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const float* src = (row < NT * 16) ? part + (long)row * 64 + l : bpart + l;
  const long stride = (row < NT * 16) ? (long)NT * 16 * 64 : 64;
  // four running sums over the waves w = g + kG + 4G j (k = 0..3), then the ragged rest onto the first: the order of the four-way loop
  // this replaces, with 64 loads in flight instead of 4 (8192 partials per element for c1: 128 dependent round trips, 37 us)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int m = (g + 3 * G < nw) ? (nw - 1 - 3 * G - g) / (4 * G) + 1 : 0;
  dd_sum_strided(s0, src + (long)g * stride, 4 * G * stride, m);
  dd_sum_strided(s1, src + (long)(g + G) * stride, 4 * G * stride, m);
  dd_sum_strided(s2, src + (long)(g + 2 * G) * stride, 4 * G * stride, m);
  dd_sum_strided(s3, src + (long)(g + 3 * G) * stride, 4 * G * stride, m);
  const int w = g + 4 * G * m;
  if (w < nw) dd_sum_strided(s0, src + (long)w * stride, G * stride, (nw - w + G - 1) / G);
  red[g][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0) return;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < G; ++i) s += red[i][l];
  if (row < NT * 16) {
    const int t = row >> 4, r = row & 15;
    const int i = chan_of(r, l >> 5), co = l & 31;
    if (CIN == 32) {
      dw[((long)co * 32 + i) * 9 + t] = s;
    } else {
      const int ky = (t == 0) ? (i >> 4) : 2, kx = (i >> 2) & 3, c = i & 3;
      if ((t == 0 || i < 16) && kx < 3 && c < 3) dw[((long)co * 3 + c) * 9 + ky * 3 + kx] = s;
    }
  } else {
    const float other = __shfl_xor(s, 32);
    if (l < 32) db[l] = s + other;
  }
}

// ------------------------------------------------------------------------------------------------
// operand images from the fp32 master weights (OIHW), rounded to bf16: image[(i*64 + lane)*8 + j]
//   CIN 32: i = tap*2 + m, element j <-> k = 16m + 8(lane>>5) + j, row = lane&31
//     kind 0 (forward):        row = co, k = ci:  W[co][ci][tap]
//     kind 1 (stride-1 dgrad): row = ci, k = co:  W[co][ci][8 - tap]
//     kind 2 (stride-2 dgrad): row = ci, k = co:  W[co][ci][tap]
//   CIN 4 (forward only): i = ky, element j <-> (kx = 2(lane>>5) + (j>>2), c = j&3), zero for kx == 3 or c == 3
// ------------------------------------------------------------------------------------------------
__global__ void bf_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int cin_real, int kind) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  float v;
  if (cin_real == 32) {
    if (idx >= 18 * 64 * 8) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, i = idx >> 9;
    const int m = i & 1, tap = i >> 1;
    const int row = lane & 31, k = 16 * m + 8 * (lane >> 5) + j;
    if (kind == 0) v = w[((long)row * 32 + k) * 9 + tap];
    else if (kind == 1) v = w[((long)k * 32 + row) * 9 + (8 - tap)];
    else v = w[((long)k * 32 + row) * 9 + tap];
  } else {
    if (idx >= 3 * 64 * 8) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, ky = idx >> 9;
    const int co = lane & 31, kx = 2 * (lane >> 5) + (j >> 2), c = j & 3;
    v = (kx < 3 && c < 3) ? w[((long)co * 3 + c) * 9 + ky * 3 + kx] : 0.f;
  }
  p[idx] = (unsigned short)(pack_bf16(v, 0.f) & 0xffffu);
}

// ------------------------------------------------------------------------------------------------
// layout / pooling companions
// ------------------------------------------------------------------------------------------------
__constant__ int kViewOrderBf[6] = {0, 1, 2, 5, 4, 3};

// [B,6,3,H,W] fp32 camera views -> wide NHWC4 bf16 (view order of wide_stitch_six_images, roadmap_bce_v2.py:53-64)
__global__ __launch_bounds__(256) void stitch6_bf16_kernel(const float* __restrict__ views, u32x2* __restrict__ wide4, int B, int H, int W) {
  const long npx = (long)B * H * 6 * W;
  const long plane = (long)H * W;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < npx; px += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(px % (6 * W));
    const int yy = (int)((px / (6 * W)) % H);
    const int b = (int)(px / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const float* src = views + (((long)b * 6 + kViewOrderBf[slot]) * 3) * plane + (long)yy * W + xx;
    u32x2 o;
    o.x = pack_bf16(src[0], src[plane]);
    o.y = pack_bf16(src[2 * plane], 0.f);
    wide4[px] = o;
  }
}

// the same from a table of per-sample base pointers (the collate's tuple, helper.py:22-23: no torch.stack of the views)
struct BfSamplePtrs {
  const float* p[64];
};
__global__ __launch_bounds__(256) void stitch6_bf16_ptrs_kernel(const BfSamplePtrs samples, u32x2* __restrict__ wide4, int B, int H, int W) {
  const long npx = (long)B * H * 6 * W;
  const long plane = (long)H * W;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < npx; px += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(px % (6 * W));
    const int yy = (int)((px / (6 * W)) % H);
    const int b = (int)(px / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const float* src = samples.p[b] + ((long)kViewOrderBf[slot] * 3) * plane + (long)yy * W + xx;
    u32x2 o;
    o.x = pack_bf16(src[0], src[plane]);
    o.y = pack_bf16(src[2 * plane], 0.f);
    wide4[px] = o;
  }
}

// the same from uint8 HWC frames (per-sample [6,H,W,3], a JPEG decoder's output): ToTensor's /255 (a true division, as
// torchvision does it) and the bf16 rounding of the quotient fused with the gather -- the values dd_stitch6_bf16 makes of
// frames.float() / 255
struct BfSamplePtrsU8 {
  const unsigned char* p[64];
};
__global__ __launch_bounds__(256) void stitch6_bf16_u8_ptrs_kernel(const BfSamplePtrsU8 samples, u32x2* __restrict__ wide4, int B, int H, int W) {
  const long npx = (long)B * H * 6 * W;
  for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < npx; px += (long)gridDim.x * blockDim.x) {
    const int xw = (int)(px % (6 * W));
    const int yy = (int)((px / (6 * W)) % H);
    const int b = (int)(px / ((long)6 * W * H));
    const int slot = xw / W, xx = xw - slot * W;
    const unsigned char* src = samples.p[b] + (((long)kViewOrderBf[slot] * H + yy) * W + xx) * 3;
    u32x2 o;
    o.x = pack_bf16((float)src[0] / 255.0f, (float)src[1] / 255.0f);
    o.y = pack_bf16((float)src[2] / 255.0f, 0.f);
    wide4[px] = o;
  }
}

// max_pool1d(4) over the NCHW-flattened feature (components.py:46-47) from an NHWC bf16 tensor, H*W % 4 == 0:
// thread = (4 consecutive flat pixels, 4 channels)
__global__ __launch_bounds__(256) void pool4_bf16_fwd(const u32x2* __restrict__ feat, float* __restrict__ pooled, int B, long HW, int C) {
  const int groups = C / 4;
  const long quads = HW / 4;
  const long total = (long)B * quads * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long qd = (i / groups) % quads;
    const long b = i / (groups * quads);
    const u32x2* src = feat + ((b * HW + 4 * qd) * groups + g);
    float m[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x2 v = __builtin_nontemporal_load(src + (long)k * groups);
      const float f[4] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y)};
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = (k == 0) ? f[c] : fmaxf(m[c], f[c]);
    }
    float* o = pooled + b * (quads * C) + qd;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[(long)(4 * g + c) * quads] = m[c];
  }
}

// backward of pool + the ReLU in front of it: the first maximum of a window takes the gradient (torch keeps the earliest
// index on ties) if it is positive; output in bf16
__global__ __launch_bounds__(256) void pool4_bf16_bwd(const float* __restrict__ dpooled, const u32x2* __restrict__ feat,
                                                      u32x2* __restrict__ dfeat, int B, long HW, int C) {
  const int groups = C / 4;
  const long quads = HW / 4;
  const long total = (long)B * quads * groups;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long qd = (i / groups) % quads;
    const long b = i / (groups * quads);
    const long base = (b * HW + 4 * qd) * groups + g;
    float f[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x2 v = __builtin_nontemporal_load(feat + base + (long)k * groups);
      f[k][0] = bf_lo(v.x); f[k][1] = bf_hi(v.x); f[k][2] = bf_lo(v.y); f[k][3] = bf_hi(v.y);
    }
    const float* gp = dpooled + b * (quads * C) + qd;
    float d[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gk = gp[(long)(4 * g + c) * quads];
      float m = f[0][c];
      int am = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (f[k][c] > m) { m = f[k][c]; am = k; }
      const float gv = (m > 0.f) ? gk : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k][c] = (am == k) ? gv : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u32x2 o;
      o.x = pack_bf16(d[k][0], d[k][1]);
      o.y = pack_bf16(d[k][2], d[k][3]);
      __builtin_nontemporal_store(o, dfeat + base + (long)k * groups);
    }
  }
}

// ---- the same pool for C = 32 in tiles of 64 windows, with the backward's routing codes (the fp32 path's pool4_fwd_quad<true> /
// pool4_bwd_idx_quad, layout_pool.hip: 4 bits per window = first maximum | (max > 0) << 2, one 16-bit word per (window, 4 channels)).
// A workgroup owns 64 consecutive windows (256 pixels, 16 KB of bf16) of one image.  Forward: thread = (window, 8 channels), four 16-byte
// loads; the 32 x 64 maxima go through LDS so that every channel plane receives 256 contiguous bytes (one thread per (window, 4
// channels) wrote 32-byte pieces: 3.4 TB/s).  Backward: the gradient tile [32][64] and the codes come in through LDS the same way and the
// 256 pixels leave as one contiguous 16 KB run of whole lines; the 0.48 GB feature is not read again (30 MB of codes instead).
__global__ __launch_bounds__(256) void pool4_bf16_fwd_tile(const u32x4* __restrict__ feat, float* __restrict__ pooled,
                                                           unsigned* __restrict__ idx, long quads, int qblocks) {
  __shared__ float t[32][65];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / qblocks;
  const long q0 = (long)(blockIdx.x - b * qblocks) * 64;
  const int nq = (int)min(64L, quads - q0);
  const int q = tid >> 2, c8 = tid & 3;
  if (q < nq) {
    const u32x4* src = feat + ((b * quads + q0 + q) * 4) * 4 + c8;      // pixel 4(q0 + q), chunk c8; a pixel is 4 chunks of 16 bytes
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(src + 4 * k);
    unsigned codes = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned w = v[k][j >> 1];
        f[k] = (j & 1) ? bf_hi(w) : bf_lo(w);
      }
      float m = f[0];
      unsigned am = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (f[k] > m) { m = f[k]; am = k; }
      t[8 * c8 + j][q] = m;
      codes |= (am | (m > 0.f ? 4u : 0u)) << (4 * j);
    }
    if (idx) idx[(b * quads + q0 + q) * 4 + c8] = codes;      // two 16-bit words: channel groups 2 c8 and 2 c8 + 1
  }
  __syncthreads();
  float* o = pooled + b * (quads * 32) + q0;
  const int qq = tid & 63;
  if (qq < nq) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (tid >> 6) + 4 * j;
      o[(long)c * quads + qq] = t[c][qq];
    }
  }
}

__global__ __launch_bounds__(256) void pool4_bf16_bwd_tile(const float* __restrict__ dpooled, const unsigned* __restrict__ idx,
                                                           u32x4* __restrict__ dfeat, long quads, int qblocks) {
  __shared__ float g[32][65];
  __shared__ unsigned cd[64][4];
  const int tid = threadIdx.x;
  const long b = blockIdx.x / qblocks;
  const long q0 = (long)(blockIdx.x - b * qblocks) * 64;
  const int nq = (int)min(64L, quads - q0);
  const float* gp = dpooled + b * (quads * 32) + q0;
  const int qq = tid & 63;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = (tid >> 6) + 4 * j;
    g[c][qq] = qq < nq ? gp[(long)c * quads + qq] : 0.f;
  }
  cd[tid >> 2][tid & 3] = (tid >> 2) < nq ? idx[(b * quads + q0) * 4 + tid] : 0u;
  __syncthreads();
  u32x4* out = dfeat + (b * quads + q0) * 16;      // 16 chunks of 16 bytes per window
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k;                   // chunk i of the tile: pixel i >> 2, channels 8 (i & 3) ..
    const int px = i >> 2, c8 = i & 3, qd = px >> 2, pos = px & 3;
    if (qd >= nq) continue;
    const unsigned codes = cd[qd][c8];
    float d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned c4 = (codes >> (4 * j)) & 15u;
      d[j] = ((c4 & 4u) && (c4 & 3u) == (unsigned)pos) ? g[8 * c8 + j][qd] : 0.f;
    }
    u32x4 o;
    o.x = pack_bf16(d[0], d[1]); o.y = pack_bf16(d[2], d[3]); o.z = pack_bf16(d[4], d[5]); o.w = pack_bf16(d[6], d[7]);
    __builtin_nontemporal_store(o, out + i);
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const f32x4* __restrict__ src, u32x2* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = src[i];
    u32x2 o;
    o.x = pack_bf16(v.x, v.y);
    o.y = pack_bf16(v.z, v.w);
    dst[i] = o;
  }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const u32x2* __restrict__ src, f32x4* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u32x2 v = src[i];
    dst[i] = f32x4{bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y)};
  }
}

// ------------------------------------------------------------------------------------------------ host side
template <typename K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : dd_fail(DD_ERR_LAUNCH, "hipFuncSetAttribute(%zu bytes LDS): %s", bytes, hipGetErrorString(e));
}

// resident grid: as many workgroups as fit on the CUs this library may use (occupancy from the runtime, cached per kernel)
template <typename K>
int resident_blocks(K kernel, int threads, size_t lds, int* out) {
  static thread_local const void* seen[32];
  static thread_local int per_cu[32];
  static thread_local int nseen = 0;
  int bpc = 0;
  for (int i = 0; i < nseen; ++i)
    if (seen[i] == (const void*)kernel) bpc = per_cu[i];
  if (bpc == 0) {
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)kernel, threads, lds);
    if (e != hipSuccess || bpc < 1) return dd_fail(DD_ERR_LAUNCH, "occupancy query failed: %s", hipGetErrorString(e));
    if (nseen < 32) { seen[nseen] = (const void*)kernel; per_cu[nseen++] = bpc; }
  }
  *out = bpc * dd_cu_budget_internal();
  return 0;
}

int check_desc(const dd_conv_desc* d) {
  DD_REQUIRE(d != nullptr, DD_ERR_BAD_ARG, "conv_bf16: NULL descriptor");
  DD_REQUIRE(d->batch > 0 && d->height > 0 && d->width > 0, DD_ERR_BAD_ARG, "conv_bf16: non-positive size");
  DD_REQUIRE(d->ksize == 3 && d->pad == 1, DD_ERR_UNSUPPORTED, "conv_bf16: only k3 p1 is implemented (got k%d p%d)", d->ksize, d->pad);
  DD_REQUIRE(d->stride == 1 || d->stride == 2, DD_ERR_UNSUPPORTED, "conv_bf16: stride %d", d->stride);
  DD_REQUIRE(d->cout == 32, DD_ERR_UNSUPPORTED, "conv_bf16: Cout %d (only 32)", d->cout);
  DD_REQUIRE((d->cin_real == 32 && d->cin_store == 32) || (d->cin_real == 3 && d->cin_store == 4), DD_ERR_UNSUPPORTED,
             "conv_bf16: Cin %d stored as %d (supported: 32/32, 3/4)", d->cin_real, d->cin_store);
  DD_REQUIRE(!(d->cin_real == 3 && d->stride == 2), DD_ERR_UNSUPPORTED, "conv_bf16: Cin 3 with stride 2");
  DD_REQUIRE((long)d->width * 64 < (1L << 31), DD_ERR_UNSUPPORTED, "conv_bf16: row too long for a 32-bit buffer descriptor");
  return 0;
}

struct Geo {
  int B, H, W, Ho, Wo, nstrips;
};
Geo geo(const dd_conv_desc* d) {
  Geo g;
  g.B = d->batch; g.H = d->height; g.W = d->width;
  g.Ho = (d->height + 2 - 3) / d->stride + 1;
  g.Wo = (d->width + 2 - 3) / d->stride + 1;
  g.nstrips = (g.Wo + 31) / 32;
  return g;
}

constexpr int kWPB = 4;

template <int CIN, int S, int EPI>
int launch_strip(const unsigned short* x, const unsigned short* wp, const float* bias, const unsigned* bits_in, unsigned short* y,
                 unsigned* bits_out, const Geo& g, hipStream_t st) {
  auto kern = bf_strip_fwd<CIN, S, EPI, kWPB>;
  const size_t lds = 128 + (size_t)kWPB * (BCfg<CIN, S>::WAVEB + 2048);
  if (int rc = allow_lds(kern, lds)) return rc;
  int grid = 0;
  if (int rc = resident_blocks(kern, kWPB * 64, lds, &grid)) return rc;
  const long tiles = (long)g.B * g.nstrips * g.Ho;
  grid = (int)min((long)grid, (tiles + kWPB - 1) / kWPB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kWPB * 64), lds, st, x, (const bf16x8*)wp, bias, bits_in, y, bits_out, g.B, g.H, g.W,
                     g.Ho, g.Wo, g.nstrips);
  DD_LAUNCH_CHECK("conv_bf16 strip");
  return 0;
}

template <int CIN, int S>
int wgrad_grid(const Geo& g, int* grid, size_t* lds, bool whole_chip = false) {
  auto kern = bf_wgrad<CIN, S, kWPB>;
  *lds = (size_t)kWPB * WCfg<CIN, S>::WAVEB;
  if (int rc = allow_lds(kern, *lds)) return rc;
  if (int rc = resident_blocks(kern, kWPB * 64, *lds, grid)) return rc;
  if (whole_chip) *grid = *grid / dd_cu_budget_internal() * DD_NUM_CU;     // workspace sizing must not depend on the budget
  const long tiles = (long)g.B * g.nstrips * g.Ho;
  *grid = (int)min((long)*grid, (tiles + kWPB - 1) / kWPB);
  return 0;
}
template <int CIN, int S>
int launch_wgrad(const unsigned short* x, const unsigned short* dy, float* dw, float* db, const Geo& g, void* ws, int64_t ws_bytes,
                 hipStream_t st) {
  int grid = 0;
  size_t lds = 0;
  if (int rc = wgrad_grid<CIN, S>(g, &grid, &lds)) return rc;
  constexpr int NT = WCfg<CIN, S>::NT;
  const long nw = (long)grid * kWPB;
  const long need = nw * (NT * 16 + 1) * 64 * 4;
  DD_REQUIRE(ws_bytes >= need, DD_ERR_BAD_ARG, "conv_bf16_wgrad: workspace %lld < %ld bytes", (long long)ws_bytes, need);
  float* part = (float*)ws;
  float* bpart = part + nw * NT * 16 * 64;
  hipLaunchKernelGGL((bf_wgrad<CIN, S, kWPB>), dim3(grid), dim3(kWPB * 64), lds, st, x, dy, part, bpart, g.B, g.H, g.W, g.Ho, g.Wo,
                     g.nstrips);
  DD_LAUNCH_CHECK("conv_bf16 wgrad");
  hipLaunchKernelGGL(bf_wgrad_reduce<CIN>, dim3(NT * 16 + 1), dim3(1024), 0, st, part, bpart, dw, db, (int)nw);
  DD_LAUNCH_CHECK("conv_bf16 wgrad reduce");
  return 0;
}

}  // namespace

extern "C" {

int64_t dd_conv_bf16_packed_elems(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  return d->cin_real == 32 ? 18 * 64 * 8 : 3 * 64 * 8;
}

int dd_conv_bf16_pack(const float* weight, const dd_conv_desc* d, int32_t kind, uint16_t* packed, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(weight && packed, DD_ERR_BAD_ARG, "conv_bf16_pack: NULL pointer");
  DD_REQUIRE(kind >= 0 && kind <= 2, DD_ERR_BAD_ARG, "conv_bf16_pack: kind %d", kind);
  DD_REQUIRE(kind == 0 || d->cin_real == 32, DD_ERR_UNSUPPORTED, "conv_bf16_pack: no data gradient for the 3-channel layer");
  DD_REQUIRE(kind != 1 || d->stride == 1, DD_ERR_BAD_ARG, "conv_bf16_pack: kind 1 is the stride-1 data gradient");
  DD_REQUIRE(kind != 2 || d->stride == 2, DD_ERR_BAD_ARG, "conv_bf16_pack: kind 2 is the stride-2 data gradient");
  const int n = (int)dd_conv_bf16_packed_elems(d);
  hipLaunchKernelGGL(bf_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, packed, d->cin_real, kind);
  DD_LAUNCH_CHECK("conv_bf16_pack");
  return 0;
}

int dd_conv_bf16_fwd(const uint16_t* x, const uint16_t* packed, const float* bias, uint16_t* y, uint32_t* relu_bits,
                     const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && packed && bias && y, DD_ERR_BAD_ARG, "conv_bf16_fwd: NULL pointer");
  DD_REQUIRE(((uintptr_t)x | (uintptr_t)packed | (uintptr_t)y) % 16 == 0, DD_ERR_BAD_ARG, "conv_bf16_fwd: misaligned buffer");
  const Geo g = geo(d);
  hipStream_t st = (hipStream_t)stream;
  if (d->cin_real == 3) return launch_strip<4, 1, 0>(x, packed, bias, nullptr, y, relu_bits, g, st);
  if (d->stride == 1) return launch_strip<32, 1, 0>(x, packed, bias, nullptr, y, relu_bits, g, st);
  return launch_strip<32, 2, 0>(x, packed, bias, nullptr, y, relu_bits, g, st);
}

int dd_conv_bf16_dgrad(const uint16_t* dy, const uint16_t* packed, const uint32_t* relu_bits, uint16_t* dx,
                       const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(dy && packed && relu_bits && dx, DD_ERR_BAD_ARG, "conv_bf16_dgrad: NULL pointer");
  DD_REQUIRE(d->cin_real == 32, DD_ERR_UNSUPPORTED, "conv_bf16_dgrad: no data gradient for the 3-channel layer");
  DD_REQUIRE(((uintptr_t)dy | (uintptr_t)packed | (uintptr_t)dx) % 16 == 0 && (uintptr_t)relu_bits % 8 == 0, DD_ERR_BAD_ARG,
             "conv_bf16_dgrad: misaligned buffer");
  const Geo g = geo(d);
  hipStream_t st = (hipStream_t)stream;
  if (d->stride == 1) return launch_strip<32, 1, 1>(dy, packed, nullptr, relu_bits, dx, nullptr, g, st);
  auto kern = bf_s2_dgrad<kWPB>;
  const size_t lds = (size_t)kWPB * (BCfg<32, 1>::WAVEB + 4096);
  int grid = 0;
  if (int rc = resident_blocks(kern, kWPB * 64, lds, &grid)) return rc;
  const long tiles = (long)g.B * g.nstrips * ((g.H + 1) / 2);
  grid = (int)min((long)grid, (tiles + kWPB - 1) / kWPB);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kWPB * 64), lds, st, dy, (const bf16x8*)packed, relu_bits, dx, g.B, g.H, g.W, g.Ho,
                     g.Wo, g.nstrips);
  DD_LAUNCH_CHECK("conv_bf16 s2 dgrad");
  return 0;
}

int64_t dd_conv_bf16_wgrad_workspace_bytes(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  const Geo g = geo(d);
  int grid = 0;
  size_t lds = 0;
  int rc;
  int nt;
  if (d->cin_real == 3) { rc = wgrad_grid<4, 1>(g, &grid, &lds, true); nt = 2; }
  else if (d->stride == 1) { rc = wgrad_grid<32, 1>(g, &grid, &lds, true); nt = 9; }
  else { rc = wgrad_grid<32, 2>(g, &grid, &lds, true); nt = 9; }
  if (rc) return -1;
  return (int64_t)grid * kWPB * (nt * 16 + 1) * 64 * 4;
}

int dd_conv_bf16_wgrad(const uint16_t* x, const uint16_t* dy, float* dweight, float* dbias, const dd_conv_desc* d, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && dy && dweight && dbias && workspace, DD_ERR_BAD_ARG, "conv_bf16_wgrad: NULL pointer");
  DD_REQUIRE(((uintptr_t)x | (uintptr_t)dy) % 16 == 0, DD_ERR_BAD_ARG, "conv_bf16_wgrad: misaligned buffer");
  const Geo g = geo(d);
  hipStream_t st = (hipStream_t)stream;
  if (d->cin_real == 3) return launch_wgrad<4, 1>(x, dy, dweight, dbias, g, workspace, workspace_bytes, st);
  if (d->stride == 1) return launch_wgrad<32, 1>(x, dy, dweight, dbias, g, workspace, workspace_bytes, st);
  return launch_wgrad<32, 2>(x, dy, dweight, dbias, g, workspace, workspace_bytes, st);
}

int dd_stitch6_bf16(const float* views, uint16_t* wide_nhwc4, int32_t batch, int32_t height, int32_t width, void* stream) {
  DD_REQUIRE(views && wide_nhwc4 && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6_bf16: bad argument");
  const long npx = (long)batch * height * 6 * width;
  hipLaunchKernelGGL(stitch6_bf16_kernel, dim3((unsigned)min((npx + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                     (hipStream_t)stream, views, (u32x2*)wide_nhwc4, batch, height, width);
  DD_LAUNCH_CHECK("stitch6_bf16");
  return 0;
}

int dd_stitch6_bf16_ptrs(const float* const* sample_ptrs, uint16_t* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                         void* stream) {
  DD_REQUIRE(sample_ptrs && wide_nhwc4 && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6_bf16_ptrs: bad argument");
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    BfSamplePtrs tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "stitch6_bf16_ptrs: null sample pointer");
    const long npx = (long)nb * height * 6 * width;
    hipLaunchKernelGGL(stitch6_bf16_ptrs_kernel, dim3((unsigned)min((npx + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                       (hipStream_t)stream, tab, (u32x2*)wide_nhwc4 + (long)b0 * height * 6 * width, nb, height, width);
    DD_LAUNCH_CHECK("stitch6_bf16_ptrs");
  }
  return 0;
}

int dd_stitch6_bf16_u8_ptrs(const unsigned char* const* sample_ptrs, uint16_t* wide_nhwc4, int32_t batch, int32_t height, int32_t width,
                            void* stream) {
  DD_REQUIRE(sample_ptrs && wide_nhwc4 && batch > 0 && height > 0 && width > 0, DD_ERR_BAD_ARG, "stitch6_bf16_u8_ptrs: bad argument");
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    BfSamplePtrsU8 tab;
    for (int i = 0; i < 64; ++i) tab.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
    for (int i = 0; i < nb; ++i) DD_REQUIRE(tab.p[i] != nullptr, DD_ERR_BAD_ARG, "stitch6_bf16_u8_ptrs: null sample pointer");
    const long npx = (long)nb * height * 6 * width;
    hipLaunchKernelGGL(stitch6_bf16_u8_ptrs_kernel, dim3((unsigned)min((npx + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0,
                       (hipStream_t)stream, tab, (u32x2*)wide_nhwc4 + (long)b0 * height * 6 * width, nb, height, width);
    DD_LAUNCH_CHECK("stitch6_bf16_u8_ptrs");
  }
  return 0;
}

int dd_pool4_bf16_fwd(const uint16_t* feat, float* pooled, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream) {
  DD_REQUIRE(feat && pooled && batch > 0 && h > 0 && w > 0 && c > 0, DD_ERR_BAD_ARG, "pool4_bf16_fwd: bad argument");
  DD_REQUIRE(((long)h * w) % 4 == 0 && c % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_bf16: H*W and C must be multiples of 4 (got %dx%d, C %d)", h, w, c);
  const long total = (long)batch * ((long)h * w / 4) * (c / 4);
  hipLaunchKernelGGL(pool4_bf16_fwd, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x2*)feat, pooled, batch, (long)h * w, c);
  DD_LAUNCH_CHECK("pool4_bf16_fwd");
  return 0;
}

int dd_pool4_relu_bf16_bwd(const float* dpooled, const uint16_t* feat, uint16_t* dfeat, int32_t batch, int32_t h, int32_t w,
                           int32_t c, void* stream) {
  DD_REQUIRE(dpooled && feat && dfeat && batch > 0 && h > 0 && w > 0 && c > 0, DD_ERR_BAD_ARG, "pool4_bf16_bwd: bad argument");
  DD_REQUIRE(((long)h * w) % 4 == 0 && c % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_bf16: H*W and C must be multiples of 4 (got %dx%d, C %d)", h, w, c);
  const long total = (long)batch * ((long)h * w / 4) * (c / 4);
  hipLaunchKernelGGL(pool4_bf16_bwd, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0, (hipStream_t)stream,
                     dpooled, (const u32x2*)feat, (u32x2*)dfeat, batch, (long)h * w, c);
  DD_LAUNCH_CHECK("pool4_bf16_bwd");
  return 0;
}

int64_t dd_pool4_bf16_idx_elems(int32_t batch, int32_t h, int32_t w, int32_t c) {
  const long HW = (long)h * w;
  if (batch <= 0 || h <= 0 || w <= 0 || c != 32 || HW % 4 != 0) {
    dd_fail(DD_ERR_UNSUPPORTED, "pool4_bf16_idx: needs C == 32 and H*W %% 4 == 0, got %d x %d x %d", h, w, c);
    return -1;
  }
  return (long)batch * (HW / 4) * (c / 4);
}

int dd_pool4_bf16_fwd_idx(const uint16_t* feat, float* pooled, uint16_t* idx, int32_t batch, int32_t h, int32_t w, int32_t c,
                          void* stream) {
  DD_REQUIRE(feat && pooled && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "pool4_bf16_fwd_idx: bad argument");
  DD_REQUIRE(c == 32 && ((long)h * w) % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_bf16_fwd_idx: needs C == 32 and H*W %% 4 == 0 (got %dx%d, C %d)", h, w, c);
  DD_REQUIRE(((uintptr_t)feat & 15) == 0 && ((uintptr_t)idx & 3) == 0, DD_ERR_BAD_ARG, "pool4_bf16_fwd_idx: feat must be 16-byte, idx 4-byte aligned");
  const long quads = (long)h * w / 4, qblocks = (quads + 63) / 64;
  DD_REQUIRE(batch * qblocks < (1L << 31), DD_ERR_UNSUPPORTED, "pool4_bf16_fwd_idx: too many tiles");
  hipLaunchKernelGGL(pool4_bf16_fwd_tile, dim3((unsigned)(batch * qblocks)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)feat, pooled,
                     (unsigned*)idx, quads, (int)qblocks);
  DD_LAUNCH_CHECK("pool4_bf16_fwd_idx");
  return 0;
}

int dd_pool4_idx_relu_bf16_bwd(const float* dpooled, const uint16_t* idx, uint16_t* dfeat, int32_t batch, int32_t h, int32_t w,
                               int32_t c, void* stream) {
  DD_REQUIRE(dpooled && idx && dfeat && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "pool4_idx_relu_bf16_bwd: bad argument");
  DD_REQUIRE(c == 32 && ((long)h * w) % 4 == 0, DD_ERR_UNSUPPORTED, "pool4_idx_relu_bf16_bwd: needs C == 32 and H*W %% 4 == 0 (got %dx%d, C %d)", h, w, c);
  DD_REQUIRE(((uintptr_t)dfeat & 15) == 0 && ((uintptr_t)idx & 3) == 0, DD_ERR_BAD_ARG, "pool4_idx_relu_bf16_bwd: dfeat must be 16-byte, idx 4-byte aligned");
  const long quads = (long)h * w / 4, qblocks = (quads + 63) / 64;
  DD_REQUIRE(batch * qblocks < (1L << 31), DD_ERR_UNSUPPORTED, "pool4_idx_relu_bf16_bwd: too many tiles");
  hipLaunchKernelGGL(pool4_bf16_bwd_tile, dim3((unsigned)(batch * qblocks)), dim3(256), 0, (hipStream_t)stream, dpooled, (const unsigned*)idx,
                     (u32x4*)dfeat, quads, (int)qblocks);
  DD_LAUNCH_CHECK("pool4_idx_relu_bf16_bwd");
  return 0;
}

int dd_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
  DD_REQUIRE(src && dst && n > 0 && n % 4 == 0, DD_ERR_BAD_ARG, "f32_to_bf16: bad argument (n must be a multiple of 4)");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)min((n / 4 + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)src, (u32x2*)dst, (long)(n / 4));
  DD_LAUNCH_CHECK("f32_to_bf16");
  return 0;
}

int dd_bf16_to_f32(const uint16_t* src, float* dst, int64_t n, void* stream) {
  DD_REQUIRE(src && dst && n > 0 && n % 4 == 0, DD_ERR_BAD_ARG, "bf16_to_f32: bad argument (n must be a multiple of 4)");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)min((n / 4 + 255) / 256, (long)DD_NUM_CU * 8)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x2*)src, (f32x4*)dst, (long)(n / 4));
  DD_LAUNCH_CHECK("bf16_to_f32");
  return 0;
}

}  // extern "C"
