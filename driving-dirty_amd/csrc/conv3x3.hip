// 3x3 convolution family for the encoder conv stack (reference src/autoencoder/components.py:19-21,41-43)
// as implicit GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32), NHWC activations.
//
// Decomposition ("strip marching"): one WAVE owns a strip of 32 output pixels and marches down the rows of
// an image column.  It keeps the three input rows a 3x3 window needs in its own LDS ring (3 slots),
// prefetches the next row(s) into registers while the matrix cores work on the current row, and never
// synchronises with another wave: no barrier in the main loop.  The GEMM view is M = 32 pixels, N = 32
// output channels, K = 9 taps x Cin; the fp32 MFMA issues once per 64 cycles per SIMD, so LDS (2 x
// ds_read_b128 per 4 MFMAs) and HBM (one 4.3 KB row per 9216 MFMA cycles) are far from binding: the kernels
// are MFMA-issue bound by construction.
//
// Scheduling: the grid is exactly the number of resident workgroups; the B*strips*rows "row tiles" are cut
// into one contiguous, equal range per wave (an image column after the other), so every wave does the same
// number of MFMAs and pays a ring prologue only where its range starts or crosses into the next column.
//
// Memory addressing: every global access is a raw buffer load/store whose descriptor covers ONE image row
// (or zero bytes for a row outside the image).  Out-of-image pixels are then hardware range-check zeros /
// dropped stores: no clamps, no selects, no branches around memory instructions, and an indexing bug cannot
// fault the GPU.
//
//   conv_strip_fwd <CIN,S,EPI>  forward (bias+ReLU epilogue) and stride-1 data gradient (ReLU-mask epilogue)
//   conv_s2_dgrad               data gradient of the stride-2 conv, by output parity class (no zero insertion)
//   conv_wgrad <CIN,S>          weight + bias gradient, register accumulators, deterministic two-stage reduction
#include <stdlib.h>

#include "dd_common.h"

namespace {

// internal epilogues on top of the public DD_EPI_* ones: the ReLU sign travels as ONE BIT per activation
// (a uint32 per 32-channel pixel) instead of being re-read from the 128-byte fp32 pixel by the backward kernels
constexpr int EPI_BIAS_RELU_BITS = 5;   // y = relu(conv + bias), bits[pixel] = mask of (y > 0) over the 32 channels
constexpr int EPI_RELU_BITS = 6;        // y = conv * bit(channel) of bits[pixel]
// Conv -> BatchNorm2d -> ReLU variant (reference components_v2.py:43-46): the conv writes the PRE-normalisation value
// and gathers the batch statistics in its epilogue; the normalise + ReLU is applied by whoever READS the tensor
// (next conv's row loader, pool, mask tests) from a per-channel (scale, shift), so the activation is never rewritten.
constexpr int EPI_BIAS_STATS = 7;       // u = conv + bias; per-lane sum / sum of squares of u -> stats[wave][lane][2]
constexpr int EPI_RELU_MASK_AFF = 8;    // y = conv * (mask*m_scale[c] + m_shift[c] > 0)
constexpr int EPI_RELU_BITS_W1 = 9;     // conv_wino2_fwd only: as EPI_RELU_BITS, but y is not stored -- it feeds the 3 -> 32 layer's weight gradient in place

// per-channel affine tables handed to the kernels: [0:32) input scale, [32:64) input shift, [64:96) mask scale, [96:128) mask shift

template <int CIN, int S>
struct StripCfg {
  static constexpr int PXB = CIN * 4;        // bytes per pixel
  static constexpr int CHUNKS = CIN / 4;     // 16-byte chunks per pixel
  static constexpr int NPX = 32 * S + 2;     // input pixels a 32-wide output strip touches (+1 spare for S=2)
  static constexpr int SLOTB = NPX * PXB;    // one ring slot = one input row of the strip
  static constexpr int NCH = NPX * CHUNKS;
  static constexpr int NLOAD = (NCH + 63) / 64;
  static constexpr int SPILLB = (NLOAD * 64 - NCH) * 16;  // landing zone of the idle lanes of the last chunk group
  static constexpr int WAVEB = 3 * SLOTB + SPILLB;        // LDS per wave: 3-slot ring + landing zone
  static constexpr int KGROUPS = (CIN == 32) ? 36 : 5;   // groups of 4 MFMA k-steps
  static constexpr int WFLOATS = KGROUPS * 64 * 4;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
typedef int i32x4s __attribute__((ext_vector_type(4)));
// The same descriptor as rsrc() as four words (an inline-asm "s" operand): base, num_records = bytes, raw dword access.
__device__ __forceinline__ i32x4s rsrc_words(const void* base, int bytes) {
  const unsigned long a = (unsigned long)base;
  return i32x4s{(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffff), bytes, 0x00020000};
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 2));
}
__device__ __forceinline__ float bload1s(__amdgpu_buffer_rsrc_t r, int off, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 2);
}

// XOR swizzle of the 16-byte chunk index inside a 128-byte pixel so that the ds_read_b128 of 16
// consecutive pixels (one lane group) covers all 64 banks: pixels q and q+1 differ in address bit 7,
// (q>>1)&7 spreads the other 8 pixel pairs over the 8 chunk positions.
template <int CIN>
__device__ __forceinline__ int swz(int q) {
  return CIN == 32 ? ((q >> 1) & 7) : 0;
}

// Row iy of image `img` ([H][W][CIN] floats): pixels gx0 .. gx0+NPX-1 into registers (zeros outside the image).
template <int CIN, int S>
__device__ __forceinline__ void load_row(const float* __restrict__ img, int H, int W, int iy, int gx0, int lane,
                                         f32x4 (&r)[StripCfg<CIN, S>::NLOAD]) {
  using C = StripCfg<CIN, S>;
  const bool rowok = (iy >= 0) && (iy < H);
  const __amdgpu_buffer_rsrc_t rs = rsrc(img + (long)(rowok ? iy : 0) * W * CIN, rowok ? W * C::PXB : 0);
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int c = lane + 64 * i;
    const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
    // a negative pixel index gives a huge unsigned offset: out of range -> zero, like the pixels right of the image
    const int off = (c < C::NCH) ? ((gx0 + q) * C::PXB + ch * 16) : -16;
    r[i] = bload4(rs, off);
  }
}

// The same with ONE multiply, add and select per row instead of a 64-bit base + row * pitch and the 16-bit split of the address:
// base = the image, the row in the scalar offset, num_records = the END of that row.  gfx950 adds the scalar offset in the range
// check (tools/ubench/soffset_probe.hip), so a pixel right of the row is out of range and one left of it (a negative lane offset) too.
template <int CIN, int S>
__device__ __forceinline__ void load_row_img(const float* __restrict__ img, int H, int W, int iy, int gx0, int lane,
                                             f32x4 (&r)[StripCfg<CIN, S>::NLOAD]) {
  using C = StripCfg<CIN, S>;
  const bool rowok = (iy >= 0) && (iy < H);
  const int so = rowok ? iy * W * C::PXB : 0;
  const __amdgpu_buffer_rsrc_t rs = rsrc(img, rowok ? so + W * C::PXB : 0);
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int c = lane + 64 * i;
    const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
    const int off = (c < C::NCH) ? ((gx0 + q) * C::PXB + ch * 16) : -16;
    r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, so, 2));
  }
}

// Registers -> ring slot.  Every lane writes (no exec-masked branch that would drag the matching load and a
// full vmcnt(0) wait into it): the lanes of the last, partial chunk group land in the wave's `spill` zone.
template <int CIN, int S, bool SWZ>
__device__ __forceinline__ void store_row(char* slot, char* spill, int lane,
                                          const f32x4 (&r)[StripCfg<CIN, S>::NLOAD]) {
  using C = StripCfg<CIN, S>;
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int c = lane + 64 * i;
    const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
    char* dst = slot + q * C::PXB + ((ch ^ (SWZ ? swz<CIN>(q) : 0)) << 4);
    if (64 * (i + 1) > C::NCH) dst = (c < C::NCH) ? dst : spill + (c - C::NCH) * 16;
    *(f32x4*)dst = r[i];
  }
}

// Same, for a tensor stored PRE-BatchNorm: in-image pixels become relu(v * scale + shift) on their way into the ring;
// padding stays zero (it pads the normalised activation).  A lane's chunks all cover channels 4*(lane&7)..+3
// (64*i is a multiple of the 8 chunks per pixel), so it carries one (scale, shift) quad.
template <int S, bool SWZ>
__device__ __forceinline__ void store_row_aff(char* slot, char* spill, int lane, const f32x4 (&r)[StripCfg<32, S>::NLOAD],
                                              bool rowok, int gx0, int W, f32x4 sc, f32x4 sh) {
  using C = StripCfg<32, S>;
#pragma unroll
  for (int i = 0; i < C::NLOAD; ++i) {
    const int c = lane + 64 * i;
    const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
    char* dst = slot + q * C::PXB + ((ch ^ (SWZ ? swz<32>(q) : 0)) << 4);
    if (64 * (i + 1) > C::NCH) dst = (c < C::NCH) ? dst : spill + (c - C::NCH) * 16;
    const bool in = rowok && (gx0 + q >= 0) && (gx0 + q < W);
    f32x4 v = r[i];
    v.x = in ? fmaxf(v.x * sc.x + sh.x, 0.f) : 0.f;
    v.y = in ? fmaxf(v.y * sc.y + sh.y, 0.f) : 0.f;
    v.z = in ? fmaxf(v.z * sc.z + sh.z, 0.f) : 0.f;
    v.w = in ? fmaxf(v.w * sc.w + sh.w, 0.f) : 0.f;
    *(f32x4*)dst = v;
  }
}

// The contiguous range of row tiles owned by global wave gw; idx = column * rows + row.
__device__ __forceinline__ void wave_range(long total, int gw, int nw, long& idx, long& end) {
  const long per = (total + nw - 1) / nw;
  idx = (long)gw * per;
  end = min(idx + per, total);
}

// ------------------------------------------------------------------------------------------------
// forward / stride-1 dgrad
// ------------------------------------------------------------------------------------------------
template <int CIN, int S, int EPI, int WPB, bool AFF = false>
__global__ __launch_bounds__(WPB * 64) void conv_strip_fwd(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ msk, float* __restrict__ y,
                                                           unsigned* __restrict__ bits_out, int B, int H, int W, int Ho,
                                                           int Wo, int nstrips, const float* __restrict__ aff = nullptr,
                                                           float* __restrict__ stats = nullptr) {
  using C = StripCfg<CIN, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    f32x4* wl4 = (f32x4*)smem;
    const f32x4* wg4 = (const f32x4*)wp;
    for (int i = tid; i < C::WFLOATS / 4; i += WPB * 64) wl4[i] = wg4[i];
  }
  __syncthreads();
  char* ring = smem + C::WFLOATS * 4 + wave * C::WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  const char* wl = smem;
  const int h = lane >> 5, n = lane & 31;
  const float bv = (EPI == DD_EPI_BIAS || EPI == DD_EPI_BIAS_RELU || EPI == EPI_BIAS_RELU_BITS || EPI == EPI_BIAS_STATS) ? bias[n] : 0.f;
  f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
  if (AFF) {
    asc = *(const f32x4*)(aff + 4 * (lane & 7));
    ash = *(const f32x4*)(aff + 32 + 4 * (lane & 7));
  }
  const float msc = (EPI == EPI_RELU_MASK_AFF) ? aff[64 + n] : 1.f, msh = (EPI == EPI_RELU_MASK_AFF) ? aff[96 + n] : 0.f;
  float st_sum = 0.f, st_sq = 0.f;

  long idx, end;
  wave_range((long)B * nstrips * Ho, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / Ho;
    const int y0 = (int)(idx - col * Ho);
    const int y1 = (int)min((long)Ho, y0 + (end - idx));
    idx += y1 - y0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * CIN;
    const int gx0 = S * x0 - 1;

#pragma unroll
    for (int d = 0; d < 3; ++d) {   // prologue: the three rows output row y0 needs
      f32x4 t[C::NLOAD];
      const int iy = S * y0 - 1 + d;
      load_row<CIN, S>(xb, H, W, iy, gx0, lane, t);
      if constexpr (AFF) store_row_aff<S, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t, iy >= 0 && iy < H, gx0, W, asc, ash);
      else store_row<CIN, S, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t);
    }

    for (int yy = y0; yy < y1; ++yy) {
      // prefetch the S new input rows output row yy+1 needs (in flight under the MFMAs below)
      f32x4 pre[S][C::NLOAD];
#pragma unroll
      for (int s = 0; s < S; ++s) load_row<CIN, S>(xb, H, W, S * yy + 2 + s, gx0, lane, pre[s]);

      const long orow = ((long)(b * Ho + yy) * Wo) * 32;
      float mreg[16];
      if (EPI == DD_EPI_RELU_MASK || EPI == EPI_RELU_MASK_AFF) {
        const __amdgpu_buffer_rsrc_t ms = rsrc(msk + orow, Wo * 128);
#pragma unroll
        for (int r = 0; r < 16; ++r) mreg[r] = bload1(ms, ((x0 + dd_acc_row(r, lane)) * 32 + n) * 4);
      }
      if (EPI == EPI_RELU_BITS) {   // one uint32 per pixel: 32x fewer mask bytes than the fp32 activation
        const __amdgpu_buffer_rsrc_t ms = rsrc((const unsigned*)msk + (long)(b * Ho + yy) * Wo, Wo * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) mreg[r] = bload1(ms, (x0 + dd_acc_row(r, lane)) * 4);
      }
      // keep the loads ABOVE the MFMA chain: left alone, hipcc sinks them to their first use (the ring store
      // behind the chain) and the wave then eats a full HBM round trip per row
      __builtin_amdgcn_sched_barrier(0);

      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;

      if (CIN == 32) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const char* rowb = ring + ((S * yy + dy) % 3) * C::SLOTB;   // slot of input row S*yy-1+dy
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int q = S * n + dx;
            const char* pa = rowb + q * 128;
            const int sw = swz<32>(q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 a = *(const f32x4*)(pa + (((2 * j + h) ^ sw) << 4));
              const f32x4 w = *(const f32x4*)(wl + (((dy * 3 + dx) * 4 + j) * 64 + lane) * 16);
              acc = DD_MFMA(a.x, w.x, acc);
              acc = DD_MFMA(a.y, w.y, acc);
              acc = DD_MFMA(a.z, w.z, acc);
              acc = DD_MFMA(a.w, w.w, acc);
            }
          }
        }
      } else {  // CIN == 4 (3 real channels): k-step = one channel of a PAIR of taps (lower / upper half-wave)
#pragma unroll
        for (int jp = 0; jp < 5; ++jp) {
          const int tap = min(2 * jp + h, 8);   // tap 9 does not exist: its packed weights are zero
          const int dy = tap / 3, dx = tap - 3 * dy;
          const char* rowb = ring + ((S * yy + dy) % 3) * C::SLOTB;
          const f32x4 a = *(const f32x4*)(rowb + (S * n + dx) * 16);
          const f32x4 w = *(const f32x4*)(wl + (jp * 64 + lane) * 16);
          acc = DD_MFMA(a.x, w.x, acc);
          acc = DD_MFMA(a.y, w.y, acc);
          acc = DD_MFMA(a.z, w.z, acc);
        }
      }

      // retire the prefetched rows into the ring BEFORE the epilogue stores are issued: the wait on the
      // (long landed) loads then never has to drain this row's stores (vmcnt counts loads and stores together)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int iy = S * yy + 2 + s;
        if constexpr (AFF) store_row_aff<S, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, pre[s], iy >= 0 && iy < H, gx0, W, asc, ash);
        else store_row<CIN, S, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, pre[s]);
      }

      // epilogue: lane = output channel, register = pixel -> 128 contiguous bytes per pixel per store;
      // pixels right of the image fall outside the row descriptor and are dropped by the hardware
      const __amdgpu_buffer_rsrc_t ys = rsrc(y + orow, Wo * 128);
      unsigned sign_word = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (EPI == DD_EPI_BIAS || EPI == EPI_BIAS_STATS) v += bv;
        if (EPI == EPI_BIAS_STATS && x0 + dd_acc_row(r, lane) < Wo) {   // batch statistics: lane = channel, no shuffles needed
          st_sum += v;
          st_sq += v * v;
        }
        if (EPI == DD_EPI_BIAS_RELU || EPI == EPI_BIAS_RELU_BITS) v = fmaxf(v + bv, 0.f);
        if (EPI == DD_EPI_RELU_MASK) v = (mreg[r] > 0.f) ? v : 0.f;
        if (EPI == EPI_RELU_MASK_AFF) v = (mreg[r] * msc + msh > 0.f) ? v : 0.f;
        if (EPI == EPI_RELU_BITS) v = ((__builtin_bit_cast(unsigned, mreg[r]) >> n) & 1u) ? v : 0.f;
        bstore1(ys, ((x0 + dd_acc_row(r, lane)) * 32 + n) * 4, v);
        if (EPI == EPI_BIAS_RELU_BITS) {
          // lanes 0-31 hold the 32 channels of pixel i = (r&3)+8(r>>2), lanes 32-63 those of pixel i+4: one ballot =
          // two mask words.  Lane p (< 32) keeps the word of strip pixel p, so the row's 32 words leave as ONE store.
          const unsigned long long m = __ballot(v > 0.f);
          const int i0 = (r & 3) + 8 * (r >> 2);
          if (n == i0) sign_word = (unsigned)m;
          if (n == i0 + 4) sign_word = (unsigned)(m >> 32);
        }
      }
      if (EPI == EPI_BIAS_RELU_BITS) {
        const __amdgpu_buffer_rsrc_t bs = rsrc(bits_out + (long)(b * Ho + yy) * Wo, Wo * 4);
        __builtin_amdgcn_raw_buffer_store_b32(sign_word, bs, (h == 0) ? (x0 + n) * 4 : -16, 0, 0);
      }
    }
  }
  if (EPI == EPI_BIAS_STATS) {   // one partial per wave-lane; waves without work still write their zeros
    const long gwl = ((long)(blockIdx.x * WPB + wave) * 64 + lane) * 2;
    stats[gwl] = st_sum;
    stats[gwl + 1] = st_sq;
  }
}

// ------------------------------------------------------------------------------------------------
// stride-2 data gradient.  dx[yi][xi] collects, per parity class (yi&1, xi&1), 1/2/2/4 taps:
//   yi = 2r   : ky = 1 from dy row r            yi = 2r+1 : ky = 0 from row r+1, ky = 2 from row r
// (same along x), so a pair of output rows x 64 output pixels costs the same 144 MFMAs the forward
// spends on 32 output pixels -- no multiplies by inserted zeros.
// ------------------------------------------------------------------------------------------------
template <int WPB, int MASK>   // 0: none, 1: fp32 activation (> 0), 2: packed sign bits, 3: pre-BN tensor with (scale, shift)
__global__ __launch_bounds__(WPB * 64) void conv_s2_dgrad(const float* __restrict__ dy, const float* __restrict__ wp,
                                                          const float* __restrict__ msk, float* __restrict__ dx,
                                                          int B, int H, int W, int Ho, int Wo, int nstrips,
                                                          const float* __restrict__ aff = nullptr) {
  using C = StripCfg<32, 1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    f32x4* wl4 = (f32x4*)smem;
    const f32x4* wg4 = (const f32x4*)wp;
    for (int i = tid; i < C::WFLOATS / 4; i += WPB * 64) wl4[i] = wg4[i];
  }
  __syncthreads();
  const int nr = (H + 1) / 2;   // output row pairs per column
  const float msc = (MASK == 3) ? aff[64 + (lane & 31)] : 1.f, msh = (MASK == 3) ? aff[96 + (lane & 31)] : 0.f;
  char* ring = smem + C::WFLOATS * 4 + wave * C::WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  const char* wl = smem;
  const int h = lane >> 5, n = lane & 31;

  long idx, end;
  wave_range((long)B * nstrips * nr, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / nr;
    const int r0 = (int)(idx - col * nr);
    const int r1 = (int)min((long)nr, r0 + (end - idx));
    idx += r1 - r0;
    const int b = (int)(col / nstrips), s0 = (int)(col % nstrips) * 32;
    const float* dyb = dy + (long)b * Ho * Wo * 32;

#pragma unroll
    for (int d = 0; d < 2; ++d) {
      f32x4 t[C::NLOAD];
      load_row<32, 1>(dyb, Ho, Wo, r0 + d, s0, lane, t);
      store_row<32, 1, true>(ring + ((r0 + d) % 3) * C::SLOTB, spill, lane, t);
    }

    for (int r = r0; r < r1; ++r) {
      f32x4 pre[C::NLOAD];
      load_row<32, 1>(dyb, Ho, Wo, r + 2, s0, lane, pre);
      __builtin_amdgcn_sched_barrier(0);   // loads stay above the MFMA chains (see conv_strip_fwd)

      const char* row_r = ring + (r % 3) * C::SLOTB;
      const char* row_r1 = ring + ((r + 1) % 3) * C::SLOTB;
      // One parity tile at a time: its ReLU-mask values are requested BEFORE its MFMA chain and consumed after,
      // so the epilogue never waits on HBM; only one 32x32 accumulator is live.
      // tap list per tile: (row offset 0/1, pixel offset 0/1, weight tap ky*3+kx)
#define DD_TILE(PY, PX, NTAP, ...)                                                              \
  {                                                                                             \
    constexpr int taps[NTAP][3] = {__VA_ARGS__};                                                \
    const int yi = 2 * r + (PY);                                                                \
    const long orow = ((long)(b * H + min(yi, H - 1)) * W) * 32;                                \
    const int obytes = (yi < H) ? W * 128 : 0;                                                  \
    float mreg[16];                                                                             \
    if (MASK == 1 || MASK == 3) {                                                               \
      const __amdgpu_buffer_rsrc_t ms = rsrc(msk + orow, obytes);                               \
      _Pragma("unroll") for (int rr = 0; rr < 16; ++rr)                                         \
        mreg[rr] = bload1(ms, ((2 * (s0 + dd_acc_row(rr, lane)) + (PX)) * 32 + n) * 4);         \
      __builtin_amdgcn_sched_barrier(0);                                                        \
    }                                                                                           \
    if (MASK == 2) {                                                                            \
      const __amdgpu_buffer_rsrc_t ms = rsrc((const unsigned*)msk + (long)(b * H + min(yi, H - 1)) * W, obytes / 32); \
      _Pragma("unroll") for (int rr = 0; rr < 16; ++rr)                                         \
        mreg[rr] = bload1(ms, (2 * (s0 + dd_acc_row(rr, lane)) + (PX)) * 4);                    \
      __builtin_amdgcn_sched_barrier(0);                                                        \
    }                                                                                           \
    f32x16 acc;                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[i] = 0.f;                                \
    _Pragma("unroll") for (int t = 0; t < NTAP; ++t) {                                          \
      const char* rowp = taps[t][0] ? row_r1 : row_r;                                           \
      const int q = n + taps[t][1];                                                             \
      const char* pa = rowp + q * 128;                                                          \
      const int sw = swz<32>(q);                                                                \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
        const f32x4 a = *(const f32x4*)(pa + (((2 * j + h) ^ sw) << 4));                        \
        const f32x4 w = *(const f32x4*)(wl + ((taps[t][2] * 4 + j) * 64 + lane) * 16);          \
        acc = DD_MFMA(a.x, w.x, acc);                                                           \
        acc = DD_MFMA(a.y, w.y, acc);                                                           \
        acc = DD_MFMA(a.z, w.z, acc);                                                           \
        acc = DD_MFMA(a.w, w.w, acc);                                                           \
      }                                                                                         \
    }                                                                                           \
    const __amdgpu_buffer_rsrc_t os = rsrc(dx + orow, obytes);                                  \
    _Pragma("unroll") for (int rr = 0; rr < 16; ++rr) {                                         \
      float v = acc[rr];                                                                        \
      if (MASK == 1) v = (mreg[rr] > 0.f) ? v : 0.f;                                            \
      if (MASK == 3) v = (mreg[rr] * msc + msh > 0.f) ? v : 0.f;                                \
      if (MASK == 2) v = ((__builtin_bit_cast(unsigned, mreg[rr]) >> n) & 1u) ? v : 0.f;        \
      bstore1(os, ((2 * (s0 + dd_acc_row(rr, lane)) + (PX)) * 32 + n) * 4, v);                  \
    }                                                                                           \
  }
      DD_TILE(1, 1, 4, {1, 1, 0}, {1, 0, 2}, {0, 1, 6}, {0, 0, 8})   // (ky,kx) = (0,0) (0,2) (2,0) (2,2)
      store_row<32, 1, true>(ring + ((r + 2) % 3) * C::SLOTB, spill, lane, pre);   // slot (r+2)%3 is not read by this pair
      DD_TILE(0, 1, 2, {0, 1, 3}, {0, 0, 5})                         // (1,0) (1,2)
      DD_TILE(1, 0, 2, {1, 0, 1}, {0, 0, 7})                         // (0,1) (2,1)
      DD_TILE(0, 0, 1, {0, 0, 4})                                    // (1,1)
#undef DD_TILE
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight / bias gradient.  GEMM view: M = 32 output channels, N = 32 input channels (one 32x32 tile per
// tap, 9 register accumulators), K = pixels.  A = dy (straight from HBM, 256 contiguous bytes per
// wave-load), B = x rows from the LDS ring (ds_read_b32, 32 consecutive dwords per half-wave).
// Each wave accumulates over its whole range and writes ONE partial.
// ------------------------------------------------------------------------------------------------
template <int CIN, int S, int WPB, bool AFF = false>
__global__ __launch_bounds__(WPB * 64) void conv_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ part, float* __restrict__ bpart, int B,
                                                       int H, int W, int Ho, int Wo, int nstrips,
                                                       const float* __restrict__ aff = nullptr) {
  using C = StripCfg<CIN, S>;
  constexpr int NT = (CIN == 32) ? 9 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + wave * C::WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  const int h = lane >> 5, n = lane & 31;
  const int gw = blockIdx.x * WPB + wave;
  f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
  if (AFF) {
    asc = *(const f32x4*)(aff + 4 * (lane & 7));
    ash = *(const f32x4*)(aff + 32 + 4 * (lane & 7));
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // CIN == 4: output column j = lane&31 stands for (tap, channel) = (j/3, j%3); columns >= 27 are ignored.
  const int tap4 = min(n / 3, 8), c4 = n % 3;
  const int dy4 = tap4 / 3, dx4 = tap4 - 3 * dy4;

  long idx, end;
  wave_range((long)B * nstrips * Ho, gw, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / Ho;
    const int y0 = (int)(idx - col * Ho);
    const int y1 = (int)min((long)Ho, y0 + (end - idx));
    idx += y1 - y0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * CIN;
    const float* dyb = dy + (long)b * Ho * Wo * 32;
    const int gx0 = S * x0 - 1;
    const int aoff = ((x0 + h) * 32 + n) * 4;   // + 256 bytes per pixel pair

#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 t[C::NLOAD];
      const int iy = S * y0 - 1 + d;
      load_row<CIN, S>(xb, H, W, iy, gx0, lane, t);
      if constexpr (AFF) store_row_aff<S, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t, iy >= 0 && iy < H, gx0, W, asc, ash);
      else store_row<CIN, S, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t);
    }
    float areg[16];
    {
      const __amdgpu_buffer_rsrc_t as = rsrc(dyb + (long)y0 * Wo * 32, Wo * 128);
#pragma unroll
      for (int pp = 0; pp < 16; ++pp) areg[pp] = bload1(as, aoff + pp * 256);
    }

    for (int yy = y0; yy < y1; ++yy) {
      f32x4 pre[S][C::NLOAD];
#pragma unroll
      for (int s = 0; s < S; ++s) load_row<CIN, S>(xb, H, W, S * yy + 2 + s, gx0, lane, pre[s]);
      float anext[16];
      {
        const bool ok = yy + 1 < Ho;
        const __amdgpu_buffer_rsrc_t as = rsrc(dyb + (long)(ok ? yy + 1 : 0) * Wo * 32, ok ? Wo * 128 : 0);
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) anext[pp] = bload1(as, aoff + pp * 256);
      }
      __builtin_amdgcn_sched_barrier(0);   // loads stay above the MFMA chains (see conv_strip_fwd)

      if (CIN == 32) {
        const char* rb0 = ring + ((S * yy + 0) % 3) * C::SLOTB;
        const char* rb1 = ring + ((S * yy + 1) % 3) * C::SLOTB;
        const char* rb2 = ring + ((S * yy + 2) % 3) * C::SLOTB;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
          const float a = areg[pp];
          bsum += a;
          const int off = (S * (2 * pp + h)) * 128 + n * 4;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            acc[0 + dx] = DD_MFMA(a, *(const float*)(rb0 + off + dx * 128), acc[0 + dx]);
            acc[3 + dx] = DD_MFMA(a, *(const float*)(rb1 + off + dx * 128), acc[3 + dx]);
            acc[6 + dx] = DD_MFMA(a, *(const float*)(rb2 + off + dx * 128), acc[6 + dx]);
          }
        }
      } else {
        const char* rb = ring + ((S * yy + dy4) % 3) * C::SLOTB + dx4 * 16 + c4 * 4;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
          const float a = areg[pp];
          bsum += a;
          acc[0] = DD_MFMA(a, *(const float*)(rb + (S * (2 * pp + h)) * 16), acc[0]);
        }
      }

#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int iy = S * yy + 2 + s;
        if constexpr (AFF) store_row_aff<S, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, pre[s], iy >= 0 && iy < H, gx0, W, asc, ash);
        else store_row<CIN, S, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, pre[s]);
      }
#pragma unroll
      for (int pp = 0; pp < 16; ++pp) areg[pp] = anext[pp];
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(((long)gw * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
  bpart[(long)gw * 64 + lane] = bsum;
}

// Second stage: sum the per-wave partials in a fixed order and scatter to OIHW.  One block per (tap, register)
// row of 64 lanes; 16 wave-groups each sum a strided sixteenth of the partials (4 loads in flight), then a
// fixed-order LDS tree: deterministic, and 75 MB of partials are read by 145 x 1024 threads instead of 145 x 256.
template <int CIN>
__global__ __launch_bounds__(1024) void conv_wgrad_reduce(const float* __restrict__ part,
                                                          const float* __restrict__ bpart, float* __restrict__ dw,
                                                          float* __restrict__ db, int nw) {
  constexpr int NT = (CIN == 32) ? 9 : 1;
  constexpr int G = 16;
  __shared__ double red[G][64];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int row = blockIdx.x;   // (t*16 + r), or NT*16 for the bias
  const float* src = (row < NT * 16) ? part + (long)row * 64 + l : bpart + l;
  const long stride = (row < NT * 16) ? (long)NT * 16 * 64 : 64;
  // The partials of different images largely CANCEL in these sums (the upstream gradient has zero batch mean behind a
  // train-mode BatchNorm, the activations a large common mean), so the second stage runs in fp64: a few thousand adds.
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int w = g;
  for (; w + 3 * G < nw; w += 4 * G) {
    s0 += (double)src[(long)w * stride];
    s1 += (double)src[(long)(w + G) * stride];
    s2 += (double)src[(long)(w + 2 * G) * stride];
    s3 += (double)src[(long)(w + 3 * G) * stride];
  }
  for (; w < nw; w += G) s0 += (double)src[(long)w * stride];
  red[g][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0) return;
  double sd = 0.0;
#pragma unroll
  for (int i = 0; i < G; ++i) sd += red[i][l];
  const float s = (float)sd;
  if (row < NT * 16) {
    const int t = row >> 4, r = row & 15;
    const int o = dd_acc_row(r, l), j = l & 31;
    if (CIN == 32) {
      dw[((long)o * 32 + j) * 9 + t] = s;
    } else if (j < 27) {
      dw[((long)o * 3 + (j % 3)) * 9 + (j / 3)] = s;
    }
  } else {
    const double other = __shfl_xor(sd, 32);   // lanes l and l+32 hold the two pixel parities of channel l&31
    if (l < 32) db[l] = (float)(sd + other);
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing: PyTorch OIHW -> the per-lane B-operand image the kernels read with ds_read_b128.
//   CIN=32: packed[((t*4 + j)*64 + lane)*4 + i] = Weff[n = lane&31][c = 8j + 4(lane>>5) + i][t]
//   CIN=4 : packed[(jp*64 + lane)*4 + i]        = W[n][i][tap = 2jp + (lane>>5)]  (0 for tap 9 / i == 3)
// kind 0: Weff = W;  kind 1: Weff[n][c][t] = W[c][n][8-t] (stride-1 dgrad);  kind 2: Weff[n][c][t] = W[c][n][t].
// ------------------------------------------------------------------------------------------------
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int cin_real, int kind) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cin_real == 32) {
    if (idx >= 36 * 64 * 4) return;
    const int i = idx & 3, lane = (idx >> 2) & 63, g = idx >> 8;
    const int j = g & 3, t = g >> 2;
    const int n = lane & 31, c = 8 * j + 4 * (lane >> 5) + i;
    float v;
    if (kind == 0) v = w[((long)n * 32 + c) * 9 + t];
    else if (kind == 1) v = w[((long)c * 32 + n) * 9 + (8 - t)];
    else v = w[((long)c * 32 + n) * 9 + t];
    p[idx] = v;
  } else {
    if (idx >= 5 * 64 * 4) return;
    const int i = idx & 3, lane = (idx >> 2) & 63, jp = idx >> 8;
    const int n = lane & 31, tap = 2 * jp + (lane >> 5);
    p[idx] = (tap < 9 && i < 3) ? w[((long)n * 3 + i) * 9 + tap] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2,3) along x for the 32 -> 32 stride-1 layer (c2 forward and its data gradient): two adjacent output
// pixels share four transformed inputs, so a 3-tap row costs 4 multiplies instead of 6 -- 192 instead of 288
// matrix-core cycles per pixel pair, in exact fp32 arithmetic (the transforms are adds and one halving):
//   inputs   d0..d3 (pixels 2t-1 .. 2t+2)      v = (d0-d2, d1+d2, d2-d1, d1-d3)
//   weights  g0..g2 (kx = 0..2, per ky)        u = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//   outputs  y(2t) = m0+m1+m2,  y(2t+1) = m1-m2-m3     with m_p = sum over ky, ci of v_p * u_p
// A wave still owns 32 output pixels = 16 pairs; the GEMM tile is 16 (pairs) x 16 (channels) x 4 (input channels) on
// v_mfma_f32_16x16x4_f32 (same flop rate as the 32x32x2 form), 4 positions x 2 channel halves = 8 accumulators of
// 4 registers, issued round-robin (the 16x16 form has a 40-cycle dependent latency on a 32-cycle issue).  Lane
// (pair t = lane&15, q = lane>>4) transforms input channels 8q..8q+7 of its own pair in registers, so the A operand
// never goes back to LDS; U (48 KB) sits in LDS next to the eight 3-slot rings (13.5 KB each): 159.7 of 160 KB.
// Rest of the machinery (ring, prefetch, ranges, buffer addressing) is conv_strip_fwd's.
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define DD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr int WINO_UFLOATS = 3 * 4 * 2 * 2 * 64 * 4;      // [ky][pos][half][chunk][lane][4]

// Packed fp32 add / subtract on 4-channel vectors: the fp32 matrix and vector instructions share the ALUs, so every VALU
// instruction beside the MFMAs costs matrix time; v_pk_add_f32 does two lanes' worth per issue.  (hipcc selects it for
// a 2-vector add but splits a subtract into scalars, hence the explicit form with the negate modifiers.)
typedef float f32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2p pk_sub2(f32x2p a, f32x2p b) {
  f32x2p r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2p pk_add2(f32x2p a, f32x2p b) {      // always packed (hipcc splits a 2-vector add whose operands are not register pairs yet)
  f32x2p r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x4 pk_add4a(f32x4 a, f32x4 b) {
  const f32x2p lo = pk_add2(f32x2p{a.x, a.y}, f32x2p{b.x, b.y}), hi = pk_add2(f32x2p{a.z, a.w}, f32x2p{b.z, b.w});
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b) {
  const f32x2p lo = f32x2p{a.x, a.y} + f32x2p{b.x, b.y}, hi = f32x2p{a.z, a.w} + f32x2p{b.z, b.w};
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b) {
  f32x2p lo, hi;
  const f32x2p alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}

template <int EPI, int WPB>   // EPI_BIAS_RELU_BITS (forward) or EPI_RELU_BITS (data gradient)
__global__ __launch_bounds__(WPB * 64) void conv_wino_fwd(const float* __restrict__ x, const float* __restrict__ up,
                                                          const float* __restrict__ bias, const unsigned* __restrict__ bits_in,
                                                          float* __restrict__ y, unsigned* __restrict__ bits_out, int B, int H,
                                                          int W, int nstrips) {
  using C = StripCfg<32, 1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    f32x4* ul4 = (f32x4*)smem;
    const f32x4* ug4 = (const f32x4*)up;
    for (int i = tid; i < WINO_UFLOATS / 4; i += WPB * 64) ul4[i] = ug4[i];
  }
  __syncthreads();
  char* ring = smem + WINO_UFLOATS * 4 + wave * C::WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  const f32x4* ul = (const f32x4*)smem;
  const int t16 = lane & 15, q4 = lane >> 4;
  const float bv0 = (EPI == EPI_BIAS_RELU_BITS) ? bias[t16] : 0.f, bv1 = (EPI == EPI_BIAS_RELU_BITS) ? bias[16 + t16] : 0.f;

  long idx, end;
  wave_range((long)B * nstrips * H, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / H;
    const int y0 = (int)(idx - col * H);
    const int y1 = (int)min((long)H, y0 + (end - idx));
    idx += y1 - y0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * 32;
    const int gx0 = x0 - 1;

#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 t[C::NLOAD];
      const int iy = y0 - 1 + d;
      load_row<32, 1>(xb, H, W, iy, gx0, lane, t);
      store_row<32, 1, true>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t);
    }

    for (int yy = y0; yy < y1; ++yy) {
      f32x4 pre[C::NLOAD];
      load_row<32, 1>(xb, H, W, yy + 2, gx0, lane, pre);
      const long opix = (long)(b * H + yy) * W;
      unsigned mw[8];
      if (EPI == EPI_RELU_BITS) {   // sign words of this lane's 8 output pixels: pairs 4q..4q+3 = pixels x0 + 8q .. +7
        const __amdgpu_buffer_rsrc_t ms = rsrc(bits_in + opix, W * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i)   // one dword each: every word is range-checked by itself at a ragged row end
          mw[i] = __builtin_amdgcn_raw_buffer_load_b32(ms, (x0 + 8 * q4 + i) * 4, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);

      f32x4v acc[4][2];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) acc[p][hf] = f32x4v{0.f, 0.f, 0.f, 0.f};

      // 6 groups (tap row ky, channel quad g) of 12 reads (4 input pixels + 8 U vectors) + 32 MFMAs; the reads of the
      // next group are issued before the MFMAs of the current one
      f32x4 dbuf[2][4], ubuf[2][4][2];
      auto rd = [&](int it, f32x4 (&d)[4], f32x4 (&u)[4][2]) {
        const int ky = it >> 1, g = it & 1;
        const char* rowb = ring + ((yy + ky) % 3) * C::SLOTB;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int px = 2 * t16 + c;
          d[c] = *(const f32x4*)(rowb + px * 128 + (((2 * q4 + g) ^ swz<32>(px)) << 4));
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) u[p][hf] = ul[((((ky * 4 + p) * 2 + hf) * 2 + g) * 64) + lane];
      };
      // the input transform of a group runs while the PREVIOUS group's MFMAs execute and lands in its own registers: an
      // MFMA never waits for a VALU result, a VALU write never waits for an MFMA that has not read its operand yet
      f32x4 vbuf[2][4];
      auto tf = [&](const f32x4 (&d)[4], f32x4 (&v)[4]) {
        v[0] = pk_sub4(d[0], d[2]);
        v[1] = pk_add4(d[1], d[2]);
        v[2] = pk_sub4(d[2], d[1]);
        v[3] = pk_sub4(d[1], d[3]);
      };
      rd(0, dbuf[0], ubuf[0]);
      tf(dbuf[0], vbuf[0]);
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        if (it + 1 < 6) {
          rd(it + 1, dbuf[(it + 1) & 1], ubuf[(it + 1) & 1]);
          tf(dbuf[(it + 1) & 1], vbuf[(it + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4(&u)[4][2] = ubuf[it & 1];
        const f32x4(&v)[4] = vbuf[it & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) acc[p][hf] = DD_MFMA16(v[p][j], u[p][hf][j], acc[p][hf]);
        __builtin_amdgcn_sched_barrier(0);
      }

      store_row<32, 1, true>(ring + ((yy + 3) % 3) * C::SLOTB, spill, lane, pre);

      // output transform + epilogue: lane = channel t16 (+16 per half), register r = pair 4q + r
      const __amdgpu_buffer_rsrc_t ys = rsrc(y + opix * 32, W * 128);
      unsigned keep[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) keep[i] = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {          // the pair's even / odd pixel
          const int opx = x0 + 2 * (4 * q4 + r) + e;
          float o[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float m0 = acc[0][hf][r], m1 = acc[1][hf][r], m2 = acc[2][hf][r], m3 = acc[3][hf][r];
            float v = e == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
            if (EPI == EPI_BIAS_RELU_BITS) v = fmaxf(v + (hf ? bv1 : bv0), 0.f);
            if (EPI == EPI_RELU_BITS) {
              v = ((mw[2 * r + e] >> (t16 + 16 * hf)) & 1u) ? v : 0.f;
            }
            o[hf] = v;
            bstore1(ys, (opx * 32 + t16 + 16 * hf) * 4, v);
          }
          if (EPI == EPI_BIAS_RELU_BITS) {
            // ballot bit L = (o > 0) of lane L = (channel L&15, pair group L>>4): 16 channel bits of 4 different pixels
            const unsigned long long b0 = __ballot(o[0] > 0.f), b1 = __ballot(o[1] > 0.f);
            // lane P (< 32) keeps the word of strip pixel P = 2*(4G + r) + e
            const int G = (lane & 31) >> 3;
            const unsigned w = (unsigned)((b0 >> (16 * G)) & 0xffffull) | ((unsigned)((b1 >> (16 * G)) & 0xffffull) << 16);
            keep[2 * r + e] = w;
          }
        }
      }
      if (EPI == EPI_BIAS_RELU_BITS) {
        const int P = lane & 31, sel = P & 7;          // P = 8G + 2r + e  ->  keep index 2r + e = P & 7
        unsigned word = keep[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) word = (sel == i) ? keep[i] : word;
        const __amdgpu_buffer_rsrc_t bs = rsrc(bits_out + opix, W * 4);
        __builtin_amdgcn_raw_buffer_store_b32(word, bs, (lane < 32) ? (x0 + P) * 4 : -16, 0, 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3): the same transform along y as well -- a 2x2 output tile from a 4x4 input patch with 16
// multiplies instead of 36 (4/9 of the direct form's matrix-core work, 2/3 of the 1-D form's).  Needs four ring rows
// per wave and 64 KB of transformed weights, which leaves room for ONE wave per SIMD (4-wave workgroups, 134 KB): every
// non-MFMA instruction therefore has to sit in the shadow of an MFMA, and no MFMA may wait for a VALU result.
// Per iteration a wave produces 2 output rows x 32 pixels = 16 tiles: 16 positions x 2 channel halves x 8 k-steps =
// 256 v_mfma_f32_16x16x4_f32 (8192 cycles).  Lane (tile t = lane&15, q = lane>>4) transforms input channels 8q..8q+7
// of its own tile; per channel quad g the x-stage results w[4][4] live in registers, the y-stage row V[u][0..3] of the
// NEXT position row is computed while the current row's 32 MFMAs execute, and its 8 U vectors are read a stage earlier.
// ------------------------------------------------------------------------------------------------
constexpr int WINO2_UFLOATS = 16 * 2 * 2 * 64 * 4;      // [pos = 4u+v][half][chunk][lane][4]

template <int EPI, int WPB>   // EPI_BIAS_RELU_BITS (forward) or EPI_RELU_BITS (data gradient)
__global__ __launch_bounds__(WPB * 64) void conv_wino2_fwd(const float* __restrict__ x, const float* __restrict__ up,
                                                           const float* __restrict__ bias, const unsigned* __restrict__ bits_in,
                                                           float* __restrict__ y, unsigned* __restrict__ bits_out, int B, int H,
                                                           int W, int nstrips, const float* __restrict__ x4 = nullptr,
                                                           float* __restrict__ w1part = nullptr) {
  using C = StripCfg<32, 1>;
  using C4 = StripCfg<4, 1>;
  constexpr int RINGB = 4 * C::SLOTB + C::SPILLB;
  constexpr bool W1 = (EPI == EPI_RELU_BITS_W1);
  constexpr bool MASKED = (EPI == EPI_RELU_BITS) || W1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    f32x4* ul4 = (f32x4*)smem;
    const f32x4* ug4 = (const f32x4*)up;
    for (int i = tid; i < WINO2_UFLOATS / 4; i += WPB * 64) ul4[i] = ug4[i];
  }
  // W1: the 3 -> 32 layer's weight gradient dW1[co][(ky,kx,ci)] += g1[pixel][co] * x[pixel + (ky-1, kx-1)][ci] taken from
  // the outputs while they are in registers (they are exactly the A operand of a 16x16x4 MFMA: lane = channel, k = the
  // four tiles 4q+r of the lane groups); B = the image patch, gathered from a 4-row ring of the NHWC4 input (16 bytes a
  // pixel) in LDS with one ds_read_b32 per MFMA pair.  Column 27 reads a constant 1 (bias gradient), 28..31 are ignored.
  constexpr int W2_XRINGB = 4 * C4::SLOTB + C4::SPILLB;
  constexpr int W2_XBASE = WINO2_UFLOATS * 4 + WPB * RINGB;
  if (W1 && tid < 32) ((float*)(smem + W2_XBASE + WPB * W2_XRINGB))[tid] = 1.f;
  __syncthreads();
  char* xring = smem + W2_XBASE + wave * W2_XRINGB;
  char* xspill = xring + 4 * C4::SLOTB;
  f32x4v wacc[2][2];      // [channel half][column half] of dW1, this wave's share
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) wacc[hf][nh] = f32x4v{0.f, 0.f, 0.f, 0.f};
  char* ring = smem + WINO2_UFLOATS * 4 + wave * RINGB;
  char* spill = ring + 4 * C::SLOTB;
  const f32x4* ul = (const f32x4*)smem;
  const int t16 = lane & 15, q4 = lane >> 4;
  int xky[2], xlane[2];      // W1: this lane's patch column c = t16 + 16 nh -> tap row, byte offset inside a ring slot
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int c = min(t16 + 16 * nh, 26), tap = c / 3, ci = c - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
    xky[nh] = ky;
    xlane[nh] = (8 * q4 + kx) * 16 + ci * 4;
  }
  const float bv0 = (EPI == EPI_BIAS_RELU_BITS) ? bias[t16] : 0.f, bv1 = (EPI == EPI_BIAS_RELU_BITS) ? bias[16 + t16] : 0.f;
  const f32x4v bias0 = {bv0, bv0, bv0, bv0}, bias1 = {bv1, bv1, bv1, bv1};
  const int HT = (H + 1) / 2;                         // tile rows

  long idx, end;
  wave_range((long)B * nstrips * HT, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / HT;
    const int r0 = (int)(idx - col * HT);
    const int r1 = (int)min((long)HT, r0 + (end - idx));
    idx += r1 - r0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * 32;
    const float* x4b = x4 + (long)b * H * W * 4;
    const int gx0 = x0 - 1;

#pragma unroll
    for (int d = 0; d < 4; ++d) {      // input rows 2*r0 - 1 .. 2*r0 + 2; slot of row iy = (iy + 1) & 3
      f32x4 t[C::NLOAD];
      const int iy = 2 * r0 - 1 + d;
      load_row<32, 1>(xb, H, W, iy, gx0, lane, t);
      store_row<32, 1, true>(ring + ((iy + 1) & 3) * C::SLOTB, spill, lane, t);
      if (W1) {      // the same four rows of the image, same slots
        f32x4 t4[C4::NLOAD];
        load_row<4, 1>(x4b, H, W, iy, gx0, lane, t4);
        store_row<4, 1, false>(xring + ((iy + 1) & 3) * C4::SLOTB, xspill, lane, t4);
      }
    }

    // Operand pipeline.  With one wave per SIMD an instruction hides only in the ~24 issue cycles an MFMA leaves free,
    // so (a) every stage is ONE scheduling region in which the 32 MFMAs and the work for LATER stages are interleaved
    // (sched_group_barrier), and (b) nothing in a stage waits for a load issued in the same stage:
    //   patch reads of the next channel quad (or of the next tile-row's first quad)   issued in stage 2 (6)
    //   their x-stage (64 VALU, in place: the old w is dead once V of stage 3 (7) exists)   in stage 3 (7)
    //   U vectors of stage s+1   read in stage s;   V of stage s+1   computed from w in stage s
    //   arriving rows -> ring   in stage 4 (the last patch read of this tile-row left in stage 2)
    auto rowslot = [&](int tr, int r) { return ring + ((2 * tr + r) & 3) * C::SLOTB; };      // slot of input row 2*tr - 1 + r
    auto patch_read = [&](int tr, int g, f32x4 (&d)[4][4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* rp = rowslot(tr, r);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int px = 2 * t16 + c;
          d[r][c] = *(const f32x4*)(rp + px * 128 + (((2 * q4 + g) ^ swz<32>(px)) << 4));
        }
      }
    };
    auto xstage = [&](const f32x4 (&d)[4][4], f32x4 (&w)[4][4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w[r][0] = pk_sub4(d[r][0], d[r][2]);
        w[r][1] = pk_add4(d[r][1], d[r][2]);
        w[r][2] = pk_sub4(d[r][2], d[r][1]);
        w[r][3] = pk_sub4(d[r][1], d[r][3]);
      }
    };
    auto ystage = [&](int u, const f32x4 (&w)[4][4], f32x4 (&v)[4]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (u == 0) v[c] = pk_sub4(w[0][c], w[2][c]);
        else if (u == 1) v[c] = pk_add4(w[1][c], w[2][c]);
        else if (u == 2) v[c] = pk_sub4(w[2][c], w[1][c]);
        else v[c] = pk_sub4(w[1][c], w[3][c]);
      }
    };
    auto uread = [&](int g, int u, f32x4 (&uu)[4][2]) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) uu[v][hf] = ul[((((u * 4 + v) * 2 + hf) * 2 + g) * 64) + lane];
    };

    f32x4 dp[4][4];                 // patch of the quad whose x-stage comes next
    f32x4 wq[4][4];                 // x-stage results of the current channel quad
    f32x4 vq[2][4], uq[2][4][2];    // V row and U vectors of the current / next stage
    patch_read(r0, 0, dp);
    xstage(dp, wq);
    ystage(0, wq, vq[0]);
    uread(0, 0, uq[0]);

    for (int tr = r0; tr < r1; ++tr) {
      f32x4 pre[2][C::NLOAD];
      load_row_img<32, 1>(xb, H, W, 2 * tr + 3, gx0, lane, pre[0]);
      load_row_img<32, 1>(xb, H, W, 2 * tr + 4, gx0, lane, pre[1]);
      f32x4 xpre[2][C4::NLOAD];
      unsigned mw[2][8];
      if (MASKED) {   // sign words of this lane's 2 x 8 output pixels (tiles 4q..4q+3)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int oy = 2 * tr + a;
          const int mso = (oy < H) ? oy * W * 4 : 0;
          const __amdgpu_buffer_rsrc_t ms = rsrc(bits_in + (long)b * H * W, (oy < H) ? mso + W * 4 : 0);
#pragma unroll
          for (int i = 0; i < 8; ++i) mw[a][i] = __builtin_amdgcn_raw_buffer_load_b32(ms, (x0 + 8 * q4 + i) * 4, mso, 0);
        }
      }

      f32x4v acc[16][2];
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int g = st >> 2, u = st & 3;
        __builtin_amdgcn_sched_barrier(0);
        // ---- work for later stages (independent of this stage's MFMAs) ----
        if (st == 2) patch_read(tr, 1, dp);
        if (st == 6) patch_read(tr + 1, 0, dp);
        if (st == 3 || st == 7) xstage(dp, wq);
        if (st + 1 < 8) {
          uread((st + 1) >> 2, (st + 1) & 3, uq[(st + 1) & 1]);
          ystage((st + 1) & 3, wq, vq[(st + 1) & 1]);
        } else {
          uread(0, 0, uq[0]);
          ystage(0, wq, vq[0]);
        }
        if (W1 && st == 7) {      // image rows for the next tile-row: they land after this tile-row's epilogue has read the old ones
          load_row_img<4, 1>(x4b, H, W, 2 * tr + 3, gx0, lane, xpre[0]);
          load_row_img<4, 1>(x4b, H, W, 2 * tr + 4, gx0, lane, xpre[1]);
        }
        if (st == 4) {      // rows 2tr+3, 2tr+4 replace rows 2tr-1, 2tr in the ring
          store_row<32, 1, true>(ring + ((2 * tr + 4) & 3) * C::SLOTB, spill, lane, pre[0]);
          store_row<32, 1, true>(ring + ((2 * tr + 5) & 3) * C::SLOTB, spill, lane, pre[1]);
        }
        // ---- this stage's 32 MFMAs ----
        const f32x4(&vv)[4] = vq[st & 1];
        const f32x4(&uu)[4][2] = uq[st & 1];
        const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              // the first MFMA of an accumulator takes a literal zero (no register clears) -- or, at position (1,1), the
              // bias: A^T e11 A = all ones, so a constant in M[1][1] reaches all four outputs of the tile once
              const f32x4v init = (EPI == EPI_BIAS_RELU_BITS && u == 1 && v == 1) ? (hf ? bias1 : bias0) : zero;
              acc[u * 4 + v][hf] = DD_MFMA16(vv[v][j], uu[v][hf][j], (g == 0 && j == 0) ? init : acc[u * 4 + v][hf]);
            }
        // interleave: one MFMA, then a few of the other instructions, 32 times
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // VALU
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
        }
        __builtin_amdgcn_sched_barrier(0);
      }

      // output transform + epilogue: lane = channel t16 (+16 per half), register r = tile 4q + r
      unsigned keep[2][8];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 8; ++i) keep[a][i] = 0;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int oy = 2 * tr + a;
        const long opix = (long)(b * H + min(oy, H - 1)) * W;
        const __amdgpu_buffer_rsrc_t ys = rsrc(y + (W1 ? 0 : opix * 32), (!W1 && oy < H) ? W * 128 : 0);
        const char* xa[2];      // W1: this lane's patch element of output pixel (row a, tile 4q, e = 0): ring slot of image row oy + ky - 1
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const char* in_ring = xring + ((2 * tr + a + xky[nh]) & 3) * C4::SLOTB + xlane[nh];
          xa[nh] = (t16 + 16 * nh >= 27) ? smem + W2_XBASE + WPB * W2_XRINGB : in_ring;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int opx = x0 + 2 * (4 * q4 + r) + e;
            float o[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              float z[4];      // z[u] = x-direction output transform of position row u
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const float m0 = acc[u * 4 + 0][hf][r], m1 = acc[u * 4 + 1][hf][r], m2 = acc[u * 4 + 2][hf][r], m3 = acc[u * 4 + 3][hf][r];
                z[u] = e == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
              }
              float v = a == 0 ? (z[0] + z[1]) + z[2] : (z[1] - z[2]) - z[3];
              if (EPI == EPI_BIAS_RELU_BITS) v = fmaxf(v, 0.f);      // the bias came in through the accumulator
              // (sign-extended 1-bit field = all ones or zero, then one AND: two vector instructions per element instead of the
              // and / compare / select of `bit ? v : 0`)
              if (MASKED) v = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe((int)mw[a][2 * r + e], (unsigned)(t16 + 16 * hf), 1u));
              o[hf] = v;
              if (!W1) bstore1(ys, (opx * 32 + t16 + 16 * hf) * 4, v);
            }
            if (W1) {
              const float b0 = *(const float*)(xa[0] + (2 * r + e) * 16), b1 = *(const float*)(xa[1] + (2 * r + e) * 16);
              wacc[0][0] = DD_MFMA16(o[0], b0, wacc[0][0]);
              wacc[0][1] = DD_MFMA16(o[0], b1, wacc[0][1]);
              wacc[1][0] = DD_MFMA16(o[1], b0, wacc[1][0]);
              wacc[1][1] = DD_MFMA16(o[1], b1, wacc[1][1]);
            }
            if (EPI == EPI_BIAS_RELU_BITS) {
              const unsigned long long b0 = __ballot(o[0] > 0.f), b1 = __ballot(o[1] > 0.f);
              const int G = (lane & 31) >> 3;
              keep[a][2 * r + e] = (unsigned)((b0 >> (16 * G)) & 0xffffull) | ((unsigned)((b1 >> (16 * G)) & 0xffffull) << 16);
            }
          }
        }
        if (EPI == EPI_BIAS_RELU_BITS) {
          const int P = lane & 31, sel = P & 7;
          unsigned word = keep[a][0];
#pragma unroll
          for (int i = 1; i < 8; ++i) word = (sel == i) ? keep[a][i] : word;
          const __amdgpu_buffer_rsrc_t bs = rsrc(bits_out + opix, (oy < H) ? W * 4 : 0);
          __builtin_amdgcn_raw_buffer_store_b32(word, bs, (lane < 32) ? (x0 + P) * 4 : -16, 0, 0);
        }
      }
      if (W1) {
        store_row<4, 1, false>(xring + ((2 * tr + 4) & 3) * C4::SLOTB, xspill, lane, xpre[0]);
        store_row<4, 1, false>(xring + ((2 * tr + 5) & 3) * C4::SLOTB, xspill, lane, xpre[1]);
      }
    }
  }
  if (W1) {
    const long gw = (long)blockIdx.x * WPB + wave;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i) w1part[((gw * 2 + hf) * 2 + nh) * 256 + i * 64 + lane] = wacc[hf][nh][i];
  }
}

// ------------------------------------------------------------------------------------------------
// The same arithmetic with the input rows held in REGISTERS (default; conv_wino2_fwd above is the LDS-ring form kept for A/B).
// A lane's x-stage results of one input row (its tile, its 8 channels: 8 float4) serve two tile-rows -- rows 2tr+1, 2tr+2 are
// r = 2, 3 of tile-row tr and r = 0, 1 of tile-row tr+1 -- so four such rows (128 VGPRs) rotate and every input row is loaded
// (straight from global memory: 16 bytes a lane, the row's 4.3 KB twice out of L1) and x-transformed ONCE, not twice; no ring,
// no LDS writes, LDS holds only U.  The stages run u-major (s = 2u + g): position row u is complete after stage 2u+1, so the
// accumulators are two rows of 32 registers, and the output transform of row u -- A^T along x, then its term of the y sum --
// sits in the shadow of the next row's MFMAs; output row 2tr leaves in stage 7, output row 2tr+1 in stages 0-1 of the NEXT
// tile-row (the first pass of a column sees an empty store window and zero mask words).  Same operands in the same order as
// conv_wino2_fwd: the same bits.
// ------------------------------------------------------------------------------------------------
template <int EPI, int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_wino2r_fwd(const float* __restrict__ x, const float* __restrict__ up,
                                                            const float* __restrict__ bias, const unsigned* __restrict__ bits_in,
                                                            float* __restrict__ y, unsigned* __restrict__ bits_out, int B, int H,
                                                            int W, int nstrips, const float* __restrict__ x4 = nullptr,
                                                            float* __restrict__ w1part = nullptr, int pf_rows = 0) {
  using C4 = StripCfg<4, 1>;
  constexpr bool W1 = (EPI == EPI_RELU_BITS_W1);
  constexpr bool MASKED = (EPI == EPI_RELU_BITS) || W1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    // U image -> LDS with the input channels regrouped: here lane group q4 and quad g hold channels 16 g + 4 q4 .. + 3 (the
    // packed image has 8 q4 + 4 g .. + 3), so that the four lane groups of a patch load read 64 contiguous bytes of a pixel.
    // [pos][half][g][lane = 16 q4 + t16]  <-  [pos][half][q4 & 1][16 (2 g + (q4 >> 1)) + t16]
    f32x4* ul4 = (f32x4*)smem;
    const f32x4* ug4 = (const f32x4*)up;
#pragma unroll 4
    for (int i = tid; i < WINO2_UFLOATS / 4; i += WPB * 64) {
      const int ln = i & 63, gq = (i >> 6) & 1, hi = i >> 7, qq = ln >> 4;
      ul4[i] = ug4[(hi * 2 + (qq & 1)) * 64 + (2 * gq + (qq >> 1)) * 16 + (ln & 15)];
    }
  }
  constexpr int W2_XRINGB = 4 * C4::SLOTB + C4::SPILLB;
  constexpr int W2_XBASE = WINO2_UFLOATS * 4;
  if (W1 && tid < 32) ((float*)(smem + W2_XBASE + WPB * W2_XRINGB))[tid] = 1.f;
  if (W1)      // the first pass of a column multiplies masked-out zeros with whatever the image ring holds: keep that finite
    for (int i = tid; i < WPB * W2_XRINGB / 16; i += WPB * 64) ((f32x4*)(smem + W2_XBASE))[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  char* xring = smem + W2_XBASE + wave * W2_XRINGB;      // W1: 4-row ring of the NHWC4 image, slot of row iy = (iy + 1) & 3
  char* xspill = xring + 4 * C4::SLOTB;
  f32x4v wacc[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) wacc[hf][nh] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int ulane = (int)(unsigned long)(__attribute__((address_space(3))) char*)smem + lane * 16;      // LDS byte address of U[..][lane]
  const int t16 = lane & 15, q4 = lane >> 4;
  int xky[2], xlane[2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int c = min(t16 + 16 * nh, 26), tap = c / 3, ci = c - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
    xky[nh] = ky;
    xlane[nh] = (8 * q4 + kx) * 16 + ci * 4;
  }
  const float bv0 = (EPI == EPI_BIAS_RELU_BITS) ? bias[t16] : 0.f, bv1 = (EPI == EPI_BIAS_RELU_BITS) ? bias[16 + t16] : 0.f;
  const f32x4v bias0 = {bv0, bv0, bv0, bv0}, bias1 = {bv1, bv1, bv1, bv1};
  const int HT = (H + 1) / 2;

  f32x4v accr[2][4][2];      // [position row u & 1][v][channel half]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) accr[i][v][hf] = f32x4v{0.f, 0.f, 0.f, 0.f};
  f32x2p o1[2][2][2];        // [register pair][column e][channel half]: output row 2tr+1 on its way (z1 - z2 - z3)
#pragma unroll
  for (int i = 0; i < 8; ++i) o1[i >> 2][(i >> 1) & 1][i & 1] = f32x2p{0.f, 0.f};

  float pfacc = 0.f;
  long idx, end;
  wave_range((long)B * nstrips * HT, blockIdx.x * WPB + wave, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / HT;
    const int r0 = (int)(idx - col * HT);
    const int r1 = (int)min((long)HT, r0 + (end - idx));
    idx += r1 - r0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * 32;
    const float* x4b = x4 + (long)b * H * W * 4;
    // Descriptors: ONE per tensor and column -- base = the image, num_records = the whole image; the row goes into the scalar
    // offset (range-checked together with the lane offset on this part: tools/ubench/soffset_probe.hip), so rows -1 and H fall
    // out of range by themselves, and the pixels left and right of a row are put out of range by the LANE offsets, which do not
    // change down a column (poff, pst, boff below).  A descriptor per row (base + row * pitch in 64 bits, the 16-bit split of the
    // address, the size select) cost the wave ~500 of its ~11,700 cycles per tile-row.
    float* yb = y + (W1 ? 0 : (long)b * H * W * 32);
    unsigned* bob = bits_out + (EPI == EPI_BIAS_RELU_BITS ? (long)b * H * W : 0);
    const unsigned* bib = bits_in + (MASKED ? (long)b * H * W : 0);
    const int pitch = W * 128, pitchb = W * 4;
    constexpr int FAR = 1 << 30;      // beyond any image (the launcher checks H * W * 128 < 2^30), no wrap with a row offset on top
    const __amdgpu_buffer_rsrc_t xrs = rsrc(xb, H * pitch), bors = rsrc(bob, H * pitchb), birs = rsrc(bib, H * pitchb);
    const i32x4s yrs = rsrc_words(yb, W1 ? 0 : H * pitch);
    const int gx0 = x0 - 1;
    int poff[4];      // byte offset of this lane's patch column c inside an input row (left / right of the image: out of range -> 0)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int px = gx0 + 2 * t16 + c;
      poff[c] = (px >= 0 && px < W) ? px * 128 + q4 * 16 : FAR;
    }
    int pst[8];       // byte offset of this lane's output pixel 2r + e of a tile-row's 8, channel t16 (+ 16 hf: the instruction offset)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int px = x0 + 8 * q4 + q;
      pst[q] = (px < W) ? px * 128 + t16 * 4 : FAR;
    }
    int pmk[8];       // ... and of its sign word (zero right of the row: with W1 the masked gradient of such a pixel meets image pixels W-1)
#pragma unroll
    for (int q = 0; q < 8; ++q) pmk[q] = (MASKED && x0 + 8 * q4 + q < W) ? (x0 + 8 * q4 + q) * 4 : FAR;
    // L2 prefetch (pf_rows tile-rows ahead): one dword of each 64-byte half of the strip's 34 pixels of an input row
    const int pfoff = (lane < 34 && gx0 + lane >= 0 && gx0 + lane < W) ? (gx0 + lane) * 128 : FAR;
    const int boff = (lane < 32 && x0 + (lane & 31) < W) ? (x0 + (lane & 31)) * 4 : FAR;      // the sign word lane P < 32 stores

    auto row_load = [&](int iy, f32x4 (&R)[2][4]) {      // input row iy: patch columns 2 t16 .. + 3, channels 16 g + 4 q4 .. + 3
      const int so = iy * pitch;      // row -1: a huge unsigned offset, out of range like row H
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) R[g][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, poff[c] + g * 64, so, 0));
    };
    auto xstage = [&](const f32x4 (&L)[2][4], f32x4 (&R)[2][4]) {      // landing registers -> row registers (or in place)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 d0 = L[g][0], d1 = L[g][1], d2 = L[g][2], d3 = L[g][3];
        R[g][0] = pk_sub4(d0, d2);
        R[g][1] = pk_add4a(d1, d2);
        R[g][2] = pk_sub4(d2, d1);
        R[g][3] = pk_sub4(d1, d3);
      }
    };
    auto ystage = [&](int u, int g, const f32x4 (&A0)[2][4], const f32x4 (&A1)[2][4], const f32x4 (&A2)[2][4], const f32x4 (&A3)[2][4],
                      f32x4 (&v)[4]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (u == 0) v[c] = pk_sub4(A0[g][c], A2[g][c]);
        else if (u == 1) v[c] = pk_add4a(A1[g][c], A2[g][c]);
        else if (u == 2) v[c] = pk_sub4(A2[g][c], A1[g][c]);
        else v[c] = pk_sub4(A1[g][c], A3[g][c]);
      }
    };
    // U vectors go from LDS straight into ACCUMULATION registers and from there into the MFMAs' B operand: they never hold a
    // VGPR (64 of them in the ring form).  The compiler does not see these reads: u_wait() below stands between them and their use.
    auto uread = [&](int g, int u, f32x4 (&uu)[4][2]) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(uu[v][hf]) : "v"(ulane), "n"(((((u * 4 + v) * 2 + hf) * 2 + g) * 1024)));
    };
    auto uread1 = [&](int g, int u, int v, int hf, f32x4& d) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(ulane), "n"(((((u * 4 + v) * 2 + hf) * 2 + g) * 1024)));
    };
    auto u_wait = [&](f32x4 (&uu)[4][2]) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+a"(uu[0][0]), "+a"(uu[0][1]), "+a"(uu[1][0]), "+a"(uu[1][1]), "+a"(uu[2][0]), "+a"(uu[2][1]), "+a"(uu[3][0]), "+a"(uu[3][1]));
    };
    auto mask_load = [&](int oy, unsigned (&m)[8]) {      // sign words of this lane's 8 output pixels (tiles 4q .. 4q+3) of row oy
      const int so = oy * pitchb;
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = __builtin_amdgcn_raw_buffer_load_b32(birs, pmk[i], so, 0);
    };
    auto acc_read2 = [&](float a0, float a1, f32x2p& d) {
      float lo, hi;
      asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(lo), "=v"(hi) : "a"(a0), "a"(a1));
      d = f32x2p{lo, hi};
    };
    // x-direction output transform of position row u (accumulators accr[u & 1]) and its term of the two y sums.  The reads are
    // pinned to the stage that calls this (volatile): left to itself the compiler puts each one right behind the MFMA that
    // finishes the accumulator, a stage earlier, and waits there for the result.
    auto transform_row = [&](int u, f32x2p (&o0)[2][2][2]) {
      asm volatile("s_nop 7\n\ts_nop 7");      // the last MFMA of accr[u & 1] is >= 11 wait states away (the compiler cannot count for us)
#pragma unroll
      for (int rp = 0; rp < 2; ++rp)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4v(&m)[4][2] = accr[u & 1];
          f32x2p mm[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) acc_read2(m[v][hf][2 * rp], m[v][hf][2 * rp + 1], mm[v]);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const f32x2p z = e == 0 ? pk_add2(pk_add2(mm[0], mm[1]), mm[2]) : pk_sub2(pk_sub2(mm[1], mm[2]), mm[3]);
            if (u == 0) o0[rp][e][hf] = z;
            if (u == 1) { o0[rp][e][hf] = pk_add2(o0[rp][e][hf], z); o1[rp][e][hf] = z; }
            if (u == 2) { o0[rp][e][hf] = pk_add2(o0[rp][e][hf], z); o1[rp][e][hf] = pk_sub2(o1[rp][e][hf], z); }
            if (u == 3) o1[rp][e][hf] = pk_sub2(o1[rp][e][hf], z);
          }
        }
    };
    // output row oy leaves: ReLU / mask, store, sign words, the c1 weight gradient's MFMAs
    auto emit_row = [&](int oy, const f32x2p (&o)[2][2][2], const unsigned (&mw)[8], float (&pend)[16], int& yso) {
      yso = oy * pitch;
      const char* xa[2];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const char* in_ring = xring + ((oy + xky[nh]) & 3) * C4::SLOTB + xlane[nh];
        xa[nh] = (t16 + 16 * nh >= 27) ? smem + W2_XBASE + WPB * W2_XRINGB : in_ring;
      }
      unsigned keep[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) keep[i] = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float ov[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            float v = o[r >> 1][e][hf][r & 1];
            if (EPI == EPI_BIAS_RELU_BITS) v = (v > 0.f) ? v : 0.f;      // one compare serves the ReLU and the sign ballot
            if (MASKED) v = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & __builtin_amdgcn_sbfe((int)mw[2 * r + e], (unsigned)(t16 + 16 * hf), 1u));
            ov[hf] = v;
            pend[(2 * r + e) * 2 + hf] = v;      // stored from inside the MFMA block (store_pending)
          }
          if (W1) {
            const float b0 = *(const float*)(xa[0] + (2 * r + e) * 16), b1 = *(const float*)(xa[1] + (2 * r + e) * 16);
            wacc[0][0] = DD_MFMA16(ov[0], b0, wacc[0][0]);
            wacc[0][1] = DD_MFMA16(ov[0], b1, wacc[0][1]);
            wacc[1][0] = DD_MFMA16(ov[1], b0, wacc[1][0]);
            wacc[1][1] = DD_MFMA16(ov[1], b1, wacc[1][1]);
          }
          if (EPI == EPI_BIAS_RELU_BITS) {
            const unsigned long long b0 = __ballot(ov[0] > 0.f), b1 = __ballot(ov[1] > 0.f);
            const int G = (lane & 31) >> 3;
            keep[2 * r + e] = (unsigned)((b0 >> (16 * G)) & 0xffffull) | ((unsigned)((b1 >> (16 * G)) & 0xffffull) << 16);
          }
        }
      }
      if (EPI == EPI_BIAS_RELU_BITS) {
        const int P = lane & 31, sel = P & 7;
        unsigned word = keep[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) word = (sel == i) ? keep[i] : word;
        __builtin_amdgcn_raw_buffer_store_b32(word, bors, boff, oy * pitchb, 0);
      }
    };

    // Output element (2r+e, hf) of this lane: pixel x0 + 8 q4 + 2r + e, channel t16 + 16 hf.  A 64-lane dword store costs the wave
    // ~16 issue cycles in the vector block and nothing behind an MFMA (the matrix pipe is busy for 32): the stores of a row are
    // issued one per MFMA.  (The compiler does not see them; it can only over-wait for its own loads because of that.)
    auto store_pending = [&](int i, const float (&pend)[16], int yso) {
      if (i & 1) asm volatile("buffer_store_dword %0, %1, %2, %3 offen offset:64 nt" : : "v"(pend[i]), "v"(pst[i >> 1]), "s"(yrs), "s"(yso) : "memory");
      else asm volatile("buffer_store_dword %0, %1, %2, %3 offen nt" : : "v"(pend[i]), "v"(pst[i >> 1]), "s"(yrs), "s"(yso) : "memory");
    };
    f32x4 R0[2][4], R1[2][4], R2[2][4], R3[2][4];      // x-stage results of four input rows
    f32x4 vq[4], uq[2][4][2];                          // V row of the current stage; U vectors of the current / next stage (AGPRs)
    unsigned m1w[8];                                   // sign words of output row 2tr+1 (loaded in stage 7, used in the next stage 1)
    f32x4 xpre[2][C4::NLOAD];
#pragma unroll
    for (int i = 0; i < 8; ++i) m1w[i] = 0;
    row_load(2 * r0 - 1, R0);
    row_load(2 * r0, R1);
    row_load(2 * r0 + 1, R2);
    row_load(2 * r0 + 2, R3);
    if (W1) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {      // image rows 2 r0 - 1, 2 r0 into the ring; 2 r0 + 1, 2 r0 + 2 follow in the first pass
        f32x4 t4[C4::NLOAD];
        load_row<4, 1>(x4b, H, W, 2 * r0 - 1 + d, gx0, lane, t4);
        store_row<4, 1, false>(xring + ((2 * r0 + d) & 3) * C4::SLOTB, xspill, lane, t4);
      }
      load_row<4, 1>(x4b, H, W, 2 * r0 + 1, gx0, lane, xpre[0]);
      load_row<4, 1>(x4b, H, W, 2 * r0 + 2, gx0, lane, xpre[1]);
    }
    xstage(R0, R0);
    xstage(R1, R1);
    xstage(R2, R2);
    uread(0, 0, uq[0]);
    float pfv[4] = {0.f, 0.f, 0.f, 0.f};
    int oy_prev = H;      // nothing to emit in the first pass

    // One tile-row.  A0 .. A3 = the register rows holding input rows 2tr-1 .. 2tr+2; A0 is refilled with row 2tr+3 and A1 with
    // row 2tr+4 (the next tile-row's r = 2, 3).  mp = sign words of the previous tile-row's second output row, mn = this one's.
    auto step = [&](int tr, f32x4 (&A0)[2][4], f32x4 (&A1)[2][4], f32x4 (&A2)[2][4], f32x4 (&A3)[2][4]) {
      f32x2p o0[2][2][2];
      unsigned m0w[8];
      float pend[16];      // the outputs of a row between the vector block that forms them and their stores
      int yso;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int u = st >> 1, g = st & 1;
        __builtin_amdgcn_sched_barrier(0);
        ystage(u, g, A0, A1, A2, A3, vq);
        // ---- work for other stages / tile-rows ----
        if (st == 0) transform_row(3, o0);                       // the previous tile-row's last position row (o0 untouched)
        if (st == 1) {
          emit_row(oy_prev, o1, m1w, pend, yso);
          row_load(2 * tr + 3, A0);                              // A0 is dead: V(u = 0, g = 1) has just been formed
        }
        if (st == 2) {
          if (pf_rows) {
            pfacc += pfv[0] + pfv[1] + pfv[2] + pfv[3];      // the previous tile-row's (landed long ago; keeps them alive)
#pragma unroll
            for (int q = 0; q < 4; ++q) pfv[q] = bload1s(xrs, pfoff + (q & 1) * 64, (2 * (tr + pf_rows) + 3 + (q >> 1)) * pitch);
          }
          transform_row(0, o0);
          if (W1) {
            store_row<4, 1, false>(xring + ((2 * tr + 2) & 3) * C4::SLOTB, xspill, lane, xpre[0]);
            store_row<4, 1, false>(xring + ((2 * tr + 3) & 3) * C4::SLOTB, xspill, lane, xpre[1]);
          }
        }
        if (st == 3) {
          xstage(A3, A3);                                        // input row 2tr+2, in flight since the previous tile-row's stage 7
          if (W1) {
            load_row<4, 1>(x4b, H, W, 2 * tr + 3, gx0, lane, xpre[0]);
            load_row<4, 1>(x4b, H, W, 2 * tr + 4, gx0, lane, xpre[1]);
          }
        }
        if (st == 4) transform_row(1, o0);
        if (st == 5) {
          xstage(A0, A0);                                        // input row 2tr+3, in flight since stage 1
          if (MASKED) mask_load(2 * tr, m0w);
        }
        if (st == 6) transform_row(2, o0);
        if (st == 7) {
          emit_row(2 * tr, o0, m0w, pend, yso);
          if (MASKED) mask_load(2 * tr + 1, m1w);
          row_load(2 * tr + 4, A1);                              // A1 is dead: V(u = 3, g = 1) has just been formed
        }
        // ---- this stage's 32 MFMAs: back to back in a region of their own.  Every switch between vector and matrix
        // instructions costs the wave ~5 cycles on top of the instructions themselves (tools/ubench/mfma_issue.hip: 48.7
        // cycles per MFMA with two v_pk_add_f32 behind each, 43.5 with 64 behind 32) and with one wave per SIMD nothing
        // overlaps anyway; written as volatile asm because the scheduler interleaves whatever it is allowed to move.
        // (Operands: V was formed at the top of the stage, U has arrived by u_wait(), the accumulators are read a stage later.)
        __builtin_amdgcn_sched_barrier(0);
        u_wait(uq[st & 1]);
        const f32x4(&vv)[4] = vq;
        const f32x4(&uu)[4][2] = uq[st & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              f32x4v& acc = accr[u & 1][v][hf];
              if (g == 0 && j == 0) {
                // the first MFMA of an accumulator takes a literal zero -- or, at position (1,1), the bias: A^T e11 A = all
                // ones, so a constant in M[1][1] reaches all four outputs of the tile once
                if (EPI == EPI_BIAS_RELU_BITS && u == 1 && v == 1)
                  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=a"(acc) : "v"(vv[v][j]), "a"(uu[v][hf][j]), "a"(hf ? bias1 : bias0));
                else
                  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(vv[v][j]), "a"(uu[v][hf][j]));
              } else {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(vv[v][j]), "a"(uu[v][hf][j]));
              }
              // behind the MFMA, for free: the next stage's U vectors (one read each behind the first eight), a row's stores
              const int i = (j * 4 + v) * 2 + hf;
              if (i < 8) uread1((st + 1) & 1, ((st + 1) & 7) >> 1, i >> 1, i & 1, uq[(st + 1) & 1][i >> 1][i & 1]);
              if (!W1 && (st == 1 || st == 7) && i >= 8 && i < 24) store_pending(i - 8, pend, yso);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      oy_prev = 2 * tr + 1;
    };

    int tr = r0;
    for (;;) {
      step(tr, R0, R1, R2, R3);
      if (++tr >= r1) break;
      step(tr, R2, R3, R0, R1);
      if (++tr >= r1) break;
    }
    {      // the column's last output row
      f32x2p unused[2][2][2];
      float pend[16];
      int yso;
      transform_row(3, unused);
      emit_row(oy_prev, o1, m1w, pend, yso);
      if (!W1)
#pragma unroll
        for (int q = 0; q < 16; ++q) store_pending(q, pend, yso);
    }
  }
  asm volatile("" : : "v"(pfacc));      // the prefetched words must not be optimised away
  if (W1) {
    const long gw = (long)blockIdx.x * WPB + wave;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; ++i) w1part[((gw * 2 + hf) * 2 + nh) * 256 + i * 64 + lane] = wacc[hf][nh][i];
  }
}

// Reduce of conv_wino2_fwd<EPI_RELU_BITS_W1>'s per-wave dW1 partials ([wave][half][column half][reg][lane]) in a fixed order and
// scatter to OIHW [32][3][3][3] + bias [32]: a block owns 8 elements, 32 thread groups share the waves.
__global__ __launch_bounds__(256) void conv_w1_reduce(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                      int nw) {
  __shared__ float red[32][8];
  const int el = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int e = blockIdx.x * 8 + el;      // < 1024
  float s0 = 0.f, s1 = 0.f;
  int w = g;
  for (; w + 32 < nw; w += 64) {
    s0 += part[(long)w * 1024 + e];
    s1 += part[(long)(w + 32) * 1024 + e];
  }
  if (w < nw) s0 += part[(long)w * 1024 + e];
  red[g][el] = s0 + s1;
  __syncthreads();
  if (g != 0) return;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) v += red[i][el];
  const int l = e & 63, reg = (e >> 6) & 3, nh = (e >> 8) & 1, hf = e >> 9;
  const int co = 16 * hf + 4 * (l >> 4) + reg, col = 16 * nh + (l & 15);
  if (col < 27) {
    const int tap = col / 3, ci = col - 3 * tap;
    dw[(co * 3 + ci) * 9 + tap] = v;
  } else if (col == 27) {
    db[co] = v;
  }
}

// U image for conv_wino2_fwd: packed[((((4u+v)*2 + half)*2 + g)*64 + lane)*4 + j] = (G Weff G^T)[u][v] of
// Weff[co = (lane&15) + 16*half][ci = 8*(lane>>4) + 4g + j]; kind as conv_wino_pack_kernel.
__global__ void conv_wino2_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int kind) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= WINO2_UFLOATS) return;
  const int j = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 1, hf = (idx >> 9) & 1, pos = idx >> 10;
  const int u = pos >> 2, v = pos & 3;
  const int co = (lane & 15) + 16 * hf, ci = 8 * (lane >> 4) + 4 * g + j;
  float t[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      t[ky][kx] = kind == 0 ? w[((long)co * 32 + ci) * 9 + ky * 3 + kx] : w[((long)ci * 32 + co) * 9 + (2 - ky) * 3 + (2 - kx)];
  float rowt[3];      // x-direction transform of each tap row
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    if (v == 0) rowt[ky] = t[ky][0];
    else if (v == 1) rowt[ky] = 0.5f * ((t[ky][0] + t[ky][1]) + t[ky][2]);
    else if (v == 2) rowt[ky] = 0.5f * ((t[ky][0] - t[ky][1]) + t[ky][2]);
    else rowt[ky] = t[ky][2];
  }
  float out;
  if (u == 0) out = rowt[0];
  else if (u == 1) out = 0.5f * ((rowt[0] + rowt[1]) + rowt[2]);
  else if (u == 2) out = 0.5f * ((rowt[0] - rowt[1]) + rowt[2]);
  else out = rowt[2];
  p[idx] = out;
}

// Weight gradient of the same layer by F(3,2) along x (the transpose of the algorithm above): per tile of two
// adjacent output pixels the three taps' products  dW_k += g_j * x_{j+k}  (j = 0,1; k = 0..2) are
//   dW_k = A^T[k][:] . ( (G g) * (B^T d) ),   G g = (g0, (g0+g1)/2, (g0-g1)/2, g1),   B^T d = (d0-d2, d1+d2, d2-d1, d3-d1)
// and the sum over tiles commutes with A^T, so the kernel keeps 3 (ky) x 4 (position) accumulators S_p of 32 x 32
// over K = tiles -- 96 instead of 144 MFMAs per 32 pixels -- and the reduce kernel applies
//   dW_0 = S0+S1+S2,  dW_1 = S1-S2,  dW_2 = S1+S2+S3     once at the end.
// GEMM view as in conv_wgrad: A = G g (lane = output channel, straight from HBM), B = B^T d (lane = input channel,
// four ds_read_b32 per tile and tap row), k-step = the tile pair (2s, 2s+1) on the two half-waves.
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_wino_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ part, float* __restrict__ bpart, int B,
                                                            int H, int W, int nstrips) {
  using C = StripCfg<32, 1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + wave * C::WAVEB;
  char* spill = ring + 3 * C::SLOTB;
  const int h = lane >> 5, n = lane & 31;
  const int gw = blockIdx.x * WPB + wave;

  f32x16 acc[3][4];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ky][p][r] = 0.f;
  float bsum = 0.f;

  long idx, end;
  wave_range((long)B * nstrips * H, gw, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / H;
    const int y0 = (int)(idx - col * H);
    const int y1 = (int)min((long)H, y0 + (end - idx));
    idx += y1 - y0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * 32;
    const float* dyb = dy + (long)b * H * W * 32;
    const int gx0 = x0 - 1;
    const int aoff = ((x0 + 2 * h) * 32 + n) * 4;   // tile 2s+h = pixels x0 + 4s + 2h, +1: + 512 bytes per s, + 128 for the odd pixel

#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 t[C::NLOAD];
      const int iy = y0 - 1 + d;
      load_row<32, 1>(xb, H, W, iy, gx0, lane, t);
      store_row<32, 1, false>(ring + ((iy + 1) % 3) * C::SLOTB, spill, lane, t);
    }
    // One wave per SIMD: whatever is not an MFMA must slip into the shadow of one, or the pipe idles.  So the row step
    //  * keeps THREE rotating register groups (dy in use / rows arriving / rows just requested): no register moves;
    //  * runs the 24 (tap row, tile pair) groups tap-row-major, so the ring slot of the oldest input row is free after
    //    the first 8 groups and the arriving row is written into it in the middle of the MFMA stream;
    //  * issues the next group's 21 loads one per MFMA group instead of in a burst at the row boundary.
    // group(t) = input row t+2 and dy row t+1, i.e. what output row t+1 adds.
    struct Group {
      f32x4 xrow[C::NLOAD];
      float d0[8], d1[8];
    };
    auto issue_part = [&](int t, Group& f, int part) {       // part 0..7: dy pair s8 = part; parts 8..12: x chunks
      if (part < 8) {
        const bool ok = (t + 1 < y1);                        // dy rows past the range belong to the next wave: read zeros
        const __amdgpu_buffer_rsrc_t as = rsrc(dyb + (long)(ok ? t + 1 : 0) * W * 32, ok ? W * 128 : 0);
        f.d0[part] = bload1(as, aoff + part * 512);
        f.d1[part] = bload1(as, aoff + part * 512 + 128);
      } else {
        const int i = part - 8;
        const int iy = t + 2;
        const bool rowok = (iy >= 0) && (iy < H);
        const __amdgpu_buffer_rsrc_t rs = rsrc(xb + (long)(rowok ? iy : 0) * W * 32, rowok ? W * 128 : 0);
        const int c = lane + 64 * i;
        const int q = c / C::CHUNKS, ch = c % C::CHUNKS;
        f.xrow[i] = bload4(rs, (c < C::NCH) ? ((gx0 + q) * C::PXB + ch * 16) : -16);
      }
    };
    auto step = [&](int yy, const Group& prev, Group& cur, Group& nxt) {
      const char* rb[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) rb[ky] = ring + ((yy + ky) % 3) * C::SLOTB + n * 4;
      float a1[8], a2[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        a1[s8] = 0.5f * (prev.d0[s8] + prev.d1[s8]);
        a2[s8] = 0.5f * (prev.d0[s8] - prev.d1[s8]);
        bsum += prev.d0[s8] + prev.d1[s8];
      }
      // reads run two groups ahead, the input transform one group ahead, each in its own registers: an MFMA must never
      // wait for a VALU result, nor a VALU write for an MFMA that has not read its operand yet
      float dq[3][4], vq[2][4];
      auto rd = [&](int it, float (&d)[4]) {
        const char* p = rb[it >> 3] + (2 * (2 * (it & 7) + h)) * 128;      // ring pixel 2t of tile t = 2s + h
        d[0] = *(const float*)p;
        d[1] = *(const float*)(p + 128);
        d[2] = *(const float*)(p + 256);
        d[3] = *(const float*)(p + 384);
      };
      auto tf = [&](const float (&d)[4], float (&v)[4]) {
        v[0] = d[0] - d[2];
        v[1] = d[1] + d[2];
        v[2] = d[2] - d[1];
        v[3] = d[3] - d[1];
      };
      rd(0, dq[0]);
      rd(1, dq[1]);
      tf(dq[0], vq[0]);
#pragma unroll
      for (int it = 0; it < 24; ++it) {
        const int ky = it >> 3, s8 = it & 7;
        if (it + 2 < 24) rd(it + 2, dq[(it + 2) % 3]);
        if (it + 1 < 24) tf(dq[(it + 1) % 3], vq[(it + 1) & 1]);
        if (it < 13) issue_part(yy + 1, nxt, it);            // next group's loads, one (pair) per MFMA group
        __builtin_amdgcn_sched_barrier(0);
        const float(&v)[4] = vq[it & 1];
        acc[ky][0] = DD_MFMA(prev.d0[s8], v[0], acc[ky][0]);
        acc[ky][1] = DD_MFMA(a1[s8], v[1], acc[ky][1]);
        acc[ky][2] = DD_MFMA(a2[s8], v[2], acc[ky][2]);
        acc[ky][3] = DD_MFMA(prev.d1[s8], v[3], acc[ky][3]);
        __builtin_amdgcn_sched_barrier(0);
        if (it == 15) {   // tap row 0 (the oldest input row) was read by groups 0..7 and tap-row-1 reads (8..15) are out too
          store_row<32, 1, false>(ring + ((yy + 3) % 3) * C::SLOTB, spill, lane, cur.xrow);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    Group ga, gb, gc;
    {   // dy row y0 (group y0-1's dy half) and group y0, synchronously enough: the loop waits for them where it uses them
      const __amdgpu_buffer_rsrc_t as = rsrc(dyb + (long)y0 * W * 32, W * 128);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        ga.d0[s8] = bload1(as, aoff + s8 * 512);
        ga.d1[s8] = bload1(as, aoff + s8 * 512 + 128);
      }
#pragma unroll
      for (int part = 0; part < 13; ++part) issue_part(y0, gb, part);
    }
    __builtin_amdgcn_sched_barrier(0);
    int yy = y0;
    do {                                      // triples of rows; rows past the range see zero dy (no contribution)
      step(yy, ga, gb, gc);
      step(yy + 1, gb, gc, ga);
      step(yy + 2, gc, ga, gb);
      yy += 3;
    } while (yy < y1);
  }

#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[((((long)gw * 3 + ky) * 4 + p) * 16 + r) * 64 + lane] = acc[ky][p][r];
  bpart[(long)gw * 64 + lane] = bsum;
}

// Weight gradient by the 2-D form F(3x3, 2x2): per 2x2 tile of dy and its 4x4 input patch
//   S[u][v] += (G g G^T)[u][v] (x) (B^T d B)[u][v],    dW = A^T S A   (A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]]),
// 16 accumulators of 32x32 over K = tiles: 128 MFMAs per 2 output rows x 32 pixels instead of 192 (1-D) or 288 (direct).
// A = transformed dy tile (lane = output channel, four 4-byte loads per tile straight from HBM), B = transformed patch
// (lane = input channel, sixteen ds_read_b32 per tile from a plain 4-slot ring), k-step = the tile pair (2s, 2s+1) on
// the two half-waves.  One wave per SIMD (256 accumulator registers): the operands of tile s+1 are transformed into
// their own registers while tile s's 16 MFMAs execute, its patch reads are issued a tile earlier still.
constexpr int W2W_STAGE4 = 65 * 64;      // float4s per staging area of the final in-LDS reduction (64 KB of sums + the bias row)
template <int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_wino2_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ part, float* __restrict__ bpart, int B,
                                                             int H, int W, int nstrips) {
  using C = StripCfg<32, 1>;
  constexpr int RINGB = 4 * C::SLOTB + C::SPILLB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + wave * RINGB;
  char* spill = ring + 4 * C::SLOTB;
  const int h = lane >> 5, n = lane & 31;
  const int gw = blockIdx.x * WPB + wave;
  const int HT = (H + 1) / 2;

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  float bsum = 0.f;

  long idx, end;
  wave_range((long)B * nstrips * HT, gw, gridDim.x * WPB, idx, end);
  while (idx < end) {
    const long col = idx / HT;
    const int r0 = (int)(idx - col * HT);
    const int r1 = (int)min((long)HT, r0 + (end - idx));
    idx += r1 - r0;
    const int b = (int)(col / nstrips), x0 = (int)(col % nstrips) * 32;
    const float* xb = x + (long)b * H * W * 32;
    const float* dyb = dy + (long)b * H * W * 32;
    const int gx0 = x0 - 1;
    const int aoff = ((x0 + 2 * h) * 32 + n) * 4;   // tile 2s+h = pixels x0 + 4s + 2h, +1: + 512 bytes per s, + 128 for the odd pixel
    // Descriptors: base = the image, the row in the scalar offset, num_records = the END of that row -- the range check adds the
    // scalar offset to the lane offset (tools/ubench/soffset_probe.hip), so a pixel right of the row is out of range and one
    // left of it (a negative lane offset) too.  One multiply, one add and one select per row instead of a 64-bit base + row *
    // pitch with its 16-bit split.
    const int pitch = W * 128;
    int roff[C::NLOAD];      // this lane's 16-byte chunks of a ring row (the idle lanes of the last group: out of range)
#pragma unroll
    for (int q = 0; q < C::NLOAD; ++q) {
      const int c = lane + 64 * q;
      roff[q] = (c < C::NCH) ? (gx0 + c / C::CHUNKS) * C::PXB + (c % C::CHUNKS) * 16 : -16;
    }
    auto row_loadq = [&](int iy, int q, f32x4& d) {      // chunk group q of input row iy
      const bool ok = (iy >= 0) && (iy < H);
      const int so = ok ? iy * pitch : 0;
      d = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(xb, ok ? so + pitch : 0), roff[q], so, 2));
    };

#pragma unroll
    for (int d = 0; d < 4; ++d) {      // input rows 2*r0 - 1 .. 2*r0 + 2; slot of row iy = (iy + 1) & 3
      f32x4 t[C::NLOAD];
      const int iy = 2 * r0 - 1 + d;
#pragma unroll
      for (int q = 0; q < C::NLOAD; ++q) row_loadq(iy, q, t[q]);
      store_row<32, 1, false>(ring + ((iy + 1) & 3) * C::SLOTB, spill, lane, t);
    }
    // dy rows 2tr, 2tr+1 of the strip: g[a][e][s]
    auto load_dy1 = [&](int tr, int s8, f32x2p (&g)[2][8]) {      // g[e][s] = (row 2tr, row 2tr+1) of the tile's column e: tile s8's four
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int oy = 2 * tr + a;
        const bool ok = (tr < r1) && (oy < H);       // rows past the range belong to the next wave: read zeros
        const int so = ok ? oy * pitch : 0;
        const __amdgpu_buffer_rsrc_t as = rsrc(dyb, ok ? so + pitch : 0);
        g[0][s8][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(as, aoff + s8 * 512, so, 2));
        g[1][s8][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(as, aoff + s8 * 512 + 128, so, 2));
      }
    };
    f32x2p ga[2][8], gb[2][8];      // dy of the current / next tile-row, alternating (no register moves)
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) load_dy1(r0, s8, ga);

    auto step = [&](int tr, const f32x2p (&gc)[2][8], f32x2p (&gn)[2][8]) {
      f32x4 pre[2][C::NLOAD];

      const char* rb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rb[r] = ring + ((2 * tr + r) & 3) * C::SLOTB + n * 4;      // slot of input row 2*tr - 1 + r
      float dq[2][4][4];          // patches of tile s+1 / s+2 (reads in flight)
      float aq[2][16], bq[2][16];   // transformed operands of tile s / s+1
      auto rd = [&](int s8, float (&d)[4][4]) {
        const int poff = (2 * (2 * s8 + h)) * 128;      // ring pixel 2t of tile t = 2s + h
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) d[r][c] = *(const float*)(rb[r] + poff + c * 128);
      };
      auto tf = [&](int s8, const float (&d)[4][4], float (&av)[16], float (&bv)[16]) {
        // packed fp32 throughout (the vector ALUs are the matrix ALUs: every instruction here is matrix time lost).
        // x stage on register pairs (c0,c1) / (c2,c3) with a broadcast operand, y stage pairwise over c.
        f32x2p wl[4], wh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x2p lo = {d[r][0], d[r][1]}, hi = {d[r][2], d[r][3]};
          asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(wl[r]) : "v"(lo), "v"(hi));                // d0 - d2, d1 + d2
          asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(wh[r]) : "v"(hi), "v"(lo));  // d2 - d1, d3 - d1
        }
        const f32x2p bl[4] = {pk_sub2(wl[0], wl[2]), wl[1] + wl[2], pk_sub2(wl[2], wl[1]), pk_sub2(wl[3], wl[1])};
        const f32x2p bh[4] = {pk_sub2(wh[0], wh[2]), wh[1] + wh[2], pk_sub2(wh[2], wh[1]), pk_sub2(wh[3], wh[1])};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bv[u * 4 + 0] = bl[u].x;
          bv[u * 4 + 1] = bl[u].y;
          bv[u * 4 + 2] = bh[u].x;
          bv[u * 4 + 3] = bh[u].y;
        }
        // dy tile (pairs over the tile's two rows), WITHOUT the halves of G = [[1,0],[.5,.5],[.5,-.5],[0,1]]: position
        // (u,v) carries the factor (1,2,2,1)[u] * (1,2,2,1)[v], taken out again (exactly) in conv_wino2_wgrad_reduce_b.
        const f32x2p g0 = gc[0][s8], g1 = gc[1][s8];
        const f32x2p gr[4] = {g0, g0 + g1, pk_sub2(g0, g1), g1};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          f32x2p sd;      // (row0 + row1, row0 - row1)
          asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(sd) : "v"(gr[v]));
          av[0 * 4 + v] = gr[v].x;
          av[1 * 4 + v] = sd.x;
          av[2 * 4 + v] = sd.y;
          av[3 * 4 + v] = gr[v].y;
        }
        bsum += av[5];                              // (1,1) position = g00 + g01 + g10 + g11: the tile's bias contribution
      };
      rd(0, dq[0]);
      rd(1, dq[1]);
      tf(0, dq[0], aq[0], bq[0]);
      // Per tile: a block of vector work (the next tile's transform), then the 16 MFMAs with everything that is not a vector
      // instruction issued behind them -- the LDS reads of the tile after next, this tile-row's share of the global loads for the
      // next one, the ring writes.  A memory instruction costs the wave 8-40 issue cycles inside the vector block and nothing
      // behind an MFMA (64 cycles of matrix pipe each; tools/ubench/mfma_issue.hip), and every switch between vector and matrix
      // instructions costs ~5 more.
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        __builtin_amdgcn_sched_barrier(0);
        if (s8 + 1 < 8) tf(s8 + 1, dq[(s8 + 1) & 1], aq[(s8 + 1) & 1], bq[(s8 + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s8 + 2 < 8) rd(s8 + 2, dq[s8 & 1]);                      // dq[s8&1] held tile s8: already transformed
        load_dy1(tr + 1, s8, gn);
#pragma unroll
        for (int q = 2 * s8; q < 2 * s8 + 2; ++q)      // the ring rows of the next tile-row: out by tile 2, written to LDS in tiles 6 and 7
          if (q < C::NLOAD) {
            row_loadq(2 * tr + 3, q, pre[0][q]);
            row_loadq(2 * tr + 4, q, pre[1][q]);
          }
        if (s8 == 6) store_row<32, 1, false>(ring + ((2 * tr + 4) & 3) * C::SLOTB, spill, lane, pre[0]);      // every patch read of this
        if (s8 == 7) store_row<32, 1, false>(ring + ((2 * tr + 5) & 3) * C::SLOTB, spill, lane, pre[1]);      // tile-row is out (tile 7's: s8 = 5)
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p] = DD_MFMA(aq[s8 & 1][p], bq[s8 & 1][p], acc[p]);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);      // SALU
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int tr = r0;
    do {      // pairs of tile-rows; a trailing odd one sees zero dy (no contribution)
      step(tr, ga, gb);
      step(tr + 1, gb, ga);
      tr += 2;
    } while (tr < r1);
  }

  // The workgroup's four partials are added in LDS, ((w0 + w1) + (w2 + w3)), before anything goes to HBM: 16 MB of
  // partials instead of 67 (the reduce kernel reads them beside the overlapped optimizer pass, at a fraction of the
  // HBM rate).  Two 65 KB staging areas over the rings, which nobody reads any more.  The accumulators only ever flow OUT of
  // their registers: waves 1 and 3 store theirs, waves 0 and 2 add theirs on top with LDS float adds (one contributor per
  // address and phase: w1 + w0 is the same number whoever arrives first, so the result stays bit-reproducible), then all 256
  // threads add the two areas and write the block's partial.  (Reading partial sums back INTO the 256 accumulators under
  // wave-uniform branches, as this epilogue first did, made the register allocator spill 48 registers to scratch.)
  static_assert(WPB == 4, "conv_wino2_wgrad: the in-LDS reduction is written for four waves");
  const int lane_e = dd_fresh_lane();                          // the epilogue's own: no lane-derived register crosses the tile loop
  float* stage = (float*)((f32x4*)smem + (wave >> 1) * W2W_STAGE4 + lane_e);      // [chunk = 4p + r/4][lane] float4, + one row for the bias sums
  __syncthreads();
  if (wave & 1) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *(f32x4*)(stage + (4 * p + c) * 256) = f32x4{acc[p][4 * c], acc[p][4 * c + 1], acc[p][4 * c + 2], acc[p][4 * c + 3]};
    *(f32x4*)(stage + 64 * 256) = f32x4{bsum, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  if (!(wave & 1)) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __hip_atomic_fetch_add(stage + (4 * p + (r >> 2)) * 256 + (r & 3), acc[p][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(stage + 64 * 256, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  {
    const f32x4* sa = (const f32x4*)smem;
    const f32x4* sb = sa + W2W_STAGE4;
    const int tid_e = lane_e + 64 * wave;
    float* out = part + (long)blockIdx.x * 256 * 64;
    for (int i = tid_e; i < 64 * 64; i += WPB * 64) {      // float4 i = chunk * 64 + lane holds rows 4 chunk .. + 3 of lane's column
      const f32x4 a = sa[i], b = sb[i];
      const int chunk = i >> 6, ln = i & 63;
#pragma unroll
      for (int k = 0; k < 4; ++k) out[(chunk * 4 + k) * 64 + ln] = a[k] + b[k];
    }
    if (tid_e < 64) bpart[(long)blockIdx.x * 64 + tid_e] = sa[64 * 64 + tid_e].x + sb[64 * 64 + tid_e].x;
  }
}

// Second stage of conv_wino2_wgrad, in two launches (one block per accumulator row would leave 17 blocks to read 64 MB):
//  a) 256 blocks = (position p, chunk c of the waves): fixed-order sum over the chunk's partials.  A wave's partial of
//     one position is 16 rows x 64 lanes = 4 KB contiguous, read as 256 float4 (the first version read one 256-byte
//     row per wave, 64 KB apart: 0.57 TB/s) -> tsum[c][p][r][64]
//  b) one block per register row r (+ one for the bias): sum of the 16 chunks, output transform A^T S A, scatter to OIHW.
constexpr int W2R_CHUNKS = 16;
// 256-thread blocks (and 64-thread ones in reduce_b): these run on a side stream beside resident kernels that hold most wave
// slots of every CU -- a 1024-thread block then waits for a whole CU's worth of them (measured: 1.3 ms in the queue).
__global__ __launch_bounds__(256) void conv_wino2_wgrad_reduce_a(const float* __restrict__ part, float* __restrict__ tsum, int nw) {
  const int f = threadIdx.x;
  const int p = blockIdx.x & 15, c = blockIdx.x >> 4;
  const int per = (nw + W2R_CHUNKS - 1) / W2R_CHUNKS;
  const int w0 = c * per, w1 = min(nw, w0 + per);
  const f32x4* src = (const f32x4*)part + (long)p * 256 + f;
  f32x4 s[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int w = w0;
  for (; w + 3 < w1; w += 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] += src[(long)(w + i) * 4096];
  }
  for (int i = 0; w < w1; ++w, ++i) s[i] += src[(long)w * 4096];
  ((f32x4*)tsum)[((long)c * 16 + p) * 256 + f] = (s[0] + s[1]) + (s[2] + s[3]);
}

// The output transform runs as running sums with coefficients 0 / +-1 in the order of the closed forms ((t0 + t1) + t2, t1 - t2,
// (t1 + t2) + t3), one column v at a time.  (Rounds 3-4 kept this kernel at 40 registers with every loop rolled so that it fitted beside
// the c2 data gradient's 464; that kernel now holds 475 and nothing fits beside it, and since round 5 the weight gradient this launch
// reduces is the LAST conv kernel of the step, so the launch is the step's tail: its loads are batched per column instead.)
__global__ __launch_bounds__(64) void conv_wino2_wgrad_reduce_b(const float* __restrict__ tsum, const float* __restrict__ bpart,
                                                                float* __restrict__ dw, float* __restrict__ db, int nw) {
  const int l = threadIdx.x;
  const int r = blockIdx.x;   // accumulator register row, or 16 for the bias
  if (r == 16) {
    // s[i] = the partials w = i, i + 4, ... in order, as the four-way loop this replaces -- with 64 loads in flight instead of 4 (the
    // bias block was the long pole of the launch: nw / 4 dependent round trips, 34 us at the tail of the step)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int full = nw & ~3;
#pragma unroll
    for (int i = 0; i < 4; ++i) dd_sum_strided(s[i], bpart + (long)i * 64 + l, 256, full / 4);
    for (int w = full, i = 0; w < nw; ++w, ++i) s[i] += bpart[(long)w * 64 + l];
    const float t = (s[0] + s[1]) + (s[2] + s[3]);
    const float other = __shfl_xor(t, 32);
    if (l < 32) db[l] = t + other;
    return;
  }
  float out[3][3];
#pragma unroll
  for (int a = 0; a < 9; ++a) out[a / 3][a % 3] = 0.f;
#pragma unroll 1
  for (int v = 0; v < 4; ++v) {
    float z[3] = {0.f, 0.f, 0.f};      // z[ky] = A^T[ky][u] t[u][v]
    // the 4 x 16 chunk sums of this column are requested together (one memory round trip per column instead of four); each t[u] is still
    // the sum of its 16 chunks in chunk order
    float tv[4][W2R_CHUNKS];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < W2R_CHUNKS; ++c) tv[u][c] = tsum[(((long)c * 16 + (u * 4 + v)) * 16 + r) * 64 + l];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < W2R_CHUNKS; ++c) t += tv[u][c];
      // the halves of G left out of conv_wino2_wgrad's dy transform (powers of two: exact)
      t = t * ((u == 1 || u == 2) ? 0.5f : 1.f) * ((v == 1 || v == 2) ? 0.5f : 1.f);
      z[0] += (u < 3 ? 1.f : 0.f) * t;
      z[1] += (u == 1 ? 1.f : u == 2 ? -1.f : 0.f) * t;
      z[2] += (u > 0 ? 1.f : 0.f) * t;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {      // dW[ky][kx] = z[ky][v] A[v][kx]
      out[ky][0] += (v < 3 ? 1.f : 0.f) * z[ky];
      out[ky][1] += (v == 1 ? 1.f : v == 2 ? -1.f : 0.f) * z[ky];
      out[ky][2] += (v > 0 ? 1.f : 0.f) * z[ky];
    }
  }
  const int o = dd_acc_row(r, l), jj = l & 31;
  float* dst = dw + ((long)o * 32 + jj) * 9;
#pragma unroll
  for (int a = 0; a < 9; ++a) dst[a] = out[a / 3][a % 3];
}

// Second stage: fixed-order sums of the per-wave partials S_p (one block per (ky, register) row), then the output
// transform and the scatter to OIHW.  Block 48 reduces the bias partials.
__global__ __launch_bounds__(1024) void conv_wino_wgrad_reduce(const float* __restrict__ part, const float* __restrict__ bpart,
                                                               float* __restrict__ dw, float* __restrict__ db, int nw) {
  constexpr int G = 16;
  __shared__ float red[4][G][64];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int row = blockIdx.x;   // ky*16 + r, or 48 for the bias
  if (row == 48) {
    float s0 = 0.f;
    for (int w = g; w < nw; w += G) s0 += bpart[(long)w * 64 + l];
    red[0][g][l] = s0;
    __syncthreads();
    if (g != 0) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) s += red[0][i][l];
    const float other = __shfl_xor(s, 32);
    if (l < 32) db[l] = s + other;
    return;
  }
  const int ky = row >> 4, r = row & 15;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int w = g; w < nw; w += G) {
#pragma unroll
    for (int p = 0; p < 4; ++p) s[p] += part[((((long)w * 3 + ky) * 4 + p) * 16 + r) * 64 + l];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) red[p][g][l] = s[p];
  __syncthreads();
  if (g != 0) return;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int i = 0; i < G; ++i) t[p] += red[p][i][l];
  const int o = dd_acc_row(r, l), j = l & 31;
  float* out = dw + ((long)o * 32 + j) * 9 + ky * 3;
  out[0] = (t[0] + t[1]) + t[2];
  out[1] = t[1] - t[2];
  out[2] = (t[1] + t[2]) + t[3];
}

// U image for conv_wino_fwd: packed[((((ky*4 + p)*2 + half)*2 + g)*64 + lane)*4 + j] = u_p of the taps
// Weff[co = (lane&15) + 16*half][ci = 8*(lane>>4) + 4g + j][ky][0..2];  kind 0: Weff = W (forward);
// kind 1: Weff[co][ci][ky][kx] = W[ci][co][2-ky][2-kx] (stride-1 data gradient).
__global__ void conv_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int kind) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= WINO_UFLOATS) return;
  const int j = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 1, hf = (idx >> 9) & 1, pos = (idx >> 10) & 3, ky = idx >> 12;
  const int co = (lane & 15) + 16 * hf, ci = 8 * (lane >> 4) + 4 * g + j;
  float t[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    t[kx] = kind == 0 ? w[((long)co * 32 + ci) * 9 + ky * 3 + kx] : w[((long)ci * 32 + co) * 9 + (2 - ky) * 3 + (2 - kx)];
  float u;
  if (pos == 0) u = t[0];
  else if (pos == 1) u = 0.5f * ((t[0] + t[1]) + t[2]);
  else if (pos == 2) u = 0.5f * ((t[0] - t[1]) + t[2]);
  else u = t[2];
  p[idx] = u;
}

// Sign words of a 32-channel NHWC activation: bits[p] bit c = x[p][c] > 0 -- what dd_conv_fwd_relu_bits writes beside its output, for an
// activation another kernel produced (the strip mosaic in front of SpatialMappingCNN's out_conv, when that layer's data gradient runs on
// the Winograd kernels).  A wave takes 64 pixels: each ballot covers two (lanes = channels, fully coalesced 256-byte loads), lane p keeps
// the word of pixel p, the 64 words leave as one store.
__global__ __launch_bounds__(256) void relu_sign_bits_kernel(const float* __restrict__ x, unsigned* __restrict__ bits, long npix) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long p0 = wave * 64; p0 < npix; p0 += nwaves * 64) {
    unsigned word = 0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const long p = p0 + 2 * i + (lane >> 5);
      const float v = p < npix ? x[p * 32 + (lane & 31)] : 0.f;
      const unsigned long long m = __ballot(v > 0.f);
      if (lane == 2 * i) word = (unsigned)m;
      if (lane == 2 * i + 1) word = (unsigned)(m >> 32);
    }
    if (p0 + lane < npix) bits[p0 + lane] = word;
  }
}

// out_pad [B, H + 2, W + 2, 32] = dy [B, H, W, 32] * (the ReLU was open: bit c of bits_pad[pixel]) in the interior, ZERO on the border ring:
// the output gradient of a padding-0 3x3 layer laid out for the padding-1 kernels (its border outputs do not exist, so they carry no
// gradient), with the layer's own ReLU backward applied from the sign words its forward wrote.
__global__ __launch_bounds__(256) void relu_bwd_pad_bits_kernel(const float* __restrict__ dy, const unsigned* __restrict__ bits_pad,
                                                                f32x4* __restrict__ out_pad, int B, int H, int W, int dy_cstore, int dy_coff) {
  const long total = (long)B * (H + 2) * (W + 2) * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 7);
    const long pp = i >> 3;
    const int xx = (int)(pp % (W + 2)), yy = (int)((pp / (W + 2)) % (H + 2));
    const long b = pp / ((long)(W + 2) * (H + 2));
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (xx >= 1 && xx <= W && yy >= 1 && yy <= H) {
      const f32x4 g = *(const f32x4*)(dy + ((b * H + (yy - 1)) * W + (xx - 1)) * dy_cstore + dy_coff + 4 * q);
      const unsigned m = bits_pad[pp] >> (4 * q);
      o.x = (m & 1u) ? g.x : 0.f;
      o.y = (m & 2u) ? g.y : 0.f;
      o.z = (m & 4u) ? g.z : 0.f;
      o.w = (m & 8u) ? g.w : 0.f;
    }
    __builtin_nontemporal_store(o, out_pad + i);
  }
}

__global__ void relu_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ y, f32x4* __restrict__ out,
                                long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g = dy[i], v = y[i];
    f32x4 o;
    o.x = v.x > 0.f ? g.x : 0.f;
    o.y = v.y > 0.f ? g.y : 0.f;
    o.z = v.z > 0.f ? g.z : 0.f;
    o.w = v.w > 0.f ? g.w : 0.f;
    out[i] = o;
  }
}

template <typename K>
int allow_lds(K kernel, size_t bytes) {
  static thread_local const void* done[48];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == (const void*)kernel) return 0;
  if (ndone < 48) done[ndone++] = (const void*)kernel;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : dd_fail(DD_ERR_LAUNCH, "hipFuncSetAttribute(%zu bytes LDS): %s", bytes, hipGetErrorString(e));
}

int check_desc(const dd_conv_desc* d) {
  DD_REQUIRE(d != nullptr, DD_ERR_BAD_ARG, "conv: NULL descriptor");
  DD_REQUIRE(d->batch > 0 && d->height > 0 && d->width > 0, DD_ERR_BAD_ARG, "conv: non-positive size");
  DD_REQUIRE(d->ksize == 3 && d->pad == 1, DD_ERR_UNSUPPORTED, "conv: only k3 p1 is implemented (got k%d p%d)", d->ksize, d->pad);
  DD_REQUIRE(d->stride == 1 || d->stride == 2, DD_ERR_UNSUPPORTED, "conv: stride %d", d->stride);
  DD_REQUIRE(d->cout == 32, DD_ERR_UNSUPPORTED, "conv: Cout %d (only 32)", d->cout);
  DD_REQUIRE((d->cin_real == 32 && d->cin_store == 32) || (d->cin_real == 3 && d->cin_store == 4), DD_ERR_UNSUPPORTED,
             "conv: Cin %d stored as %d (supported: 32/32, 3/4)", d->cin_real, d->cin_store);
  DD_REQUIRE(!(d->cin_real == 3 && d->stride == 2), DD_ERR_UNSUPPORTED, "conv: Cin 3 with stride 2");
  DD_REQUIRE((long)d->width * 128 < (1L << 31), DD_ERR_UNSUPPORTED, "conv: row too long for a 32-bit buffer descriptor");
  return 0;
}

// Workgroups launched = workgroups resident: `per_cu` blocks on each CU (bounded by the LDS ring + registers
// of the instantiation), fewer when the problem has fewer wave-sized pieces.  rows_per_task (a testing / tuning
// knob, never changes results) asks for at least that many rows per wave, i.e. fewer and longer ranges.
int resident_grid(const dd_conv_desc* d, long row_tiles, int wpb, int per_cu) {
  long blocks = (long)dd_cu_budget_internal() * per_cu;
  if (d->rows_per_task > 0) blocks = min(blocks, max(1L, row_tiles / d->rows_per_task / wpb));
  return (int)max(1L, min(blocks, (row_tiles + wpb - 1) / wpb));
}

template <int CIN, int S, int EPI, int WPB, bool AFF = false>
int launch_fwd(const float* x, const float* wp, const float* bias, const float* msk, float* y, const dd_conv_desc* d,
               hipStream_t st, unsigned* bits_out = nullptr, const float* aff = nullptr, float* stats = nullptr) {
  using C = StripCfg<CIN, S>;
  const int Ho = dd_conv_out(d->height, S), Wo = dd_conv_out(d->width, S);
  const int nstrips = (Wo + 31) / 32;
  const size_t lds = C::WFLOATS * 4 + (size_t)WPB * C::WAVEB;
  const int per_cu = (int)max((size_t)1, min((size_t)2, (size_t)(160 * 1024) / lds));
  const int grid = resident_grid(d, (long)d->batch * nstrips * Ho, WPB, per_cu);
  auto k = conv_strip_fwd<CIN, S, EPI, WPB, AFF>;
  if (int rc = allow_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, x, wp, bias, msk, y, bits_out, d->batch, d->height, d->width,
                     Ho, Wo, nstrips, aff, stats);
  DD_LAUNCH_CHECK("conv_strip_fwd");
  return 0;
}

template <int EPI>
int launch_wino2(const float* x, const float* up, const float* bias, const unsigned* bits_in, float* y, unsigned* bits_out,
                 const dd_conv_desc* d, hipStream_t st, const float* x4 = nullptr, float* w1part = nullptr, int* nw_out = nullptr) {
  using C = StripCfg<32, 1>;
  using C4 = StripCfg<4, 1>;
  constexpr int WPB = 4;
  const int nstrips = (d->width + 31) / 32;
  // Register-row kernels by default.  DD_WINO2_RING=1: the LDS-ring form everywhere, DD_WINO2_RING_W1=1: for the fused data
  // gradient only (A/B; DESIGN.md 3.1c: beside the optimizer pass the choice depends on how that pass is launched -- csrc/dense.hip).
  static const bool ring_env = getenv("DD_WINO2_RING") != nullptr;
  static const bool ring_w1 = getenv("DD_WINO2_RING_W1") != nullptr;
  const bool ring = ring_env || (EPI == EPI_RELU_BITS_W1 && ring_w1);
  DD_REQUIRE((long)d->height * d->width * 128 < (1L << 30), DD_ERR_UNSUPPORTED, "conv_wino2: image of %d x %d pixels: the kernel addresses an image with 30-bit offsets",
             d->height, d->width);
  const size_t lds = (size_t)WINO2_UFLOATS * 4 + (ring ? (size_t)WPB * (4 * C::SLOTB + C::SPILLB) : 0) +
                     (EPI == EPI_RELU_BITS_W1 ? (size_t)WPB * (4 * C4::SLOTB + C4::SPILLB) + 128 : 0);
  const int grid = resident_grid(d, (long)d->batch * nstrips * ((d->height + 1) / 2), WPB, 1);
  if (nw_out) *nw_out = grid * WPB;
  if (int rc = ring ? allow_lds(conv_wino2_fwd<EPI, WPB>, lds) : allow_lds(conv_wino2r_fwd<EPI, WPB>, lds)) return rc;
  // rows of the next tile-row are pulled into L2 a tile-row ahead (in the step: forward 1.195 -> 1.163 ms against no prefetch)
  constexpr int pf_rows = 1;
  auto kring = conv_wino2_fwd<EPI, WPB>;
  auto kreg = conv_wino2r_fwd<EPI, WPB>;
  if (ring) hipLaunchKernelGGL(kring, dim3(grid), dim3(WPB * 64), lds, st, x, up, bias, bits_in, y, bits_out, d->batch, d->height, d->width, nstrips, x4, w1part);
  else hipLaunchKernelGGL(kreg, dim3(grid), dim3(WPB * 64), lds, st, x, up, bias, bits_in, y, bits_out, d->batch, d->height, d->width, nstrips, x4, w1part, pf_rows);
  DD_LAUNCH_CHECK("conv_wino2_fwd");
  return 0;
}

template <int EPI>
int launch_wino(const float* x, const float* up, const float* bias, const unsigned* bits_in, float* y, unsigned* bits_out,
                const dd_conv_desc* d, hipStream_t st) {
  using C = StripCfg<32, 1>;
  constexpr int WPB = 8;
  const int nstrips = (d->width + 31) / 32;
  const size_t lds = (size_t)WINO_UFLOATS * 4 + (size_t)WPB * C::WAVEB;
  const int grid = resident_grid(d, (long)d->batch * nstrips * d->height, WPB, 1);
  auto k = conv_wino_fwd<EPI, WPB>;
  if (int rc = allow_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, x, up, bias, bits_in, y, bits_out, d->batch, d->height, d->width, nstrips);
  DD_LAUNCH_CHECK("conv_wino_fwd");
  return 0;
}

}  // namespace

extern "C" {

int64_t dd_conv_packed_floats(const dd_conv_desc* d, int32_t kind) {
  if (check_desc(d)) return -1;
  if (kind != 0 && d->cin_real != 32) return -1;
  return d->cin_real == 32 ? 36 * 64 * 4 : 5 * 64 * 4;
}

int dd_conv_pack(const float* w_oihw, float* packed, const dd_conv_desc* d, int32_t kind, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(w_oihw && packed, DD_ERR_BAD_ARG, "conv_pack: NULL pointer");
  DD_REQUIRE(kind >= 0 && kind <= 2, DD_ERR_BAD_ARG, "conv_pack: kind %d", kind);
  DD_REQUIRE(kind == 0 || d->cin_real == 32, DD_ERR_UNSUPPORTED, "conv_pack: dgrad packing needs Cin 32");
  const int n = d->cin_real == 32 ? 36 * 64 * 4 : 5 * 64 * 4;
  hipLaunchKernelGGL(conv_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, packed,
                     d->cin_real, kind);
  DD_LAUNCH_CHECK("conv_pack");
  return 0;
}

int dd_conv_fwd(const float* x, const float* packed_fwd, const float* bias, const float* mask, float* y,
                const dd_conv_desc* d, int32_t epilogue, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && packed_fwd && y, DD_ERR_BAD_ARG, "conv_fwd: NULL pointer");
  DD_REQUIRE(epilogue != DD_EPI_RELU_MASK || mask, DD_ERR_BAD_ARG, "conv_fwd: RELU_MASK epilogue needs a mask");
  DD_REQUIRE((epilogue != DD_EPI_BIAS && epilogue != DD_EPI_BIAS_RELU) || bias, DD_ERR_BAD_ARG, "conv_fwd: bias epilogue needs a bias");
  hipStream_t st = (hipStream_t)stream;
#define DD_DISPATCH(CIN, S, WPB)                                                                                    \
  switch (epilogue) {                                                                                               \
    case DD_EPI_NONE: return launch_fwd<CIN, S, DD_EPI_NONE, WPB>(x, packed_fwd, bias, mask, y, d, st);             \
    case DD_EPI_BIAS: return launch_fwd<CIN, S, DD_EPI_BIAS, WPB>(x, packed_fwd, bias, mask, y, d, st);             \
    case DD_EPI_BIAS_RELU: return launch_fwd<CIN, S, DD_EPI_BIAS_RELU, WPB>(x, packed_fwd, bias, mask, y, d, st);   \
    case DD_EPI_RELU_MASK: return launch_fwd<CIN, S, DD_EPI_RELU_MASK, WPB>(x, packed_fwd, bias, mask, y, d, st);   \
    default: return dd_fail(DD_ERR_BAD_ARG, "conv_fwd: epilogue %d", epilogue);                                     \
  }
  if (d->cin_store == 4) { DD_DISPATCH(4, 1, 8) }
  if (d->stride == 1) { DD_DISPATCH(32, 1, 8) }
  DD_DISPATCH(32, 2, 4)
#undef DD_DISPATCH
}

int dd_conv_fwd_relu_bits(const float* x, const float* packed_fwd, const float* bias, float* y, uint32_t* relu_bits,
                          const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && packed_fwd && bias && y && relu_bits, DD_ERR_BAD_ARG, "conv_fwd_relu_bits: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  if (d->cin_store == 4) return launch_fwd<4, 1, EPI_BIAS_RELU_BITS, 8>(x, packed_fwd, bias, nullptr, y, d, st, relu_bits);
  if (d->stride == 1) return launch_fwd<32, 1, EPI_BIAS_RELU_BITS, 8>(x, packed_fwd, bias, nullptr, y, d, st, relu_bits);
  return launch_fwd<32, 2, EPI_BIAS_RELU_BITS, 4>(x, packed_fwd, bias, nullptr, y, d, st, relu_bits);
}

static int conv_dgrad_impl(const float* dy, const float* packed_dgrad, const float* relu_src, int mask_mode, float* dx,
                           const dd_conv_desc* d, void* stream, const float* aff = nullptr);

int dd_conv_dgrad(const float* dy, const float* packed_dgrad, const float* relu_src, float* dx, const dd_conv_desc* d,
                  void* stream) {
  return conv_dgrad_impl(dy, packed_dgrad, relu_src, relu_src ? 1 : 0, dx, d, stream);
}

int dd_conv_dgrad_relu_bits(const float* dy, const float* packed_dgrad, const uint32_t* relu_bits, float* dx,
                            const dd_conv_desc* d, void* stream) {
  DD_REQUIRE(relu_bits, DD_ERR_BAD_ARG, "conv_dgrad_relu_bits: NULL mask");
  return conv_dgrad_impl(dy, packed_dgrad, (const float*)relu_bits, 2, dx, d, stream);
}

int dd_conv_dgrad_bn(const float* dy, const float* packed_dgrad, const float* pre_bn, const float* affine, float* dx,
                     const dd_conv_desc* d, void* stream) {
  DD_REQUIRE(pre_bn && affine, DD_ERR_BAD_ARG, "conv_dgrad_bn: NULL mask / affine");
  return conv_dgrad_impl(dy, packed_dgrad, pre_bn, 3, dx, d, stream, affine);
}

static int conv_dgrad_impl(const float* dy, const float* packed_dgrad, const float* relu_src, int mask_mode, float* dx,
                           const dd_conv_desc* d, void* stream, const float* aff) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(dy && packed_dgrad && dx, DD_ERR_BAD_ARG, "conv_dgrad: NULL pointer");
  DD_REQUIRE(d->cin_real == 32, DD_ERR_UNSUPPORTED, "conv_dgrad: Cin %d (the first layer has no data gradient)", d->cin_real);
  hipStream_t st = (hipStream_t)stream;
  if (d->stride == 1) {
    // a stride-1 k3 p1 data gradient IS a k3 p1 convolution of dy with the flipped / transposed weights
    if (mask_mode == 3) return launch_fwd<32, 1, EPI_RELU_MASK_AFF, 8>(dy, packed_dgrad, nullptr, relu_src, dx, d, st, nullptr, aff);
    if (mask_mode == 2) return launch_fwd<32, 1, EPI_RELU_BITS, 8>(dy, packed_dgrad, nullptr, relu_src, dx, d, st);
    return mask_mode ? launch_fwd<32, 1, DD_EPI_RELU_MASK, 8>(dy, packed_dgrad, nullptr, relu_src, dx, d, st)
                     : launch_fwd<32, 1, DD_EPI_NONE, 8>(dy, packed_dgrad, nullptr, nullptr, dx, d, st);
  }
  using C = StripCfg<32, 1>;
  constexpr int WPB = 8;
  const int H = d->height, W = d->width, Ho = dd_conv_out(H, 2), Wo = dd_conv_out(W, 2);
  const int ns = (W + 1) / 2, nr = (H + 1) / 2;
  const int nstrips = (ns + 31) / 32;
  const size_t lds = C::WFLOATS * 4 + (size_t)WPB * C::WAVEB;
  const int grid = resident_grid(d, (long)d->batch * nstrips * nr, WPB, 1);
  if (mask_mode == 3) {
    auto k = conv_s2_dgrad<WPB, 3>;
    if (int rc = allow_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, dy, packed_dgrad, relu_src, dx, d->batch, H, W, Ho, Wo,
                       nstrips, aff);
  } else if (mask_mode == 2) {
    auto k = conv_s2_dgrad<WPB, 2>;
    if (int rc = allow_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, dy, packed_dgrad, relu_src, dx, d->batch, H, W, Ho, Wo,
                       nstrips, aff);
  } else if (mask_mode == 1) {
    auto k = conv_s2_dgrad<WPB, 1>;
    if (int rc = allow_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, dy, packed_dgrad, relu_src, dx, d->batch, H, W, Ho, Wo,
                       nstrips, aff);
  } else {
    auto k = conv_s2_dgrad<WPB, 0>;
    if (int rc = allow_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, dy, packed_dgrad, relu_src, dx, d->batch, H, W, Ho, Wo,
                       nstrips, aff);
  }
  DD_LAUNCH_CHECK("conv_s2_dgrad");
  return 0;
}

int64_t dd_conv_wgrad_workspace_bytes(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  const int nt = d->cin_real == 32 ? 9 : 1;
  const int64_t waves = 4 * (int64_t)DD_NUM_CU * 2;   // upper bound of resident waves of any instantiation
  return waves * ((int64_t)nt * 1024 + 64) * 4;
}

static int conv_wgrad_impl(const float* x, const float* aff, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                           int64_t workspace_bytes, const dd_conv_desc* d, void* stream);

int dd_conv_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                  int64_t workspace_bytes, const dd_conv_desc* d, void* stream) {
  return conv_wgrad_impl(x, nullptr, dy, dw_oihw, dbias, workspace, workspace_bytes, d, stream);
}

int dd_conv_wgrad_bn(const float* pre_bn, const float* affine, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                     int64_t workspace_bytes, const dd_conv_desc* d, void* stream) {
  DD_REQUIRE(affine, DD_ERR_BAD_ARG, "conv_wgrad_bn: NULL affine");
  DD_REQUIRE(d && d->cin_real == 32, DD_ERR_UNSUPPORTED, "conv_wgrad_bn: the input of a BN'd layer has 32 channels");
  return conv_wgrad_impl(pre_bn, affine, dy, dw_oihw, dbias, workspace, workspace_bytes, d, stream);
}

int64_t dd_conv_stats_floats(void) { return (int64_t)DD_NUM_CU * 2 * 8 * 64 * 2; }

int dd_conv_fwd_stats(const float* x, const float* packed_fwd, const float* bias, const float* in_affine, float* u,
                      float* stats, const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && packed_fwd && bias && u && stats, DD_ERR_BAD_ARG, "conv_fwd_stats: NULL pointer");
  DD_REQUIRE(!(in_affine && d->cin_store == 4), DD_ERR_UNSUPPORTED, "conv_fwd_stats: the image layer has no BN'd input");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, dd_conv_stats_floats() * sizeof(float), st) != hipSuccess)
    return dd_fail(DD_ERR_LAUNCH, "conv_fwd_stats: hipMemsetAsync failed");
  if (d->cin_store == 4) return launch_fwd<4, 1, EPI_BIAS_STATS, 8>(x, packed_fwd, bias, nullptr, u, d, st, nullptr, nullptr, stats);
  if (d->stride == 1)
    return in_affine ? launch_fwd<32, 1, EPI_BIAS_STATS, 8, true>(x, packed_fwd, bias, nullptr, u, d, st, nullptr, in_affine, stats)
                     : launch_fwd<32, 1, EPI_BIAS_STATS, 8, false>(x, packed_fwd, bias, nullptr, u, d, st, nullptr, nullptr, stats);
  return in_affine ? launch_fwd<32, 2, EPI_BIAS_STATS, 4, true>(x, packed_fwd, bias, nullptr, u, d, st, nullptr, in_affine, stats)
                   : launch_fwd<32, 2, EPI_BIAS_STATS, 4, false>(x, packed_fwd, bias, nullptr, u, d, st, nullptr, nullptr, stats);
}

static int conv_wgrad_impl(const float* x, const float* aff, const float* dy, float* dw_oihw, float* dbias, void* workspace,
                           int64_t workspace_bytes, const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && dy && dw_oihw && dbias && workspace, DD_ERR_BAD_ARG, "conv_wgrad: NULL pointer");
  DD_REQUIRE(workspace_bytes >= dd_conv_wgrad_workspace_bytes(d), DD_ERR_WORKSPACE, "conv_wgrad: workspace %ld < %ld bytes",
             (long)workspace_bytes, (long)dd_conv_wgrad_workspace_bytes(d));
  hipStream_t st = (hipStream_t)stream;
  constexpr int WPB = 4;
  const int S = d->stride, H = d->height, W = d->width, Ho = dd_conv_out(H, S), Wo = dd_conv_out(W, S);
  const int nstrips = (Wo + 31) / 32;
  const int nt = d->cin_real == 32 ? 9 : 1;
  // Cin 32: 144 accumulator registers -> one wave per SIMD -> one 4-wave block per CU; Cin 3: two
  const int grid = resident_grid(d, (long)d->batch * nstrips * Ho, WPB, d->cin_store == 4 ? 2 : 1);
  const int nw = grid * WPB;
  float* part = (float*)workspace;
  float* bpart = part + (size_t)nw * nt * 1024;
#define DD_WG(CIN, SS, AFF)                                                                                         \
  {                                                                                                                 \
    auto k = conv_wgrad<CIN, SS, WPB, AFF>;                                                                         \
    const size_t lds = (size_t)WPB * StripCfg<CIN, SS>::WAVEB;                                                      \
    if (int rc = allow_lds(k, lds)) return rc;                                                                      \
    hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, x, dy, part, bpart, d->batch, H, W, Ho, Wo, nstrips, \
                       aff);                                                                                        \
  }
  if (d->cin_store == 4) DD_WG(4, 1, false)
  else if (S == 1 && aff) DD_WG(32, 1, true)
  else if (S == 1) DD_WG(32, 1, false)
  else if (aff) DD_WG(32, 2, true)
  else DD_WG(32, 2, false)
#undef DD_WG
  DD_LAUNCH_CHECK("conv_wgrad");
  if (d->cin_store == 4)
    hipLaunchKernelGGL(conv_wgrad_reduce<4>, dim3(nt * 16 + 1), dim3(1024), 0, st, part, bpart, dw_oihw, dbias, nw);
  else
    hipLaunchKernelGGL(conv_wgrad_reduce<32>, dim3(nt * 16 + 1), dim3(1024), 0, st, part, bpart, dw_oihw, dbias, nw);
  DD_LAUNCH_CHECK("conv_wgrad_reduce");
  return 0;
}

int dd_relu_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream) {
  DD_REQUIRE(dy && y && out && n > 0, DD_ERR_BAD_ARG, "relu_bwd: bad argument");
  DD_REQUIRE(n % 4 == 0, DD_ERR_UNSUPPORTED, "relu_bwd: n %% 4 != 0");
  const long n4 = n / 4;
  const int grid = (int)min((n4 + 255) / 256, (long)DD_NUM_CU * 8);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dy, (const f32x4*)y,
                     (f32x4*)out, n4);
  DD_LAUNCH_CHECK("relu_bwd");
  return 0;
}

int dd_relu_sign_bits(const float* x, uint32_t* bits, int64_t npix, void* stream) {
  DD_REQUIRE(x && bits && npix > 0, DD_ERR_BAD_ARG, "relu_sign_bits: bad argument");
  const long waves = (npix + 63) / 64;
  const int grid = (int)min((waves + 3) / 4, (long)DD_NUM_CU * 8);
  hipLaunchKernelGGL(relu_sign_bits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bits, (long)npix);
  DD_LAUNCH_CHECK("relu_sign_bits");
  return 0;
}

int dd_relu_bwd_pad_bits(const float* dy, const uint32_t* bits_pad, float* out_pad, int32_t batch, int32_t h, int32_t w, int32_t dy_cstore,
                         int32_t dy_coff, void* stream) {
  DD_REQUIRE(dy && bits_pad && out_pad && batch > 0 && h > 0 && w > 0, DD_ERR_BAD_ARG, "relu_bwd_pad_bits: bad argument");
  DD_REQUIRE(dy_cstore >= 32 && dy_cstore % 4 == 0 && dy_coff >= 0 && dy_coff % 4 == 0 && dy_coff + 32 <= dy_cstore, DD_ERR_BAD_ARG,
             "relu_bwd_pad_bits: dy's 32 channels must be a 4-aligned slice of its %d stored ones (offset %d)", dy_cstore, dy_coff);
  DD_REQUIRE((((uintptr_t)dy | (uintptr_t)out_pad) & 15) == 0, DD_ERR_BAD_ARG, "relu_bwd_pad_bits: dy and out_pad must be 16-byte aligned");
  const long total = (long)batch * (h + 2) * (w + 2) * 8;
  const int grid = (int)min((total + 255) / 256, (long)DD_NUM_CU * 8);
  hipLaunchKernelGGL(relu_bwd_pad_bits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, bits_pad, (f32x4*)out_pad,
                     batch, h, w, dy_cstore, dy_coff);
  DD_LAUNCH_CHECK("relu_bwd_pad_bits");
  return 0;
}

// ---- Winograd F(2,3) path of the 32 -> 32 stride-1 layer ----
int64_t dd_conv_wino_packed_floats(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  if (d->cin_real != 32 || d->stride != 1) return -1;
  return WINO_UFLOATS;
}

int dd_conv_wino_pack(const float* w_oihw, float* packed, const dd_conv_desc* d, int32_t kind, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(w_oihw && packed, DD_ERR_BAD_ARG, "conv_wino_pack: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  DD_REQUIRE(kind == 0 || kind == 1, DD_ERR_BAD_ARG, "conv_wino_pack: kind %d (0 forward, 1 data gradient)", kind);
  hipLaunchKernelGGL(conv_wino_pack_kernel, dim3((WINO_UFLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, kind);
  DD_LAUNCH_CHECK("conv_wino_pack");
  return 0;
}

int64_t dd_conv_wino2_packed_floats(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  if (d->cin_real != 32 || d->stride != 1) return -1;
  return WINO2_UFLOATS;
}

int dd_conv_wino2_pack(const float* w_oihw, float* packed, const dd_conv_desc* d, int32_t kind, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(w_oihw && packed, DD_ERR_BAD_ARG, "conv_wino2_pack: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  DD_REQUIRE(kind == 0 || kind == 1, DD_ERR_BAD_ARG, "conv_wino2_pack: kind %d (0 forward, 1 data gradient)", kind);
  hipLaunchKernelGGL(conv_wino2_pack_kernel, dim3((WINO2_UFLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, kind);
  DD_LAUNCH_CHECK("conv_wino2_pack");
  return 0;
}

int dd_conv_wino2_fwd_relu_bits(const float* x, const float* packed, const float* bias, float* y, uint32_t* relu_bits,
                                const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && packed && bias && y && relu_bits, DD_ERR_BAD_ARG, "conv_wino2_fwd_relu_bits: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  return launch_wino2<EPI_BIAS_RELU_BITS>(x, packed, bias, nullptr, y, relu_bits, d, (hipStream_t)stream);
}

int dd_conv_wino2_dgrad_relu_bits(const float* dy, const float* packed, const uint32_t* relu_bits, float* dx,
                                  const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(dy && packed && relu_bits && dx, DD_ERR_BAD_ARG, "conv_wino2_dgrad_relu_bits: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  return launch_wino2<EPI_RELU_BITS>(dy, packed, nullptr, relu_bits, dx, nullptr, d, (hipStream_t)stream);
}

int64_t dd_conv_wino2_dgrad_w1_workspace_bytes(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  if (d->cin_real != 32 || d->stride != 1) return -1;
  return (int64_t)4 * DD_NUM_CU * 1024 * 4;      // one 32 x 32 partial per wave
}

int dd_conv_wino2_dgrad_w1(const float* dy, const float* packed, const uint32_t* relu_bits, const float* x_nhwc4, float* dw1_oihw,
                           float* dbias1, void* workspace, int64_t workspace_bytes, const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(dy && packed && relu_bits && x_nhwc4 && dw1_oihw && dbias1 && workspace, DD_ERR_BAD_ARG, "conv_wino2_dgrad_w1: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  DD_REQUIRE(workspace_bytes >= dd_conv_wino2_dgrad_w1_workspace_bytes(d), DD_ERR_WORKSPACE, "conv_wino2_dgrad_w1: workspace %ld < %ld bytes",
             (long)workspace_bytes, (long)dd_conv_wino2_dgrad_w1_workspace_bytes(d));
  int nw = 0;
  if (int rc = launch_wino2<EPI_RELU_BITS_W1>(dy, packed, nullptr, relu_bits, nullptr, nullptr, d, (hipStream_t)stream, x_nhwc4,
                                              (float*)workspace, &nw))
    return rc;
  hipLaunchKernelGGL(conv_w1_reduce, dim3(128), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw1_oihw, dbias1, nw);
  DD_LAUNCH_CHECK("conv_w1_reduce");
  return 0;
}

int64_t dd_conv_wino_wgrad_workspace_bytes(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  if (d->cin_real != 32 || d->stride != 1) return -1;
  return (int64_t)4 * DD_NUM_CU * ((int64_t)12 * 1024 + 64) * 4;      // one 4-wave workgroup per CU, 12 accumulators per wave
}

int64_t dd_conv_wino2_wgrad_workspace_bytes(const dd_conv_desc* d) {
  if (check_desc(d)) return -1;
  if (d->cin_real != 32 || d->stride != 1) return -1;
  return ((int64_t)4 * DD_NUM_CU * ((int64_t)16 * 1024 + 64) + 16 * 16 * 1024) * 4;      // per-wave partials (16 accumulators) + 16 chunk sums
}

namespace {
int wino2_wgrad_grid(const dd_conv_desc* d) {
  return resident_grid(d, (long)d->batch * ((d->width + 31) / 32) * ((d->height + 1) / 2), 4, 1);
}
}  // namespace

int dd_conv_wino2_wgrad_partials(const float* x, const float* dy, void* workspace, int64_t workspace_bytes, const dd_conv_desc* d,
                                 void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && dy && workspace, DD_ERR_BAD_ARG, "conv_wino2_wgrad: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  DD_REQUIRE(workspace_bytes >= dd_conv_wino2_wgrad_workspace_bytes(d), DD_ERR_WORKSPACE, "conv_wino2_wgrad: workspace %ld < %ld bytes",
             (long)workspace_bytes, (long)dd_conv_wino2_wgrad_workspace_bytes(d));
  DD_REQUIRE((long)d->height * d->width * 128 < (1L << 30), DD_ERR_UNSUPPORTED, "conv_wino2_wgrad: image of %d x %d pixels: the kernel addresses an image with 30-bit offsets",
             d->height, d->width);
  constexpr int WPB = 4;
  const int nstrips = (d->width + 31) / 32;
  const int grid = wino2_wgrad_grid(d);      // one partial per WORKGROUP (its four waves are added in LDS)
  float* part = (float*)workspace;
  float* bpart = part + (size_t)grid * 16 * 1024;
  auto k = conv_wino2_wgrad<WPB>;
  const size_t lds = max((size_t)WPB * (4 * StripCfg<32, 1>::SLOTB + StripCfg<32, 1>::SPILLB), (size_t)2 * W2W_STAGE4 * 16);
  if (int rc = allow_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, (hipStream_t)stream, x, dy, part, bpart, d->batch, d->height, d->width, nstrips);
  DD_LAUNCH_CHECK("conv_wino2_wgrad");
  return 0;
}

int dd_conv_wino2_wgrad_finish(void* workspace, int64_t workspace_bytes, float* dw_oihw, float* dbias, const dd_conv_desc* d,
                               void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(workspace && dw_oihw && dbias, DD_ERR_BAD_ARG, "conv_wino2_wgrad_finish: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  DD_REQUIRE(workspace_bytes >= dd_conv_wino2_wgrad_workspace_bytes(d), DD_ERR_WORKSPACE, "conv_wino2_wgrad_finish: workspace %ld < %ld bytes",
             (long)workspace_bytes, (long)dd_conv_wino2_wgrad_workspace_bytes(d));
  hipStream_t st = (hipStream_t)stream;
  const int nw = wino2_wgrad_grid(d);
  float* part = (float*)workspace;
  float* bpart = part + (size_t)nw * 16 * 1024;
  float* tsum = bpart + (size_t)nw * 64;
  hipLaunchKernelGGL(conv_wino2_wgrad_reduce_a, dim3(256), dim3(256), 0, st, part, tsum, nw);
  DD_LAUNCH_CHECK("conv_wino2_wgrad_reduce_a");
  hipLaunchKernelGGL(conv_wino2_wgrad_reduce_b, dim3(17), dim3(64), 0, st, tsum, bpart, dw_oihw, dbias, nw);
  DD_LAUNCH_CHECK("conv_wino2_wgrad_reduce_b");
  return 0;
}

int dd_conv_wino2_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, void* workspace, int64_t workspace_bytes,
                        const dd_conv_desc* d, void* stream) {
  DD_REQUIRE(dw_oihw && dbias, DD_ERR_BAD_ARG, "conv_wino2_wgrad: NULL pointer");
  if (int rc = dd_conv_wino2_wgrad_partials(x, dy, workspace, workspace_bytes, d, stream)) return rc;
  return dd_conv_wino2_wgrad_finish(workspace, workspace_bytes, dw_oihw, dbias, d, stream);
}

int dd_conv_wino_wgrad(const float* x, const float* dy, float* dw_oihw, float* dbias, void* workspace, int64_t workspace_bytes,
                       const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && dy && dw_oihw && dbias && workspace, DD_ERR_BAD_ARG, "conv_wino_wgrad: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  DD_REQUIRE(workspace_bytes >= dd_conv_wino_wgrad_workspace_bytes(d), DD_ERR_WORKSPACE, "conv_wino_wgrad: workspace %ld < %ld bytes",
             (long)workspace_bytes, (long)dd_conv_wino_wgrad_workspace_bytes(d));
  hipStream_t st = (hipStream_t)stream;
  constexpr int WPB = 4;
  const int nstrips = (d->width + 31) / 32;
  const int grid = resident_grid(d, (long)d->batch * nstrips * d->height, WPB, 1);
  const int nw = grid * WPB;
  float* part = (float*)workspace;
  float* bpart = part + (size_t)nw * 12 * 1024;
  auto k = conv_wino_wgrad<WPB>;
  const size_t lds = (size_t)WPB * StripCfg<32, 1>::WAVEB;
  if (int rc = allow_lds(k, lds)) return rc;
  hipLaunchKernelGGL(k, dim3(grid), dim3(WPB * 64), lds, st, x, dy, part, bpart, d->batch, d->height, d->width, nstrips);
  DD_LAUNCH_CHECK("conv_wino_wgrad");
  hipLaunchKernelGGL(conv_wino_wgrad_reduce, dim3(49), dim3(1024), 0, st, part, bpart, dw_oihw, dbias, nw);
  DD_LAUNCH_CHECK("conv_wino_wgrad_reduce");
  return 0;
}

int dd_conv_wino_fwd_relu_bits(const float* x, const float* packed, const float* bias, float* y, uint32_t* relu_bits,
                               const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(x && packed && bias && y && relu_bits, DD_ERR_BAD_ARG, "conv_wino_fwd_relu_bits: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  return launch_wino<EPI_BIAS_RELU_BITS>(x, packed, bias, nullptr, y, relu_bits, d, (hipStream_t)stream);
}

int dd_conv_wino_dgrad_relu_bits(const float* dy, const float* packed, const uint32_t* relu_bits, float* dx,
                                 const dd_conv_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  DD_REQUIRE(dy && packed && relu_bits && dx, DD_ERR_BAD_ARG, "conv_wino_dgrad_relu_bits: NULL pointer");
  DD_REQUIRE(d->cin_real == 32 && d->stride == 1, DD_ERR_UNSUPPORTED, "conv_wino: only the 32 -> 32 stride-1 layer");
  return launch_wino<EPI_RELU_BITS>(dy, packed, nullptr, relu_bits, dx, nullptr, d, (hipStream_t)stream);
}

}  // extern "C"
