// Forward of the dilated stride-1 ConvTranspose2d layers (spatial_bb/components.py:135-136), "input-aligned" form.
//
// The flipped-tap gather of dconv.hip makes every OUTPUT pixel visit all k x k taps; with out = in + d(k-1) a quarter of
// those (pixel, tap) pairs read the zero border (298^2 x 49 visits for 256^2 x 49 products: x1.36; tap skipping per
// 64-pixel wave tile and 4-row group recovers little because EVERY tile of a 43-wide residue class is near a border).
// Here the m-tiles are aligned to the INPUT row instead: for tap column kx the 32 input pixels [32t, 32t + 32) feed the 32
// output pixels [32t + s(kx), ...), s(kx) = pad_w - d*kx -- one ACCUMULATOR TILE PER (m-tile, tap column), and no border
// zero is ever multiplied:
//
//   * a workgroup (8 waves) owns one output row oy of one image x one 32-channel column tile: the k_valid <= k input rows
//     oy - pad + d*ky of an 8-channel chunk sit in LDS (double buffered: row ky of chunk q + 1 is fetched while tap row ky of
//     chunk q is multiplied); tap rows outside the image are skipped for the whole workgroup -- exact in y as well;
//   * the row's (m-tile, kx) accumulator tiles are dealt evenly to the 8 waves (up_conv_1: 8 m-tiles x 7 = 7 per wave, one
//     m-tile each; up_conv_2: 10 x 7 = 70 -> 9 per wave); per (chunk, tap row, tile) one ds_read_b128 and one lane-linear
//     16-byte weight load at a scalar offset (the image dd_dconv_pack writes) feed 4 MFMAs; both are requested one whole tap
//     row ahead, into the registers the tile's MFMAs have just read;
//   * epilogue: the k partial rows are added into one LDS row image at their shifts (k barrier-separated passes: a fixed
//     order, deterministic), then bias / ReLU and 16-byte stores.
//
// Tasks are dealt to the XCDs in (image, residue class, phase row, column tile) order, so the workgroups resident on one XCD
// walk neighbouring rows of ONE class and share their input rows (k users each) in that XCD's L2.
#include <stdlib.h>

#include "dd_common.h"

namespace {

constexpr int TF_THREADS = 512;

template <int K, int IWP>
struct TfGeom {
  static constexpr int ROWF = IWP * 8;                    // floats of a patch row (IWP pixels x 8 channels)
  static constexpr int BUFF = K * ROWF;                   // floats of one buffer
  static constexpr int NPR = (IWP * 2 + TF_THREADS - 1) / TF_THREADS;      // 16-byte pieces per thread and patch row
  static constexpr int OPITCH = 40;                       // floats per pixel of the output row image (32 channels + 8: lanes 32..63
                                                          // of an accumulator tile sit 4 pixels further = 160 floats = 32 banks on)
  static constexpr int OWMAX = BUFF / OPITCH;             // widest output row (+ 1 spare pixel) the image (one buffer) holds
};

// NSLOT accumulator tiles per wave; ONE_MT: the tiles of a wave are (m-tile = wave, kx = slot) (needs n_mt <= 8, NSLOT == K)
template <int K, int D, int NSLOT, bool ONE_MT, int IWP>
__global__ __launch_bounds__(TF_THREADS) void dconv_tfwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                const dd_gconv_desc d, int epi, int wp_bytes) {
  using G = TfGeom<K, IWP>;
  __shared__ __attribute__((aligned(16))) float lds[2][G::BUFF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, n = lane & 31;
  const int NC = d.cin >> 3, NTC = (d.cout + 31) >> 5;
  const int rows_max = (d.out_h + D - 1) / D;
  const int n_mt = (d.in_w + 31) >> 5;
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  const __amdgpu_buffer_rsrc_t ws = dd_rsrc(wp, wp_bytes);

  // fill pieces of this thread inside a patch row: piece p = tid + 512*i -> pixel p >> 1, channel half p & 1; pixels past
  // in_w get an offset the range check rejects whatever row offset is added: they read zeros, so the accumulator rows of
  // the ragged last m-tile add zeros to the output row image
  int poff[G::NPR];
#pragma unroll
  for (int i = 0; i < G::NPR; ++i) {
    const int p = tid + TF_THREADS * i;
    poff[i] = (p >> 1) < d.in_w ? ((p >> 1) * d.in_cstore + d.in_coff + 4 * (p & 1)) * 4 : (int)0xC0000000;
  }
  // this wave's tiles: linear index L = m-tile * K + kx
  int s_mt[NSLOT], s_kx[NSLOT];
  bool s_ok[NSLOT];
  int aoff[NSLOT];                                       // byte offset of this lane's A fragment of the slot in a patch row
  int ioff[NSLOT];                                       // float index of accumulator element 0 of the slot in the output row image
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int L = ONE_MT ? wave * K + i : wave * NSLOT + i;
    s_ok[i] = L < n_mt * K;
    s_mt[i] = s_ok[i] ? L / K : 0;
    s_kx[i] = s_ok[i] ? L - s_mt[i] * K : 0;
    aoff[i] = ((s_mt[i] * 32 + n) * 8 + 4 * h) * 4;
    ioff[i] = (s_mt[i] * 32 + 4 * h) * G::OPITCH + n;
  }

  // ---- tasks of this workgroup: XCD = blockIdx % 8 owns the x-th eighth of the (image, residue, phase row, column tile) list
  const int per_x = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_img = D * rows_max * NTC;
  const long len = (long)d.batch * per_img;
  const long seg0 = len * xcd / 8, seg1 = len * (xcd + 1) / 8;

  for (long t = seg0 + slot; t < seg1; t += per_x) {
    const int b = (int)(t / per_img);
    int rem = (int)(t - (long)b * per_img);
    const int nt = rem % NTC;
    rem /= NTC;
    const int r = rem / rows_max, jy = rem - r * rows_max;
    const int oy = r + D * jy;
    if (oy >= d.out_h) continue;
    const int ry = oy - d.pad_h;
    const int ky0 = ry >= 0 ? 0 : (-ry + D - 1) / D;
    const int ky1 = min(K - 1, (d.in_h - 1 - ry) >= 0 ? (d.in_h - 1 - ry) / D : -1);
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);

    f32x16 acc[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    if (ky1 >= ky0) {
      // ---- chunk 0 of this task (not overlapped: once per ~300k cycles of multiplying)
      for (int ky = ky0; ky <= ky1; ++ky) {
        const int rowoff = (ry + D * ky) * d.in_w * d.in_cstore * 4;
#pragma unroll
        for (int i = 0; i < G::NPR; ++i) {
          const int p = tid + TF_THREADS * i;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, poff[i] + rowoff, 0, 0));
          if (p < IWP * 2) *(f32x4*)&lds[0][ky * G::ROWF + p * 4] = v;
        }
      }
      // weight fragments of the first tap row (the ring then runs one tap row ahead, across chunk boundaries)
      constexpr int AR = 3;                      // general form: A fragments are requested AR tiles ahead
      static_assert(ONE_MT || NSLOT % AR == 0, "A ring");
      f32x4 Bf[NSLOT], Af[ONE_MT ? 1 : AR];
      auto bload = [&](int i, int q, int ky) {
        const int soff = (((q * K + ky) * K + s_kx[i]) * NTC + nt) * 1024;
        Bf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, lane * 16, soff, 0));
      };
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) bload(i, 0, ky0);
      __syncthreads();
      int par = 0;
      for (int q = 0; q < NC; ++q) {
        const bool more = q + 1 < NC;
        const char* lbase = (const char*)&lds[par][0];
        float* nbuf = &lds[par ^ 1][0];
        // A fragments of the chunk's first tap row
        if constexpr (ONE_MT) {
          Af[0] = *(const f32x4*)(lbase + aoff[0] + ky0 * (G::ROWF * 4));
        } else {
#pragma unroll
          for (int i = 0; i < AR; ++i) Af[i] = *(const f32x4*)(lbase + aoff[i] + ky0 * (G::ROWF * 4));
        }
        for (int ky = ky0; ky <= ky1; ++ky) {
          const bool lastk = ky == ky1;
          // the tap row whose weights are requested now: the next one of this chunk, the first one of the next chunk, or (at
          // the very end) this one again -- a harmless reload, so that the loads stay unconditional
          const int qn = lastk ? (more ? q + 1 : q) : q;
          const int kyn = lastk ? (more ? ky0 : ky) : ky + 1;
          const int kya = lastk ? ky : ky + 1;               // A comes from THIS buffer: past the last tap row, re-read it
          f32x4 stage[G::NPR];
          if (more) {
            const int rowoff = (ry + D * ky) * d.in_w * d.in_cstore * 4;
#pragma unroll
            for (int i = 0; i < G::NPR; ++i)
              stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, poff[i] + rowoff, 32 * (q + 1), 0));
          }
          if constexpr (ONE_MT) {
            const f32x4 An = *(const f32x4*)(lbase + aoff[0] + kya * (G::ROWF * 4));
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
              acc[i] = DD_MFMA(Af[0].x, Bf[i].x, acc[i]);
              acc[i] = DD_MFMA(Af[0].y, Bf[i].y, acc[i]);
              acc[i] = DD_MFMA(Af[0].z, Bf[i].z, acc[i]);
              acc[i] = DD_MFMA(Af[0].w, Bf[i].w, acc[i]);
              bload(i, qn, kyn);                           // in flight for the NSLOT - 1 tiles until this slot comes round again
              __builtin_amdgcn_sched_barrier(0);
            }
            Af[0] = An;
          } else {
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
              acc[i] = DD_MFMA(Af[i % AR].x, Bf[i].x, acc[i]);
              acc[i] = DD_MFMA(Af[i % AR].y, Bf[i].y, acc[i]);
              acc[i] = DD_MFMA(Af[i % AR].z, Bf[i].z, acc[i]);
              acc[i] = DD_MFMA(Af[i % AR].w, Bf[i].w, acc[i]);
              bload(i, qn, kyn);
              Af[i % AR] = *(const f32x4*)(lbase + aoff[(i + AR) % NSLOT] + (i + AR < NSLOT ? ky : kya) * (G::ROWF * 4));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (more) {
#pragma unroll
            for (int i = 0; i < G::NPR; ++i) {
              const int p = tid + TF_THREADS * i;
              if (p < IWP * 2) *(f32x4*)&nbuf[ky * G::ROWF + p * 4] = stage[i];
            }
          }
        }
        __syncthreads();
        par ^= 1;
      }
    }

    // ---- epilogue: the K partial rows are added into one output row image at their shifts, pass kx = tap column kx
    float* img = &lds[0][0];
    for (int i = tid; i < (n_mt * 32 + D * (K - 1)) * (G::OPITCH / 4); i += TF_THREADS) *(f32x4*)&img[i * 4] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < K; ++pass) {
      const int shift = D * (K - 1 - pass);                // = pad_w - D*pass (the launcher checked pad_w)
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        if (ONE_MT ? (i == pass && s_ok[i]) : (s_ok[i] && s_kx[i] == pass)) {
          float* p0 = img + ioff[i];
#pragma unroll
          for (int e = 0; e < 16; ++e) p0[((e & 3) + 8 * (e >> 2) + shift) * G::OPITCH] += acc[i][e];
        }
      }
      __syncthreads();
    }
    // write-out: 8 lanes per pixel, 16 bytes each
    const int base = ((oy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff;
    for (int i = tid; i < d.out_w * 8; i += TF_THREADS) {
      const int px = i >> 3, c = 32 * nt + 4 * (i & 7);
      if (c < d.cout) {
        f32x4 v = *(const f32x4*)&img[px * G::OPITCH + 4 * (i & 7)];
        if (epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) v += *(const f32x4*)&bias[c];
        if (epi == DD_EPI_BIAS_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ys,
                                               (base + px * d.out_cstore + c) * 4, 0, 0);
      }
    }
    __syncthreads();
  }
}

}  // namespace

// Launches the input-aligned forward if the layer is one it is built for; returns false (nothing launched) otherwise.
bool dd_dconv_tfwd_launch(const float* x, const float* packed, const float* bias, float* y, const dd_gconv_desc* d, int epilogue,
                          int wp_bytes, hipStream_t st) {
  static const bool off = getenv("DD_DCONV_TFWD_OFF") && atoi(getenv("DD_DCONV_TFWD_OFF")) != 0;
  if (off) return false;
  if (epilogue == DD_EPI_RELU_MASK) return false;
  if (d->kh != d->kw || d->dil_h != d->dil_w) return false;
  const int k = d->kh, dl = d->dil_h;
  if (d->pad_h != dl * (k - 1) || d->pad_w != dl * (k - 1)) return false;                    // the full transposed form only
  if (d->out_h < d->in_h + dl * (k - 1) || d->out_w < d->in_w + dl * (k - 1)) return false;  // every partial lands inside the row
  if (d->cin % 8 || d->cout < 32 || d->cout % 4 || d->out_coff % 4 || d->out_cstore % 4) return false;
  if (((uintptr_t)bias & 15) != 0) return false;
  const int grid = dd_cu_budget_internal() & ~7;
  if (grid < 8) return false;
  const int n_mt = (d->in_w + 31) / 32;
#define DD_TF(KK, DD_, NS, ONE, IWP_)                                                                                          \
  do {                                                                                                                         \
    using G = TfGeom<KK, IWP_>;                                                                                                \
    if (d->in_w > IWP_ || n_mt * 32 + DD_ * (KK - 1) > G::OWMAX || d->out_w > G::OWMAX || n_mt * KK > 8 * NS || (ONE && n_mt > 8)) return false;                    \
    hipLaunchKernelGGL((dconv_tfwd_kernel<KK, DD_, NS, ONE, IWP_>), dim3(grid), dim3(TF_THREADS), 0, st, x, packed, bias, y,  \
                       *d, epilogue, wp_bytes);                                                                                \
    return true;                                                                                                               \
  } while (0)
  if (k == 7 && dl == 7 && d->in_w <= 256) DD_TF(7, 7, 7, true, 256);
  if (k == 7 && dl == 7 && d->in_w <= 320) DD_TF(7, 7, 9, false, 320);
#undef DD_TF
  return false;
}
