// Forward of the dilated stride-1 ConvTranspose2d layers (spatial_bb/components.py:135-137), "input-aligned" form.
//
// The flipped-tap gather of dconv.hip makes every OUTPUT pixel visit all k x k taps; with out = in + d(k-1) a quarter of
// those (pixel, tap) pairs read the zero border (298^2 x 49 visits for 256^2 x 49 products: x1.36; tap skipping per
// 64-pixel wave tile and 4-row group recovers little because EVERY tile of a 43-wide residue class is near a border).
// Here the m-tiles are aligned to the INPUT row instead: for tap column kx the input pixels [Wt, Wt + W) feed the output
// pixels [Wt + s(kx), ...), s(kx) = pad_w - d*kx -- one ACCUMULATOR TILE PER (m-tile, tap column), and no border zero is
// ever multiplied:
//
//   * a workgroup (8 waves) owns one output row oy of one image x one column tile of output channels: the k_valid <= k
//     input rows oy - pad + d*ky of an 8-channel chunk sit in LDS (double buffered: row ky of chunk q + 1 -- after the last
//     chunk, of the workgroup's NEXT row -- is fetched while tap row ky of chunk q is multiplied); tap rows outside the image
//     are skipped for the whole workgroup: exact in y as well;
//   * the row's (m-tile, kx) accumulator tiles are dealt evenly to the 8 waves; per (chunk, tap row, tile) one LDS read and
//     one lane-linear weight load at a scalar offset (the images dd_dconv_pack writes) feed the tile's MFMAs; the weight
//     fragment is requested one whole tap row ahead, into the registers the tile's MFMAs have just read;
//   * epilogue: the k partial rows are added into one LDS row image at their shifts (k barrier-separated passes: a fixed
//     order, deterministic), then bias / ReLU and 16-byte stores.
//
// Two forms (MODE), both for Cout > 16:
//   0  at most 8 m-tiles of 32 pixels (up_conv_1, 256 wide): wave w owns m-tile w and its k tap columns -- one A fragment
//      per (chunk, tap row) feeds 4k v_mfma_f32_32x32x2_f32;
//   1  wider rows (up_conv_2, 298 wide -> 10 m-tiles x 7 = 70 tiles, 9 per wave).
// (Cout <= 16 -- up_conv_3 -- stays on dconv_fwd_kernel: a 16 x 16-tile form of this kernel measured 2.24 ms against 1.98.)
//
// Tasks are dealt to the XCDs in (image, residue class, phase row, column tile) order, so the workgroups resident on one XCD
// walk neighbouring rows of ONE class and share their input rows (k users each) in that XCD's L2.
#include <stdlib.h>

#include "dd_common.h"

namespace {

constexpr int TF_THREADS = 512;

template <int K, int D, int MODE, int IWP>
struct TfGeom {
  static constexpr int TW = 32;                           // pixels of an m-tile
  static constexpr int NE = 16;                           // accumulator registers of a tile
  static constexpr int ROWF = IWP * 8;                    // floats of a patch row (IWP pixels x 8 channels)
  static constexpr int BUFF = K * ROWF;                   // floats of one buffer
  static constexpr int NPR = (IWP * 2 + TF_THREADS - 1) / TF_THREADS;      // 16-byte pieces per thread and patch row
  // floats per pixel of the output row image: lanes 32..63 of a tile sit 4 pixels = 160 floats = 32 banks further
  static constexpr int P = 40;
  static constexpr int HALO = D * (K - 1);
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding vector-memory operation
// (vmcnt(0)): here that would drain the weight fragments requested a tap row ahead at every chunk boundary -- a full L2
// round trip with all 8 waves of the workgroup idle (measured: ~3k cycles per chunk, 12 % of up_conv_1's forward).  Nothing this
// kernel reads from global memory is written by it, so only the LDS patch / row image need the ordering.
__device__ __forceinline__ void tf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


struct TfTask {
  int b, nt, oy, ry, ky0, ky1;
};

template <int K, int D, int NSLOT, int MODE, int IWP>
__global__ __launch_bounds__(TF_THREADS) void dconv_tfwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                const dd_gconv_desc d, int epi, int wp_bytes, int dbg_repeat) {
  using G = TfGeom<K, D, MODE, IWP>;
  constexpr bool ONE_MT = MODE == 0;
  constexpr int TW = G::TW, NE = G::NE, P = G::P;
  using frag = f32x4;
  using acc_t = f32x16;
  __shared__ __attribute__((aligned(16))) float lds[2][G::BUFF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = d.cin >> 3, NTC = (d.cout + 31) >> 5;
  const int rows_max = (d.out_h + D - 1) / D;
  const int n_mt = (d.in_w + TW - 1) / TW;
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
  int s_kx[NSLOT];
  bool s_ok[NSLOT];
  int aoff[NSLOT];                                       // byte offset of this lane's A fragment of the slot in a patch row
  int s_mt[NSLOT];                                       // (wave-uniform: scalar registers)
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int L = ONE_MT ? wave * K + i : wave * NSLOT + i;
    s_ok[i] = L < n_mt * K;
    const int mt = s_ok[i] ? L / K : 0;
    s_kx[i] = s_ok[i] ? L - mt * K : 0;
    s_mt[i] = mt;
    aoff[i] = ((mt * 32 + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4;
  }
  // float index of this lane's accumulator element 0 of m-tile 0 in the output row image
  const int ioff_lane = 4 * (lane >> 5) * P + (lane & 31);

  // ---- tasks of this workgroup: XCD = blockIdx % 8 owns the x-th eighth of the (image, residue, phase row, column tile) list
  const int per_x = gridDim.x >> 3;      // the launchers round the grid down to a multiple of 8 (one segment per XCD)
  const int xcd = blockIdx.x & 7;
  const int per_img = D * rows_max * NTC;
  const long len = (long)d.batch * per_img;
  const long seg1 = len * (xcd + 1) / 8;
  auto decode = [&](long t, TfTask& k) -> bool {      // false: a phase row past the last output row of its residue class
    k.b = (int)(t / per_img);
    int rem = (int)(t - (long)k.b * per_img);
    k.nt = rem % NTC;
    rem /= NTC;
    const int r = rem / rows_max, jy = rem - r * rows_max;
    k.oy = r + D * jy;
    k.ry = k.oy - d.pad_h;
    k.ky0 = k.ry >= 0 ? 0 : (-k.ry + D - 1) / D;
    k.ky1 = min(K - 1, (d.in_h - 1 - k.ry) >= 0 ? (d.in_h - 1 - k.ry) / D : -1);
    return k.oy < d.out_h;
  };
  auto next_task = [&](long t, TfTask& k) -> long {   // first valid task at or after t in this workgroup's sequence; seg1 if none
    while (t < seg1 && !decode(t, k)) t += per_x;
    return t < seg1 ? t : seg1;
  };

  frag Bf[NSLOT];
  auto bload = [&](int i, int q, int ky, int nt) {
    const int soff = (((q * K + ky) * K + s_kx[i]) * NTC + nt) * 1024;
    Bf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, lane * 16, soff, 0));
  };

  TfTask cur, nxt;
  long t = next_task(len * xcd / 8 + (blockIdx.x >> 3), cur);
  bool prefetched = false;                             // chunk 0 of `cur` (and its first weight fragments) already requested
  int par = 0;
  while (t < seg1) {
    const long tn = next_task(t + per_x, nxt);
    const bool have_next = tn < seg1;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)cur.b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const __amdgpu_buffer_rsrc_t xn = dd_rsrc(x + (long)nxt.b * d.in_h * d.in_w * d.in_cstore, have_next ? in_bytes : 0);
    const int ky0 = cur.ky0, ky1 = cur.ky1, ry = cur.ry;

    acc_t acc[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[i][e] = 0.f;

    bool fetched_next = false;
    if (ky1 >= ky0) {
      if (!prefetched) {
        // ---- chunk 0 of this task, all rows requested before the first is stored (first task of the workgroup only)
        f32x4 v[K][G::NPR];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const int rowoff = (ky >= ky0 && ky <= ky1) ? (ry + D * ky) * d.in_w * d.in_cstore * 4 : (int)0xC0000000;
#pragma unroll
          for (int i = 0; i < G::NPR; ++i) v[ky][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, (int)((unsigned)poff[i] + (unsigned)rowoff), 0, 0));
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
          for (int i = 0; i < G::NPR; ++i) {
            const int p = tid + TF_THREADS * i;
            if (p < IWP * 2) *(f32x4*)&lds[par][ky * G::ROWF + p * 4] = v[ky][i];
          }
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) bload(i, 0, ky0, cur.nt);
      }
      tf_barrier();
      constexpr int AR = ONE_MT ? 1 : 3;                    // A fragments are requested AR tiles ahead
      static_assert(ONE_MT || NSLOT % AR == 0, "A ring");
      frag Af[AR];
      for (int q = 0; q < NC; ++q) {
        const bool more = q + 1 < NC;
        const char* lbase = (const char*)&lds[par][0];
        float* nbuf = &lds[par ^ 1][0];
#pragma unroll
        for (int i = 0; i < AR; ++i) Af[i] = *(const frag*)(lbase + aoff[i] + ky0 * (G::ROWF * 4));
        for (int ky = ky0; ky <= ky1; ++ky) {
          const bool lastk = ky == ky1;
          // the tap row whose weights are requested now: the next one of this chunk, the first one of the next chunk, the
          // first one of the next task, or (at the very end) this one again -- a harmless reload: the loads stay unconditional
          const bool to_next = lastk && !more && have_next && nxt.ky1 >= nxt.ky0;
          const int qn = lastk ? (more ? q + 1 : (to_next ? 0 : q)) : q;
          const int kyn = lastk ? (more ? ky0 : (to_next ? nxt.ky0 : ky)) : ky + 1;
          const int ntn = to_next ? nxt.nt : cur.nt;
          const int kya = lastk ? ky : ky + 1;               // A comes from THIS buffer: past the last tap row, re-read it
          // the patch row fetched during this tap row: row ky of the next chunk, or (last chunk) a row of the next task's chunk 0
          const int nky = nxt.ky0 + (ky - ky0);
          const bool st_next = !more && have_next && nky <= nxt.ky1;
#ifdef TF_ABL_NOFILL      // ablation builds (tools/build_variant.sh; results are then wrong): no patch fills at all
          const bool st = false;
#else
          const bool st = more || st_next;
#endif
          const int srow = more ? ky : nky;
          f32x4 stage[G::NPR];
          if (st) {
            const int rowoff = ((more ? ry : nxt.ry) + D * srow) * d.in_w * d.in_cstore * 4;
            const __amdgpu_buffer_rsrc_t rs = more ? xs : xn;
#pragma unroll
            for (int i = 0; i < G::NPR; ++i)
#ifdef TF_ABL_CHUNKMAJOR      // ablation: the addresses a chunk-major image [row][chunk][pixel][8] would have (whole lines, one contiguous run)
              stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + TF_THREADS * i) * 16 + rowoff, (more ? q + 1 : 0) * d.in_w * 32, 0));
#else
              stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[i] + rowoff, more ? 32 * (q + 1) : 0, 0));
#endif
          }
          for (int rep = 1; rep < dbg_repeat; ++rep) {      // DD_DCONV_REPEAT: timing diagnostic only (results are then wrong)
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
              acc[i] = DD_MFMA(Af[i % AR].x, Bf[i].x, acc[i]);
              acc[i] = DD_MFMA(Af[i % AR].y, Bf[i].y, acc[i]);
              acc[i] = DD_MFMA(Af[i % AR].z, Bf[i].z, acc[i]);
              acc[i] = DD_MFMA(Af[i % AR].w, Bf[i].w, acc[i]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if constexpr (ONE_MT) {
            const frag An = *(const frag*)(lbase + aoff[0] + kya * (G::ROWF * 4));
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
              acc[i] = DD_MFMA(Af[0].x, Bf[i].x, acc[i]);
              acc[i] = DD_MFMA(Af[0].y, Bf[i].y, acc[i]);
              acc[i] = DD_MFMA(Af[0].z, Bf[i].z, acc[i]);
              acc[i] = DD_MFMA(Af[0].w, Bf[i].w, acc[i]);
              bload(i, qn, kyn, ntn);                      // in flight for the NSLOT - 1 tiles until this slot comes round again
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
#ifndef TF_ABL_NOB      // ablation: weight fragments loaded once
              bload(i, qn, kyn, ntn);
#endif
              Af[i % AR] = *(const frag*)(lbase + aoff[(i + AR) % NSLOT] + (i + AR < NSLOT ? ky : kya) * (G::ROWF * 4));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (st) {
#pragma unroll
            for (int i = 0; i < G::NPR; ++i) {
              const int p = tid + TF_THREADS * i;
              if (p < IWP * 2) *(f32x4*)&nbuf[srow * G::ROWF + p * 4] = stage[i];
            }
          }
        }
        tf_barrier();
        par ^= 1;
      }
      // lds[par] now holds what was fetched during the last chunk: the first rows of the next task's chunk 0.  Rows it has
      // beyond this task's count (a border row followed by a fuller one) are fetched here, not overlapped.
      fetched_next = have_next && nxt.ky1 >= nxt.ky0;
      if (fetched_next) {
        for (int nky = nxt.ky0 + (ky1 - ky0 + 1); nky <= nxt.ky1; ++nky) {
          const int rowoff = (nxt.ry + D * nky) * d.in_w * d.in_cstore * 4;
#pragma unroll
          for (int i = 0; i < G::NPR; ++i) {
            const int p = tid + TF_THREADS * i;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xn, poff[i] + rowoff, 0, 0));
            if (p < IWP * 2) *(f32x4*)&lds[par][nky * G::ROWF + p * 4] = v;
          }
        }
      }
    }

    // ---- epilogue in the buffer the next task does not start from: the K partial rows are added into one output row image
    // at their shifts.  Tap column K-1 (shift 0) goes first and is STORED (its tiles cover pixels [0, n_mt*TW); the HALO pixels
    // beyond are zeroed in the same phase), the others are added in K-1 barrier-separated passes.
    float* img = &lds[par ^ 1][0];
    constexpr int LPP = 8;                                   // lanes per pixel of the write-out (16 bytes each)
    constexpr int WIT = ((IWP + G::HALO) * LPP + TF_THREADS - 1) / TF_THREADS;      // out_w <= n_mt*TW + HALO <= IWP + HALO
    const int c4 = 4 * (tid % LPP), cch = 32 * cur.nt + c4;  // 512 % LPP == 0: a thread always writes the same channel group
    f32x4 bvec = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) && cch < d.cout) bvec = *(const f32x4*)&bias[cch];
    {
      if (tid < G::HALO * (P / 4)) *(f32x4*)&img[n_mt * TW * P + tid * 4] = f32x4{0.f, 0.f, 0.f, 0.f};
      static_assert(G::HALO * (P / 4) <= TF_THREADS, "one zeroing store per thread");
#ifdef TF_ABL_NOEPI      // ablation: one pass of the K
#pragma unroll
      for (int pass = K - 1; pass >= K - 1; --pass) {
#else
#pragma unroll
      for (int pass = K - 1; pass >= 0; --pass) {
#endif
        const int shift = D * (K - 1 - pass);                // = pad_w - D*pass (the launcher checked pad_w)
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          if (ONE_MT ? (i == pass && s_ok[i]) : (s_ok[i] && s_kx[i] == pass)) {
            float* p0 = img + ioff_lane + s_mt[i] * (TW * P);
            // (pass < K - 1: all 16 reads, then all 16 adds and writes -- written as `*pe += acc` the compiler made eight serial
            // ds_read2 -> s_waitcnt -> v_add -> ds_write2 round trips of it, ~2000 cycles per pass with every wave of the CU waiting;
            // the registers of the tiles stored in earlier passes are free by now.  ds_add_f32, the LDS unit's own float add, is far
            // slower still: up_conv_1's forward 9.6 -> 13.8 ms)
            float t[NE];
            if (pass != K - 1) {
#pragma unroll
              for (int e = 0; e < NE; ++e) t[e] = p0[((e & 3) + 8 * (e >> 2) + shift) * P];
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
              float* pe = p0 + ((e & 3) + 8 * (e >> 2) + shift) * P;
              *pe = pass == K - 1 ? acc[i][e] : t[e] + acc[i][e];
            }
          }
        }
        tf_barrier();
      }
    }
    {
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const int base = ((cur.oy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff + cch;
      f32x4 v[WIT];
#pragma unroll
      for (int j = 0; j < WIT; ++j) {
        const int px = (tid + TF_THREADS * j) / LPP;
        v[j] = px < d.out_w ? *(const f32x4*)&img[px * P + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < WIT; ++j) {
        const int px = (tid + TF_THREADS * j) / LPP;
        f32x4 o = v[j] + bvec;
        if (epi == DD_EPI_BIAS_RELU) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        const int off = (px < d.out_w && cch < d.cout) ? (base + px * d.out_cstore) * 4 : -16;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), ys, off, 0, 0);
      }
    }
    tf_barrier();
    prefetched = fetched_next;
    cur = nxt;
    t = tn;
  }
}


// =====================================================================================================================
// The DATA GRADIENT of the same layers (a plain dilated convolution: pad 0, out = in - d(k-1), every tap valid), one output
// row per workgroup like the forward above: no row-group quantisation, and a tap loop without iterators -- tap columns
// unrolled, one ds_read_b128 and NTC weight loads per 4*NTC MFMAs, each requested a whole tap row ahead into the registers
// the tap's MFMAs have just read (the tap loop of dconv_fwd_kernel spends 17 % of its issue slots on its (ky, kx) iterator,
// operand addresses and waits: T(r) = r*9.8 + 0.45 ms against 8.4 ms of MFMAs for up_conv_1).
//   wave w owns m-tile w (32 output pixels) x all NTC column tiles: rows of at most 8 m-tiles (up_conv_1: 256).  Wider rows
//   (up_conv_2: 298 -> 10 m-tiles) stay on dconv_fwd_kernel: dealing the two extra m-tiles to the 8 waves as tap-split shared
//   tiles was tried and measured slower (5.19 against 5.02 ms: registers spilled inside the tap loop).
template <int K, int D, int NTC, int IWP>
__global__ __launch_bounds__(TF_THREADS) void dconv_gfwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, const float* __restrict__ msk,
                                                                float* __restrict__ y, const dd_gconv_desc d, int epi, int wp_bytes) {
  using G = TfGeom<K, D, 1, IWP>;
  constexpr int T = K * K;
  __shared__ __attribute__((aligned(16))) float lds[2][G::BUFF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, n = lane & 31;
  const int NC = d.cin >> 3;
  const int rows_max = (d.out_h + D - 1) / D;
  const int n_mt = (d.out_w + 31) >> 5;
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  const __amdgpu_buffer_rsrc_t ws = dd_rsrc(wp, wp_bytes);

  int poff[G::NPR];
#pragma unroll
  for (int i = 0; i < G::NPR; ++i) {
    const int p = tid + TF_THREADS * i;
    poff[i] = (p >> 1) < d.in_w ? ((p >> 1) * d.in_cstore + d.in_coff + 4 * (p & 1)) * 4 : (int)0xC0000000;
  }
  const int aoff_own = ((wave * 32 + n) * 8 + 4 * h) * 4;

  float bv[NTC];
#pragma unroll
  for (int nt = 0; nt < NTC; ++nt) {
    const int ch = nt * 32 + n;
    bv[nt] = (bias && (epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) && ch < d.cout) ? bias[ch] : 0.f;
  }

  const int per_x = gridDim.x >> 3;      // the launchers round the grid down to a multiple of 8 (one segment per XCD)
  const int xcd = blockIdx.x & 7;
  const int per_img = D * rows_max;
  const long len = (long)d.batch * per_img;
  const long seg1 = len * (xcd + 1) / 8;
  auto decode = [&](long t, int& b, int& oy) -> bool {
    b = (int)(t / per_img);
    const int rem = (int)(t - (long)b * per_img);
    const int r = rem / rows_max, jy = rem - r * rows_max;
    oy = r + D * jy;
    return oy < d.out_h;
  };
  auto next_task = [&](long t, int& b, int& oy) -> long {
    while (t < seg1 && !decode(t, b, oy)) t += per_x;
    return t < seg1 ? t : seg1;
  };

  f32x4 Bf[K][NTC], Aown[K];
  auto bload = [&](int kx, int q, int ky) {
    const int soff = ((q * T + ky * K + kx) * NTC) * 1024;
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
      Bf[kx][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws, lane * 16, soff + nt * 1024, 0));
  };

  int cb, coy, nb, noy;
  long t = next_task(len * xcd / 8 + (blockIdx.x >> 3), cb, coy);
  bool prefetched = false;
  int par = 0;
  while (t < seg1) {
    const long tn = next_task(t + per_x, nb, noy);
    const bool have_next = tn < seg1;
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)cb * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const __amdgpu_buffer_rsrc_t xn = dd_rsrc(x + (long)nb * d.in_h * d.in_w * d.in_cstore, have_next ? in_bytes : 0);

    f32x16 acc[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    if (!prefetched) {
      f32x4 v[K][G::NPR];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int rowoff = (coy + D * ky) * d.in_w * d.in_cstore * 4;
#pragma unroll
        for (int i = 0; i < G::NPR; ++i) v[ky][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xs, poff[i] + rowoff, 0, 0));
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int i = 0; i < G::NPR; ++i) {
          const int p = tid + TF_THREADS * i;
          if (p < IWP * 2) *(f32x4*)&lds[par][ky * G::ROWF + p * 4] = v[ky][i];
        }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) bload(kx, 0, 0);
    }
    tf_barrier();
    for (int q = 0; q < NC; ++q) {
      const bool more = q + 1 < NC;
      const char* lbase = (const char*)&lds[par][0];
      float* nbuf = &lds[par ^ 1][0];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) Aown[kx] = *(const f32x4*)(lbase + aoff_own + kx * D * 32);
      for (int ky = 0; ky < K; ++ky) {
        const bool lastk = ky == K - 1;
        const bool to_next = lastk && !more && have_next;
        const int qn = lastk ? (more ? q + 1 : (to_next ? 0 : q)) : q;
        const int kyn = lastk ? ((more || to_next) ? 0 : ky) : ky + 1;
        const int kya = lastk ? ky : ky + 1;
        const bool st = more || have_next;
        f32x4 stage[G::NPR];
        if (st) {
          const int rowoff = ((more ? coy : noy) + D * ky) * d.in_w * d.in_cstore * 4;
          const __amdgpu_buffer_rsrc_t rs = more ? xs : xn;
#pragma unroll
          for (int i = 0; i < G::NPR; ++i)
            stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[i] + rowoff, more ? 32 * (q + 1) : 0, 0));
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int nt = 0; nt < NTC; ++nt) {
            acc[nt] = DD_MFMA(Aown[kx].x, Bf[kx][nt].x, acc[nt]);
            acc[nt] = DD_MFMA(Aown[kx].y, Bf[kx][nt].y, acc[nt]);
            acc[nt] = DD_MFMA(Aown[kx].z, Bf[kx][nt].z, acc[nt]);
            acc[nt] = DD_MFMA(Aown[kx].w, Bf[kx][nt].w, acc[nt]);
          }
          bload(kx, qn, kyn);
          Aown[kx] = *(const f32x4*)(lbase + aoff_own + kya * (G::ROWF * 4) + kx * D * 32);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (st) {
#pragma unroll
          for (int i = 0; i < G::NPR; ++i) {
            const int p = tid + TF_THREADS * i;
            if (p < IWP * 2) *(f32x4*)&nbuf[ky * G::ROWF + p * 4] = stage[i];
          }
        }
      }
      tf_barrier();
      par ^= 1;
    }

    // ---- write-out (as in dconv_fwd_kernel: all mask values of a tile requested before the first is used)
    {
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cb * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const __amdgpu_buffer_rsrc_t ms = dd_rsrc(msk ? msk + (long)cb * d.omem_h * d.omem_w * d.out_cstore : y, msk ? out_bytes : 0);
      const bool masked = epi == DD_EPI_RELU_MASK;
      const int base = ((coy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff;
      auto put = [&](const f32x16& a, int mt, int nt, float bvn, bool on) {
        const int ch = nt * 32 + n;
        const bool pass = d.out_coff + ch >= d.mask_pass_lo && d.out_coff + ch < d.mask_pass_hi;
        int off[16];
        float mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int xo = mt * 32 + dd_acc_row(r, lane);
          const bool ok = on && xo < d.out_w && ch < d.cout;
          off[r] = ok ? (base + xo * d.out_cstore + ch) * 4 : -16;
          mv[r] = 1.f;
        }
        if (masked) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float m = dd_bload1(ms, pass ? -16 : off[r]);
            mv[r] = pass ? 1.f : m;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = a[r] + bvn;
          if (epi == DD_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
          v = mv[r] > 0.f ? v : 0.f;
          dd_bstore1(ys, off[r], v);
        }
      };
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) put(acc[nt], wave, nt, bv[nt], wave < n_mt);
    }
    tf_barrier();
    prefetched = have_next;
    cb = nb; coy = noy;
    t = tn;
  }
}

}  // namespace

// Launches the input-aligned forward if the layer is one it is built for; returns false (nothing launched) otherwise.
bool dd_dconv_tfwd_launch(const float* x, const float* packed, const float* bias, float* y, const dd_gconv_desc* d, int epilogue,
                          int wp_bytes, hipStream_t st) {
  if (!dd_dconv_desc_ok(d)) return false;                // the shared eligibility test (stride, div, ostride, channel bounds, sizes)
  if (epilogue != DD_EPI_NONE && epilogue != DD_EPI_BIAS && epilogue != DD_EPI_BIAS_RELU) return false;
  if (d->kh != d->kw || d->dil_h != d->dil_w) return false;
  const int k = d->kh, dl = d->dil_h;
  if (d->pad_h != dl * (k - 1) || d->pad_w != dl * (k - 1)) return false;                    // the full transposed form only
  if (d->out_h < d->in_h + dl * (k - 1) || d->out_w < d->in_w + dl * (k - 1)) return false;  // every partial lands inside the row
  if (d->cin % 8 || d->cout <= 16 || d->cout % 4 || d->out_coff % 4 || d->out_cstore % 4) return false;
  if (((uintptr_t)bias & 15) != 0) return false;
  if ((long)d->in_h * d->in_w * d->in_cstore * 4 >= (1L << 30)) return false;      // rejected offsets must stay rejected with a row offset added
  const int grid = dd_cu_budget_internal() & ~7;
  if (grid < 8) return false;
#ifdef DD_TIMING_DIAG      // diagnostic builds only (hipcc -DDD_TIMING_DIAG): repeating the tap loop times it, the results are then wrong
  static const int dbg_repeat = getenv("DD_DCONV_REPEAT") ? atoi(getenv("DD_DCONV_REPEAT")) : 1;
#else
  constexpr int dbg_repeat = 1;
#endif
#define DD_TF(KK, DD_, NS, MODE_, IWP_)                                                                                        \
  do {                                                                                                                         \
    using G = TfGeom<KK, DD_, MODE_, IWP_>;                                                                                    \
    const int n_mt = (d->in_w + G::TW - 1) / G::TW;                                                                            \
    if (d->in_w > IWP_ || (n_mt * G::TW + G::HALO) * G::P > G::BUFF || d->out_w > n_mt * G::TW + G::HALO || n_mt * KK > 8 * NS ||      \
        (MODE_ == 0 && n_mt > 8))                                                                                              \
      return false;                                                                                                            \
    hipLaunchKernelGGL((dconv_tfwd_kernel<KK, DD_, NS, MODE_, IWP_>), dim3(grid), dim3(TF_THREADS), 0, st, x, packed, bias, y, \
                       *d, epilogue, wp_bytes, dbg_repeat);                                                                   \
    return true;                                                                                                               \
  } while (0)
  if (k == 7 && dl == 7 && d->cout > 16 && d->in_w <= 256) DD_TF(7, 7, 7, 0, 256);
  if (k == 7 && dl == 7 && d->cout > 16 && d->in_w <= 320) DD_TF(7, 7, 9, 1, 320);
#undef DD_TF
  return false;
}

// The same for the data gradient (gather form, pad 0): false = not one of its layers, nothing launched.
bool dd_dconv_gfwd_launch(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                          int epilogue, int wp_bytes, hipStream_t st) {
  if (!dd_dconv_desc_ok(d)) return false;
  if (epilogue != DD_EPI_NONE && epilogue != DD_EPI_BIAS && epilogue != DD_EPI_BIAS_RELU && epilogue != DD_EPI_RELU_MASK) return false;
  if (epilogue == DD_EPI_RELU_MASK && !mask) return false;
  if (d->kh != d->kw || d->dil_h != d->dil_w || d->pad_h != 0 || d->pad_w != 0) return false;
  const int k = d->kh, dl = d->dil_h, halo = dl * (k - 1);
  if (d->out_h > d->in_h - halo || d->out_w > d->in_w - halo || d->cin % 8) return false;
  if ((long)d->in_h * d->in_w * d->in_cstore * 4 >= (1L << 30)) return false;
  const int grid = dd_cu_budget_internal() & ~7;
  if (grid < 8) return false;
  const int n_mt = (d->out_w + 31) / 32, ntc = (d->cout + 31) / 32;
#define DD_GF(KK, DD_, NTC_, IWP_)                                                                                             \
  do {                                                                                                                         \
    hipLaunchKernelGGL((dconv_gfwd_kernel<KK, DD_, NTC_, IWP_>), dim3(grid), dim3(TF_THREADS), 0, st, x, packed, bias,        \
                       mask, y, *d, epilogue, wp_bytes);                                                                       \
    return true;                                                                                                               \
  } while (0)
  if (k == 7 && dl == 7 && ntc == 3 && n_mt <= 8 && d->in_w <= 320) DD_GF(7, 7, 3, 320);
#undef DD_GF
  return false;
}

namespace {

// =====================================================================================================================
// up_conv_4's forward (ConvTranspose2d 16 -> 8, k7, dilation 3: spatial_bb/components.py:138), input-aligned like dconv_tfwd_kernel, with
// FOUR TAP COLUMNS PACKED INTO ONE COLUMN TILE.  Cout = 8 fills a quarter of a 32-wide tile (half of a 16-wide one: the gather kernel it
// ran on, dconv_fwd_kernel<7, 3, 0>, reached 0.32 of the fp32 pipe, the least of any matrix kernel of the box heads).  In the input-aligned
// form the A operand -- 32 input pixels x 8 channels -- is the same for every tap column; only the weights and the output shift differ.
// So the B tile is [kx = 4g: 8 co | 4g + 1 | 4g + 2 | 4g + 3] for g = 0, 1 (column 7 of g = 1 is zero): 7/8 of every MFMA is useful, two
// accumulator tiles per m-tile instead of seven, and lane (kxl = col >> 3, co = col & 7) adds its column at ITS shift pad_w - D kx.
//   * a workgroup (8 waves) owns one output row; per tap row ky the input row oy - pad_h + D ky (in_w pixels x 16 channels, 80-byte pixel
//     pitch: an odd number of 16-byte slots, conflict-free ds_read_b128) sits in LDS, double buffered across tap rows AND tasks;
//   * all 49 x 16 x 8 weights sit in LDS for the workgroup's life (28 KB, re-laid from the standard image once);
//   * 12 m-tiles x 2 groups = 24 tiles, three per wave (all of one group and one m-tile parity: the epilogue's four barrier-separated
//     classes -- (group, parity) -- never touch the same pixel from two waves; within a wave LDS operations are ordered).
template <int K, int D>
__global__ __launch_bounds__(TF_THREADS) void dconv_tfwd8_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 const dd_gconv_desc d, int epi) {
  constexpr int IWP = 384, PITCH = 80;     // bytes per pixel of a row image (16 channels + 16 bytes)
  constexpr int ROWB = IWP * PITCH;
  constexpr int HALO = D * (K - 1);
  constexpr int NPR = (IWP * 4 + TF_THREADS - 1) / TF_THREADS;      // 16-byte pieces per thread and row
  constexpr int IMGF = (IWP + HALO) * 8;                   // floats of the output row image: 8 channels per pixel
  __shared__ __attribute__((aligned(16))) char rows[2][ROWB];
  __shared__ __attribute__((aligned(16))) f32x4 wl[K][2][2][64];
  __shared__ __attribute__((aligned(16))) float img[4][IMGF];      // one plane per tap column of a group (see the epilogue)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, n = lane & 31;
  const int g = wave & 1;                                   // this wave's tap-column group; its m-tiles: (wave >> 1) + 4 j
  const int in_bytes = d.in_h * d.in_w * d.in_cstore * 4;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;

  // ---- weights: wl[ky][g][q][l] = four channels (8q + 4h + j) of column (kxl = n >> 3, co = n & 7) of tap (ky, 4g + kxl), from the standard
  // image of dd_dconv_pack for Cout <= 16: wp[((q*T + tap)*64 + ((c8 >> 1) << 4 | co))*2 + (c8 & 1)], c8 = the channel inside its chunk
  for (int i = tid; i < K * 2 * 2 * 64; i += TF_THREADS) {
    const int l = i & 63, q = (i >> 6) & 1, gg = (i >> 7) & 1, ky = i >> 8;
    const int kx = 4 * gg + ((l & 31) >> 3), co = l & 7, hh = l >> 5;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kx < K && co < d.cout) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c8 = 4 * hh + j;
        v[j] = wp[((q * (K * K) + ky * K + kx) * 64 + (((c8 >> 1) << 4) | co)) * 2 + (c8 & 1)];
      }
    }
    wl[ky][gg][q][l] = v;
  }

  // ---- fill plan: piece p = tid + 512 i of a row -> pixel p >> 2, 16-byte slot p & 3
  int goff[NPR], loff[NPR];
#pragma unroll
  for (int i = 0; i < NPR; ++i) {
    const int p = tid + TF_THREADS * i, px = p >> 2, sl = p & 3;
    goff[i] = (px < d.in_w) ? (px * d.in_cstore + d.in_coff + 4 * sl) * 4 : (int)0xC0000000;
    loff[i] = (px < IWP) ? px * PITCH + sl * 16 : -1;
  }
  // per-lane constants of the epilogue: element r of a tile -> float index (32 mt + row(r, h) + shift) * 8 + co
  const int kx_lane = 4 * g + (n >> 3);
  const bool col_ok = kx_lane < K && (n & 7) < d.cout;
  const int e_lane = (n >> 3) * IMGF + (4 * h + d.pad_w - D * kx_lane) * 8 + (n & 7);      // (the launcher checked pad_w == D (K - 1): every shift >= 0)

  // ---- tasks: as dconv_tfwd_kernel -- XCD = blockIdx % 8 owns an eighth of the (image, residue class, phase row) list
  const int rows_max = (d.out_h + D - 1) / D;
  const int per_x = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int per_img = D * rows_max;
  const long len = (long)d.batch * per_img;
  const long seg1 = len * (xcd + 1) / 8;
  auto decode = [&](long t, TfTask& k) -> bool {
    k.b = (int)(t / per_img);
    const int rem = (int)(t - (long)k.b * per_img);
    const int r = rem / rows_max, jy = rem - r * rows_max;
    k.nt = 0;
    k.oy = r + D * jy;
    k.ry = k.oy - d.pad_h;
    k.ky0 = k.ry >= 0 ? 0 : (-k.ry + D - 1) / D;
    k.ky1 = min(K - 1, (d.in_h - 1 - k.ry) >= 0 ? (d.in_h - 1 - k.ry) / D : -1);
    return k.oy < d.out_h && k.ky1 >= k.ky0;
  };
  auto next_task = [&](long t, TfTask& k) -> long {
    while (t < seg1 && !decode(t, k)) t += per_x;
    return t < seg1 ? t : seg1;
  };
  auto request = [&](const TfTask& k, int ky, f32x4 (&st)[NPR]) {
    const __amdgpu_buffer_rsrc_t xs = dd_rsrc(x + (long)k.b * d.in_h * d.in_w * d.in_cstore, in_bytes);
    const int rowoff = (k.ry + D * ky) * d.in_w * d.in_cstore * 4;
#pragma unroll
    for (int i = 0; i < NPR; ++i) st[i] = dd_bload4(xs, (int)((unsigned)goff[i] + (unsigned)rowoff));
  };
  auto retire = [&](int buf, const f32x4 (&st)[NPR]) {
#pragma unroll
    for (int i = 0; i < NPR; ++i)
      if (loff[i] >= 0) *(f32x4*)(rows[buf] + loff[i]) = st[i];
  };

  TfTask cur, nxt;
  long t = next_task(len * xcd / 8 + (blockIdx.x >> 3), cur);
  if (t >= seg1) return;
  f32x4 stage[NPR];
  request(cur, cur.ky0, stage);
  retire(0, stage);
  __syncthreads();                                          // weights and the first row
  int par = 0;
  while (t < seg1) {
    const long tn = next_task(t + per_x, nxt);
    const bool have_next = tn < seg1;
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int ky = cur.ky0; ky <= cur.ky1; ++ky) {
      const bool lastk = ky == cur.ky1;
      const bool st = !lastk || have_next;
      if (st) request(lastk ? nxt : cur, lastk ? nxt.ky0 : ky + 1, stage);
      __builtin_amdgcn_sched_barrier(0);
      const char* rb = rows[par] + n * PITCH + h * 16;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 B = wl[ky][g][q][lane];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int mt = (wave >> 1) + 4 * i;
          const f32x4 A = *(const f32x4*)(rb + mt * 32 * PITCH + q * 32);
          acc[i] = DD_MFMA(A.x, B.x, acc[i]);
          acc[i] = DD_MFMA(A.y, B.y, acc[i]);
          acc[i] = DD_MFMA(A.z, B.z, acc[i]);
          acc[i] = DD_MFMA(A.w, B.w, acc[i]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (st) retire(par ^ 1, stage);
      if (!lastk) {
        tf_barrier();
        par ^= 1;
      }
    }
    // ---- epilogue: zero the row image, add the tiles class by class, write the row out.  (The fill of the next task's first row was
    // retired into the other buffer above; the barriers below publish it too.)  The image has FOUR PLANES, one per tap column of a group:
    // inside a tile the columns of different taps land on the same output pixel from different accumulator rows (row + shift), and a
    // batched read-add-write of all 16 rows would lose one of the two; per plane a tile's lanes never meet, the write-out adds the planes.
    for (int i = tid; i < 4 * IMGF / 4; i += TF_THREADS) ((f32x4*)&img[0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    tf_barrier();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c == ((wave & 1) | (((wave >> 1) & 1) << 1)) && col_ok) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float* p0 = &img[0][0] + e_lane + ((wave >> 1) + 4 * i) * 256;
          float tv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) tv[e] = p0[((e & 3) + 8 * (e >> 2)) * 8];
#pragma unroll
          for (int e = 0; e < 16; ++e) p0[((e & 3) + 8 * (e >> 2)) * 8] = tv[e] + acc[i][e];
        }
      }
      tf_barrier();
    }
    {
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const int base = ((cur.oy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff;
      for (int i = tid; i < d.out_w * 2; i += TF_THREADS) {      // 16-byte pieces: pixel i >> 1, channels 4 (i & 1) ..
        const int px = i >> 1, c4 = 4 * (i & 1);
        f32x4 o = (*(const f32x4*)&img[0][px * 8 + c4] + *(const f32x4*)&img[1][px * 8 + c4]) +
                  (*(const f32x4*)&img[2][px * 8 + c4] + *(const f32x4*)&img[3][px * 8 + c4]);
        if (epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) {
          o.x += c4 + 0 < d.cout ? bias[c4 + 0] : 0.f; o.y += c4 + 1 < d.cout ? bias[c4 + 1] : 0.f;
          o.z += c4 + 2 < d.cout ? bias[c4 + 2] : 0.f; o.w += c4 + 3 < d.cout ? bias[c4 + 3] : 0.f;
        }
        if (epi == DD_EPI_BIAS_RELU) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        const int off = c4 < d.cout ? (base + px * d.out_cstore + c4) * 4 : -16;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), ys, off, 0, 0);
      }
    }
    tf_barrier();
    par ^= 1;
    cur = nxt;
    t = tn;
  }
}

}  // namespace

bool dd_dconv_tfwd8_launch(const float* x, const float* packed, const float* bias, float* y, const dd_gconv_desc* d, int epilogue,
                           hipStream_t st) {
  if (!dd_dconv_desc_ok(d)) return false;
  if (epilogue != DD_EPI_NONE && epilogue != DD_EPI_BIAS && epilogue != DD_EPI_BIAS_RELU) return false;
  if (d->kh != 7 || d->kw != 7 || d->dil_h != 3 || d->dil_w != 3 || d->pad_h != 18 || d->pad_w != 18) return false;
  if (d->cin != 16 || d->cout < 1 || d->cout > 8 || d->cout % 4 || d->out_coff % 4 || d->out_cstore % 4) return false;
  if (d->in_w > 384 || d->in_w < 32 || d->out_w != d->in_w + 18 || d->out_h != d->in_h + 18) return false;
  if ((long)d->in_h * d->in_w * d->in_cstore * 4 >= (1L << 30)) return false;
  if (((uintptr_t)bias & 3) != 0) return false;
  const int grid = dd_cu_budget_internal() & ~7;
  if (grid < 8) return false;
  hipLaunchKernelGGL((dconv_tfwd8_kernel<7, 3>), dim3(grid), dim3(TF_THREADS), 0, st, x, packed, bias, y, *d, epilogue);
  return true;
}
