// EXPERIMENT (round 4; off by default): the forward of the dilated stride-1 ConvTranspose2d layers of the box heads
// (spatial_bb/components.py:135-136) with fp32-equivalent products on the BF16 matrix pipe.
//
// The exact-fp32 kernels of dconv_t.hip keep the fp32 matrix pipe 87-90 % busy: on that pipe (157 TF) they are done.  The bf16
// pipe runs 16x faster per multiply, and an fp32 number splits into three bf16 pieces (8 + 8 + 8 significant bits: x = hi + mid + lo to
// 2^-27 |x|, each piece the round-to-nearest of the remainder of the one before), so
//     a * b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + [am*bl + al*bm + al*bl]
// and the three terms in brackets are below 2^-23 of |a*b| -- the size of fp32's own rounding of the product.  The six others are
// issued on v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact in fp32, accumulation is fp32), smallest first:
// 6 x 32 cycles per 32 x 32 x 16 block against 8 x 64 for v_mfma_f32_32x32x2_f32 = 6/16 of the matrix time.
//
// Operands are split ONCE, outside the kernel, into the images the kernel's LDS wants, so that every stage fill is a plain
// lane-linear copy done by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass):
//   * dd_dconv_split_input: x (NHWC fp32, a channel slice) -> xs[b][y][q][px][112 B]: per pixel and 16-channel chunk q the three
//     planes hi / mid / lo (16 bf16 = 32 B each) and 16 B of padding -- 112 B = 7 sixteen-byte slots, an odd number, so the sixteen
//     lanes of a ds_read_b128 group (pixels p, p+1, ..: 7p mod 16 is a bijection) cover all 64 banks: conflict-free as stored;
//   * dd_dconv_split_pack: W -> wp[q][ky][nt][kx][plane][lane][8 bf16]: the B fragments of one (chunk, tap row, column tile) are
//     one contiguous 21 KB block in lane order.
// The kernel is dconv_tfwd_kernel's input-aligned form (one accumulator tile per (m-tile, tap column), partial rows added at their
// shifts in LDS: no border zero is multiplied) with the stage cut at (16-channel chunk, tap row): per stage a wave issues 3 + 21
// ds_read_b128 and 42 MFMAs; the next stage's 49-56 KB arrive by DMA meanwhile (two LDS buffers, one s_barrier per stage).
//
// Same operands, same products up to 2^-23 each, same fp32 accumulation as the exact kernels in a different order: held to the
// exact kernels' own tests (2e-5 of peak against fp64).  The headline numbers and `dtype: f32` never use this path.
#include <stdlib.h>

#include "dd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
#define SP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int SP_THREADS = 512;
constexpr int SP_PXB = 112;                 // bytes of one pixel of one 16-channel chunk in xs and in LDS

// a wave-uniform value the compiler keeps in vector registers (it then wraps every descriptor built from it in a readfirstlane loop): back to scalars
__device__ __forceinline__ int sp_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char* sp_uni(const char* p) {
  const unsigned long a = (unsigned long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (const char*)(((unsigned long)hi << 32) | lo);
}

__device__ __forceinline__ void sp_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- x = hi + mid + lo: three ROUNDINGS to nearest-even (v_cvt_pk_bf16_f32), each of the remainder of the one before.  Both
// remainders are exact fp32 subtractions, so the three pieces miss x by the last rounding only (<= 2^-27 |x|).  Truncation would make
// the sum exact, but every piece then has the sign of x and the three dropped cross products (am*bl + al*bm + al*bl) the sign of a*b: a
// BIASED 2^-23 |a b| per product, which a 67 M-term weight-gradient sum with heavy cancellation turns into 2e-5 of its result (measured:
// the B = 32 box step against the mean of 32 single-scene steps); with rounded pieces the dropped terms are zero-mean and the error
// grows like a random walk, as the exact fp32 kernels' own rounding does.
typedef __attribute__((ext_vector_type(2))) __bf16 sp_bf16x2;
typedef __attribute__((ext_vector_type(2))) float sp_f32x2;
__device__ __forceinline__ unsigned sp_rne_bits(float x) {      // the bf16 nearest x, as the upper half of an fp32 pattern
  const unsigned two = __builtin_bit_cast(unsigned, __builtin_convertvector((sp_f32x2){x, 0.f}, sp_bf16x2));
  return two << 16;
}
__device__ __forceinline__ void sp_split(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = sp_rne_bits(x);
  const float r = x - __builtin_bit_cast(float, hi);
  mid = sp_rne_bits(r);
  const float s = r - __builtin_bit_cast(float, mid);
  lo = sp_rne_bits(s);
}

// one thread per (image row, chunk, pixel, 8-channel half): 32 B in, 3 x 16 B out (+ the 16 B pad from half 0)
__global__ __launch_bounds__(256) void split_input_kernel(const float* __restrict__ x, char* __restrict__ xs, long rows, int W, int cstore,
                                                          int coff, int NC) {
  const long total = rows * NC * W * 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int half = (int)(idx & 1);
    const long t = idx >> 1;
    const int px = (int)(t % W);
    const long u = t / W;
    const int q = (int)(u % NC);
    const long r = u / NC;
    const float* src = x + (r * W + px) * cstore + coff + 16 * q + 8 * half;
    const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sp_split(v[j], h[j], m[j], l[j]);
    char* dst = xs + ((r * NC + q) * W + px) * SP_PXB + half * 16;
    u32x4s oh, om, ol;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      oh[j] = (h[2 * j] >> 16) | h[2 * j + 1];
      om[j] = (m[2 * j] >> 16) | m[2 * j + 1];
      ol[j] = (l[2 * j] >> 16) | l[2 * j + 1];
    }
    *(u32x4s*)dst = oh;
    *(u32x4s*)(dst + 32) = om;
    *(u32x4s*)(dst + 64) = ol;
    if (half == 0) *(u32x4s*)(dst + 96) = u32x4s{0u, 0u, 0u, 0u};
  }
}

// wp[(((((q*K + ky)*NT + nt)*K + kx)*3 + plane)*64 + lane)*8 + j] = plane of W(n = 32 nt + (lane & 31), c = 16 q + 8 (lane >> 5) + j, tap)
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ p, int nchunks, int NT, int K,
                                                         long w_off, long sn, long sc, int flip, int n_real, int c_real) {
  const long total = (long)nchunks * K * NT * K * 512;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  long g = idx >> 9;
  const int kx = (int)(g % K);
  g /= K;
  const int nt = (int)(g % NT);
  g /= NT;
  const int ky = (int)(g % K), q = (int)(g / K);
  const int tap = ky * K + kx, T = K * K;
  const int c = 16 * q + 8 * (lane >> 5) + j, n = nt * 32 + (lane & 31);
  float v = 0.f;
  if (n < n_real && c < c_real) v = w[w_off + n * sn + c * sc + (flip ? T - 1 - tap : tap)];
  unsigned hi, mid, lo;
  sp_split(v, hi, mid, lo);
  const long base = ((((long)(q * K + ky) * NT + nt) * K + kx) * 3) * 512 + lane * 8 + j;
  p[base] = (unsigned short)(hi >> 16);
  p[base + 512] = (unsigned short)(mid >> 16);
  p[base + 1024] = (unsigned short)(lo >> 16);
}

struct SpTask {
  int b, nt, oy, ry, ky0, ky1;
};

template <int K, int D, int NSLOT, int MODE, int IWP, int abl, int NBUF>
__global__ __launch_bounds__(SP_THREADS) void dconv_stfwd_kernel(const char* __restrict__ xs, const char* __restrict__ wp,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 char* __restrict__ yplanes, const dd_gconv_desc d, int epi) {
  // yplanes (optional): the three-plane bf16 image of the OUTPUT, [b][oy][cout / 16][out_w][112 B] -- what dd_dconv_split_rows would make
  // of y, written from the epilogue's registers so that the next layer's split pass (read 4 B, write 6 B per element) disappears.
  // abl (template): timing ablations of DD_TIMING_DIAG builds (results are then wrong; 0 in every other build): 1 no MFMAs, 2 no DMA after a
  // task's first stage, 4 no shift-add passes in the epilogue, 8 no operand reads after a stage's first
  constexpr bool ONE_MT = MODE == 0;
  constexpr int TW = 32, NE = 16, P = 40, HALO = D * (K - 1);
  constexpr int AROW = IWP * SP_PXB;                       // bytes of the A image of a stage (one input row, one chunk)
  constexpr int BST = K * 3 * 1024;                        // bytes of its B image (K tap columns x 3 planes x 64 lanes x 16 B)
  constexpr int STAGE = AROW + BST;
  constexpr int NA = AROW / 1024, NB = K * 3, NI = NA + NB;      // 1 KB wave-instructions of a stage fill
  static_assert(AROW % 1024 == 0, "the A image is filled in whole 1 KB wave-instructions");
  constexpr int IMG = (IWP + HALO) * P * 4;
  constexpr int LDSB = NBUF * STAGE > IMG ? NBUF * STAGE : IMG;
  static_assert(NBUF == 2 || NBUF == 3, "two buffers: the next stage's fill must land inside this stage; three: it has two");
  static_assert(LDSB <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char lds[LDSB];
  using acc_t = f32x16;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NC = d.cin >> 4, NTC = (d.cout + 31) >> 5;
  const int rows_max = (d.out_h + D - 1) / D;
  const int n_mt = (d.in_w + TW - 1) / TW;
  const int row_bytes = d.in_w * SP_PXB;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;

  // this wave's tiles: linear index L = m-tile * K + kx (MODE 0: m-tile = wave, kx = slot).  Slot -> (valid, m-tile, kx) is
  // recomputed where it is used from a wave id the compiler cannot see through (`wv`, laundered per use): kept as arrays the 3 x
  // NSLOT loop-invariant scalars live across the whole persistent loop and spill (MODE 1: 206 SGPRs parked in VGPR lanes, 72 B of
  // scratch on top)
  auto slot = [&](int wv, int i, bool& ok, int& mt, int& kx) {
    if constexpr (ONE_MT) {
      ok = true; mt = wv; kx = i;
    } else {
      const int L = wv * NSLOT + i;
      ok = L < n_mt * K;
      mt = ok ? L / K : 0;
      kx = ok ? L - mt * K : 0;
    }
  };
  const int a_lane = (lane & 31) * SP_PXB + (lane >> 5) * 16;      // this lane's A fragment inside an m-tile's 32 pixels
  const int ioff_lane = 4 * (lane >> 5) * P + (lane & 31);         // its accumulator element 0 in the output row image

  // ---- tasks: as dconv_tfwd_kernel (XCD = blockIdx % 8 owns the x-th eighth of the (image, residue, phase row, column tile) list)
  const int per_x = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7;
  const int per_img = D * rows_max * NTC;
  const long len = (long)d.batch * per_img;
  const long seg1 = len * (xcd + 1) / 8;
  auto decode = [&](long t, SpTask& k) -> bool {
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
  auto next_task = [&](long t, SpTask& k) -> long {
    while (t < seg1 && !decode(t, k)) t += per_x;
    return t < seg1 ? t : seg1;
  };

  // ---- stage fill by LDS-DMA: wave-instruction j of the stage copies 1 KB; lanes past the end of the row read zeros (the buffer
  // descriptor ends with the row), which is what the pixels of a ragged last m-tile must hold
  // piece i of a wave's share of a stage fill (wave-instruction j = wave + 8 i of the stage's NI); i = -1: the whole share
  constexpr int NPW = (NI + 7) / 8;
  auto fill = [&](int buf, const SpTask& k, int q, int ky, int only) {
    const long arow = ((long)(k.b * d.in_h + k.ry + D * ky) * NC + q) * row_bytes;
    const __amdgpu_buffer_rsrc_t ra = dd_rsrc(xs + arow, row_bytes);
    const __amdgpu_buffer_rsrc_t rb = dd_rsrc(wp + ((long)(q * K + ky) * NTC + k.nt) * BST, BST);
    char* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (only >= 0 && i != only) continue;
      const int j = wave + 8 * i;
      if (j < NA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, lane * 16, j * 1024, 0, 0);
      else if (j < NI)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(base + AROW + (j - NA) * 1024), 16, lane * 16,
                                                 (j - NA) * 1024, 0, 0);
    }
  };
  SpTask cur;
  long t = next_task(len * xcd / 8 + (blockIdx.x >> 3), cur);
  while (t < seg1) {
    const int ky0 = cur.ky0, ky1 = cur.ky1;
    acc_t acc[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[i][e] = 0.f;

    if (ky1 >= ky0) {
      // NBUF - 1 stages are in flight ahead of the one being multiplied.  A wave's vector-memory counter retires in order, so
      // "everything but the pieces of the newest fill" is a counted wait: this wave issued `mine` pieces per fill.
      const int mine = (NI - 1 - wave) / 8 + 1;
      auto wait_all_but_newest = [&](bool newest_issued) {
        if (!newest_issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (mine == NPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW - 1) : "memory");
      };
      const int nky = ky1 - ky0 + 1, nstage = NC * nky;
      int par = 0;
      // (q, ky) of the stage being filled, NBUF - 1 ahead of the one being multiplied
      int fq = 0, fky = ky0, fbuf = 0, filled = 0;
      auto fill_next = [&]() {
        fill(fbuf, cur, fq, fky, -1);
        ++filled;
        fbuf = fbuf + 1 == NBUF ? 0 : fbuf + 1;
        if (++fky > ky1) { fky = ky0; ++fq; }
      };
      fill_next();
      if (NBUF == 3 && nstage > 1) fill_next();
      wait_all_but_newest(NBUF == 3 && nstage > 1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      for (int s = 0; s < nstage; ++s) {
        const char* sb = lds + par * STAGE;
        const char* bp = sb + AROW + lane * 16;
        // Operand reads run ONE SLOT AHEAD of the MFMAs that consume them (left to itself the compiler issues each slot's reads
        // right in front of its MFMAs and waits out the LDS latency three times a slot: measured 2x the MFMA time).  The stage's
        // first reads go out before the next stage's fill is issued.  (One DMA piece per slot instead of all seven here: 5.42 ->
        // 5.71 ms, dropped.)
        bf16x8 A[3], An[3], Bc[3], Bn[3];
        int wv = wave;
        asm volatile("" : "+s"(wv));
        auto lda = [&](int i, bf16x8(&a)[3]) {
          bool ok; int mt, kx;
          slot(wv, i, ok, mt, kx);
          const char* ap = sb + mt * (32 * SP_PXB) + a_lane;
          a[0] = *(const bf16x8*)ap;
          a[1] = *(const bf16x8*)(ap + 32);
          a[2] = *(const bf16x8*)(ap + 64);
        };
        auto ldb = [&](int i, bf16x8(&b)[3]) {
          bool ok; int mt, kx;
          slot(wv, i, ok, mt, kx);
          const char* bq = bp + kx * 3072;
          b[0] = *(const bf16x8*)bq;
          b[1] = *(const bf16x8*)(bq + 1024);
          b[2] = *(const bf16x8*)(bq + 2048);
        };
        auto valid = [&](int i) { bool ok; int mt, kx; slot(wv, i, ok, mt, kx); return ok; };
        lda(0, A);
        ldb(0, Bc);
        __builtin_amdgcn_sched_barrier(0);
        const bool dma = filled < nstage && !(abl & 2);
        if (dma) fill_next();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          if (i + 1 < NSLOT && valid(i + 1) && !(abl & 8)) {
            ldb(i + 1, Bn);
            if constexpr (!ONE_MT) lda(i + 1, An);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (valid(i) && !(abl & 1)) {                       // hi = [0], mid = [1], lo = [2]; smallest products first
            acc[i] = SP_MFMA(A[2], Bc[0], acc[i]);
            acc[i] = SP_MFMA(A[0], Bc[2], acc[i]);
            acc[i] = SP_MFMA(A[1], Bc[1], acc[i]);
            acc[i] = SP_MFMA(A[1], Bc[0], acc[i]);
            acc[i] = SP_MFMA(A[0], Bc[1], acc[i]);
            acc[i] = SP_MFMA(A[0], Bc[0], acc[i]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            Bc[pl] = Bn[pl];
            if constexpr (!ONE_MT) A[pl] = An[pl];
          }
        }
        // the NEXT stage's fill has landed (this wave's share; the barrier adds the others'), every wave is done with buffer `par`
        if (abl & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else wait_all_but_newest(NBUF == 3 && filled > s + 2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        par = par + 1 == NBUF ? 0 : par + 1;
      }
    }

    // ---- epilogue (as dconv_tfwd_kernel): the K partial rows are added into one output row image at their shifts.  Tap column
    // K-1 (shift 0) goes first and is STORED, the others are added in K-1 barrier-separated passes: a fixed order.
    float* img = (float*)lds;
    constexpr int LPP = 8;
    constexpr int WIT = ((IWP + HALO) * LPP + SP_THREADS - 1) / SP_THREADS;
    const int c4 = 4 * (tid % LPP), cch = 32 * cur.nt + c4;
    f32x4 bvec = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((epi == DD_EPI_BIAS || epi == DD_EPI_BIAS_RELU) && cch < d.cout) bvec = *(const f32x4*)&bias[cch];
    {
      if (tid < HALO * (P / 4)) *(f32x4*)&img[n_mt * TW * P + tid * 4] = f32x4{0.f, 0.f, 0.f, 0.f};
      static_assert(HALO * (P / 4) <= SP_THREADS, "one zeroing store per thread");
#pragma unroll
      for (int pass = K - 1; pass >= ((abl & 4) ? K - 1 : 0); --pass) {
        const int shift = D * (K - 1 - pass);
        int wv = wave;
        asm volatile("" : "+s"(wv));
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
          bool ok; int mt, kx;
          slot(wv, i, ok, mt, kx);
          if (ONE_MT ? i == pass : (ok && kx == pass)) {
            float* p0 = img + ioff_lane + mt * (TW * P);
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
        sp_barrier();
      }
    }
    {
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cur.b * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const int base = ((cur.oy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff + cch;
      f32x4 v[WIT];
#pragma unroll
      for (int j = 0; j < WIT; ++j) {
        const int px = (tid + SP_THREADS * j) / LPP;
        v[j] = px < d.out_w ? *(const f32x4*)&img[px * P + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < WIT; ++j) {
        const int px = (tid + SP_THREADS * j) / LPP;
        f32x4 o = v[j] + bvec;
        if (epi == DD_EPI_BIAS_RELU) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        const int off = (px < d.out_w && cch < d.cout) ? (base + px * d.out_cstore) * 4 : -16;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), ys, off, 0, 0);
        if (yplanes) {      // this thread's four channels of the pixel: 8 bytes per plane
          const int ncp = d.cout >> 4;
          const __amdgpu_buffer_rsrc_t ps = dd_rsrc(yplanes + (long)cur.b * d.out_h * ncp * d.out_w * SP_PXB, d.out_h * ncp * d.out_w * SP_PXB);
          const int poff = (px < d.out_w && cch < d.cout) ? ((cur.oy * ncp + (cch >> 4)) * d.out_w + px) * SP_PXB + (cch & 15) * 2 : -16;
          unsigned h[4], m[4], l[4];
          sp_split(o.x, h[0], m[0], l[0]); sp_split(o.y, h[1], m[1], l[1]); sp_split(o.z, h[2], m[2], l[2]); sp_split(o.w, h[3], m[3], l[3]);
          typedef __attribute__((ext_vector_type(2))) unsigned u32x2s;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2s{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]}, ps, poff, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2s{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]}, ps, poff < 0 ? poff : poff + 32, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2s{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]}, ps, poff < 0 ? poff : poff + 64, 0, 0);
        }
      }
    }
    sp_barrier();
    t = next_task(t + per_x, cur);
  }
}

// =====================================================================================================================
// The DATA GRADIENT of the same layers with split products: a plain dilated convolution (pad 0, out = in - d(k-1)) in gather form, as
// dconv_gfwd_kernel: one output row per workgroup, wave w = m-tile w (32 output pixels) x all NTC column tiles, one accumulator per
// column tile (no shift-add epilogue).  The A operand of a (chunk, tap row) -- the seven tap-shifted fragments of the wave's pixels, three
// planes each -- is read ONCE from LDS into registers (84 VGPRs) and multiplied with the B images of the NTC column tiles in turn; the B
// image of one (chunk, tap row, column tile) is 21 KB, so the stage is cut per column tile: two A buffers (alternating per (chunk, tap
// row)), two B buffers (alternating per sub-stage), everything filled by LDS-DMA one sub-stage ahead, one s_barrier per sub-stage of 42
// MFMAs per wave.  NW = 8 waves (rows up to 256 output pixels: up_conv_1), 10 (up to 320: up_conv_2; SIMDs then hold 3 / 3 / 2 / 2 waves) or 11 (up to
// 352: up_conv_3, one column tile).
template <int K, int D, int NTC, int NW, int IWP>
__global__ __launch_bounds__(NW * 64) void dconv_sgfwd_kernel(const char* __restrict__ xs, const char* __restrict__ wp,
                                                              const float* __restrict__ msk, float* __restrict__ y, char* __restrict__ yplanes,
                                                              const dd_gconv_desc d, int epi) {
  constexpr int AROW = IWP * SP_PXB, BST = K * 3 * 1024;
  constexpr int NA = AROW / 1024, NB = K * 3;
  static_assert(AROW % 1024 == 0, "the A image is filled in whole 1 KB wave-instructions");
  static_assert(2 * AROW + 2 * BST <= 160 * 1024, "LDS");
  static_assert(NW * 32 + D * (K - 1) <= IWP, "the last wave's last tap stays inside the A image");
  __shared__ __attribute__((aligned(1024))) char lds[2 * AROW + 2 * BST];
  char* const abuf = lds;
  char* const bbuf = lds + 2 * AROW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31;
  const int NC = d.cin >> 4;
  const int rows_max = (d.out_h + D - 1) / D;
  const int n_mt = (d.out_w + 31) >> 5;
  const int row_bytes = d.in_w * SP_PXB;
  const int out_bytes = d.omem_h * d.omem_w * d.out_cstore * 4;
  const int a_lane = (wave * 32 + n) * SP_PXB + (lane >> 5) * 16;      // this lane's A fragment of tap column 0 inside a row image

  const int per_x = gridDim.x >> 3;
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

  auto fill_a = [&](int buf, int b, int row, int q) {
    const long arow = ((long)(b * d.in_h + row) * NC + q) * row_bytes;
    const __amdgpu_buffer_rsrc_t ra = dd_rsrc(xs + arow, row_bytes);
#pragma unroll
    for (int i = 0; i < (NA + NW - 1) / NW; ++i) {
      const int j = wave + NW * i;
      if (j < NA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(abuf + buf * AROW + j * 1024), 16, lane * 16, j * 1024, 0, 0);
    }
  };
  auto fill_b = [&](int buf, int q, int ky, int nt) {
    const __amdgpu_buffer_rsrc_t rb = dd_rsrc(wp + ((long)(q * K + ky) * NTC + nt) * BST, BST);
#pragma unroll
    for (int i = 0; i < (NB + NW - 1) / NW; ++i) {
      const int j = wave + NW * i;
      if (j < NB)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(bbuf + buf * BST + j * 1024), 16, lane * 16, j * 1024, 0, 0);
    }
  };

  int cb, coy;
  long t = next_task(len * xcd / 8 + (blockIdx.x >> 3), cb, coy);
  while (t < seg1) {
    f32x16 acc[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    fill_a(0, cb, coy, 0);
    fill_b(0, 0, 0, 0);
    sp_barrier();
    int apar = 0, bpar = 0;
    bf16x8 A[K][3];
    for (int q = 0; q < NC; ++q) {
      for (int ky = 0; ky < K; ++ky) {
        const bool last_qk = q == NC - 1 && ky == K - 1;
        const int qn = ky == K - 1 ? q + 1 : q, kyn = ky == K - 1 ? 0 : ky + 1;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
          const char* bp = bbuf + bpar * BST + lane * 16;
          bf16x8 Bc[3], Bn[3];
          if (nt == 0) {      // the (chunk, tap row)'s A fragments: read once, used for every column tile
            const char* ap = abuf + apar * AROW + a_lane;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              A[kx][0] = *(const bf16x8*)(ap + kx * (D * SP_PXB));
              A[kx][1] = *(const bf16x8*)(ap + kx * (D * SP_PXB) + 32);
              A[kx][2] = *(const bf16x8*)(ap + kx * (D * SP_PXB) + 64);
            }
          }
          Bc[0] = *(const bf16x8*)bp;
          Bc[1] = *(const bf16x8*)(bp + 1024);
          Bc[2] = *(const bf16x8*)(bp + 2048);
          __builtin_amdgcn_sched_barrier(0);
          // the next sub-stage's images: the next column tile's B, or (last column tile) the next (chunk, tap row)'s A and first B
          if (nt + 1 < NTC) {
            fill_b(bpar ^ 1, q, ky, nt + 1);
          } else if (!last_qk) {
            fill_a(apar ^ 1, cb, coy + D * kyn, qn);
            fill_b(bpar ^ 1, qn, kyn, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            if (kx + 1 < K) {
              Bn[0] = *(const bf16x8*)(bp + (kx + 1) * 3072);
              Bn[1] = *(const bf16x8*)(bp + (kx + 1) * 3072 + 1024);
              Bn[2] = *(const bf16x8*)(bp + (kx + 1) * 3072 + 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[nt] = SP_MFMA(A[kx][2], Bc[0], acc[nt]);      // hi = [0], mid = [1], lo = [2]; smallest products first
            acc[nt] = SP_MFMA(A[kx][0], Bc[2], acc[nt]);
            acc[nt] = SP_MFMA(A[kx][1], Bc[1], acc[nt]);
            acc[nt] = SP_MFMA(A[kx][1], Bc[0], acc[nt]);
            acc[nt] = SP_MFMA(A[kx][0], Bc[1], acc[nt]);
            acc[nt] = SP_MFMA(A[kx][0], Bc[0], acc[nt]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Bc[pl] = Bn[pl];
          }
          sp_barrier();
          bpar ^= 1;
        }
        apar ^= 1;
      }
    }

    // ---- write-out (as dconv_gfwd_kernel: all mask values of a tile requested before the first is used).  Lane-derived values come from
    // a fresh lane id (dd_fresh_lane) so that nothing of the epilogue's addressing lives across the MFMA loop (170 registers at 10-11 waves)
    {
      const int fl = dd_fresh_lane();
      const int n = fl & 31, lane = fl;
      const __amdgpu_buffer_rsrc_t ys = dd_rsrc(y + (long)cb * d.omem_h * d.omem_w * d.out_cstore, out_bytes);
      const __amdgpu_buffer_rsrc_t ms = dd_rsrc(msk ? msk + (long)cb * d.omem_h * d.omem_w * d.out_cstore : y, msk ? out_bytes : 0);
      const bool masked = epi == DD_EPI_RELU_MASK;
      const int base = ((coy + d.ooff_h) * d.omem_w + d.ooff_w) * d.out_cstore + d.out_coff;
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) {
        const int ch = nt * 32 + n;
        const bool pass = d.out_coff + ch >= d.mask_pass_lo && d.out_coff + ch < d.mask_pass_hi;
        int off[16];
        float mv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int xo = wave * 32 + dd_acc_row(r, lane);
          const bool ok = wave < n_mt && xo < d.out_w && ch < d.cout;
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
          acc[nt][r] = mv[r] > 0.f ? acc[nt][r] : 0.f;      // masked in place: the plane stores below need no mask and no offset kept alive
          dd_bstore1(ys, off[r], acc[nt][r]);
        }
        if (yplanes) {      // the output's bf16 planes (see dconv_stfwd_kernel): a lane holds ONE channel of 16 pixels -> 2-byte stores
          const int ncp = d.cout >> 4;
          const __amdgpu_buffer_rsrc_t ps = dd_rsrc(yplanes + (long)cb * d.out_h * ncp * d.out_w * SP_PXB, d.out_h * ncp * d.out_w * SP_PXB);
          const int pbase = ((coy * ncp + (ch >> 4)) * d.out_w) * SP_PXB + (ch & 15) * 2;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int xo = wave * 32 + dd_acc_row(r, lane);
            const int poff = (wave < n_mt && xo < d.out_w && ch < d.cout) ? pbase + xo * SP_PXB : (int)0x80000000;
            unsigned hi, mid, lo;
            sp_split(acc[nt][r], hi, mid, lo);
            __builtin_amdgcn_raw_buffer_store_b16((short)(hi >> 16), ps, poff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((short)(mid >> 16), ps, poff + 32, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((short)(lo >> 16), ps, poff + 64, 0, 0);
          }
        }
      }
    }
    sp_barrier();
    t = next_task(t + per_x, cb, coy);
  }
}

// =====================================================================================================================
// The WEIGHT GRADIENT of the same layers with split products:
//     dW[c][o][ky][kx] = sum over images and input pixels (iy, ix) of  x[iy][ix][c] * g[iy + D*ky][ix + D*kx][o]
// GEMM per tap: M = Cin, N = Cout, K = pixels -- both operands need PIXELS along k, eight consecutive ones per lane.  They come out of
// the same [pixel][plane][16 channels] images the forward and the data gradient use (xs of the layer's input, gs of dL/dy: no further
// split pass) through the hardware transpose read ds_read_b64_tr_b16: per 16-lane group a block of 4 pixels x 16 channels, delivered
// channel-major, so a tap's pixel shift is just a row offset of the read.  As dconv_wgrad_kernel a workgroup owns ONE tap row ky and a
// range of input rows; per step a 32-pixel piece of an x row (all chunks) and the matching 74-pixel piece of the g row iy + D*ky sit in
// LDS (two buffers, filled by LDS-DMA one step ahead); wave w < K owns tap column kx = w and its (Cin/32) x (Cout/32) accumulator tiles
// (the eighth wave only helps with the fills); partials per workgroup, fixed-order fp64 second stage (deterministic).
typedef __attribute__((ext_vector_type(4))) short sp_s16x4;
typedef sp_s16x4 __attribute__((address_space(3))) * sp_lds_s16x4_ptr;
__device__ __forceinline__ sp_s16x4 sp_tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((sp_lds_s16x4_ptr)p); }
__device__ __forceinline__ bf16x8 sp_join(sp_s16x4 a, sp_s16x4 b) {
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int C, int O>
struct SwGeom {
  static constexpr int K = 7, D = 7, XT = 32, GT = XT + D * (K - 1);
  static constexpr int NCX = C / 16, NCO = O / 16, MT = C / 32, NTO = O / 32, TPW = MT * NTO;
  static constexpr int PX = (XT * SP_PXB + 1023) / 1024, PG = (GT * SP_PXB + 1023) / 1024;      // 1 KB DMA pieces per chunk piece
  static constexpr int RX = PX * 1024, RG = PG * 1024;                                          // LDS bytes per chunk region
  static constexpr int BUF = NCX * RX + NCO * RG;
  static constexpr int NP = NCX * PX + NCO * PG;
  static_assert(C % 32 == 0 && O % 32 == 0 && 2 * BUF <= 160 * 1024, "channel tiles / LDS");
};

template <int C, int O>
__global__ __launch_bounds__(SP_THREADS) void dconv_swgrad_kernel(const char* __restrict__ xs, const char* __restrict__ gs,
                                                                  float* __restrict__ part, int wg_per_ky, int B, int H, int W, int gh, int gw) {
  using G = SwGeom<C, O>;
  constexpr int K = G::K, D = G::D;
  __shared__ __attribute__((aligned(1024))) char lds[2 * G::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ky = blockIdx.x / wg_per_ky, wl = blockIdx.x - ky * wg_per_ky;
  const long rows = (long)B * H;
  const long r0 = rows * wl / wg_per_ky, r1 = rows * (wl + 1) / wg_per_ky;
  const int nxt = (W + G::XT - 1) / G::XT;
  const long nsteps = (r1 - r0) * nxt;

  // transpose-read coordinates of this lane: group (k half, channel half), row q and element quad p inside the 4 x 16 block
  const int grp = lane >> 4, e = lane & 15, q = e >> 2, p = e & 3;
  const int hk = grp >> 1, half = grp & 1;
  const int a_lane = half * G::RX + (8 * hk + q) * SP_PXB + 8 * p;
  const int b_lane = G::NCX * G::RX + half * G::RG + (8 * hk + q + D * wave) * SP_PXB + 8 * p;      // tap column kx = wave

  f32x16 acc[G::TPW];
#pragma unroll
  for (int t = 0; t < G::TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // the step being FILLED, as 32-bit scalars advanced by hand (a 64-bit `s / nxt` is expanded on the vector ALU, the compiler then treats
  // everything derived from it -- descriptors, LDS addresses -- as divergent and wraps each DMA in a readfirstlane loop)
  int f_b = (int)(r0 / H), f_iy = (int)(r0 - (long)f_b * H), f_xt = 0;
  auto fill = [&](int buf) {
    const int b = __builtin_amdgcn_readfirstlane(f_b), iy = __builtin_amdgcn_readfirstlane(f_iy), xt = __builtin_amdgcn_readfirstlane(f_xt);
    const int x0 = xt * G::XT, gy = iy + D * ky;
    const int xlen = sp_uni(min(G::XT, W - x0) * SP_PXB), glen = sp_uni(max(0, min(G::GT, gw - x0)) * SP_PXB);
    const char* xrow = xs + ((long)(b * H + iy) * G::NCX * W + x0) * SP_PXB;
    const char* grow = gs + ((long)(b * gh + gy) * G::NCO * gw + x0) * SP_PXB;
    char* base = lds + buf * G::BUF;
#pragma unroll
    for (int i = 0; i < (G::NP + 7) / 8; ++i) {
      const int j = wave + 8 * i;
      if (j < G::NCX * G::PX) {
        const int c = j / G::PX, pc = j - c * G::PX;
        const __amdgpu_buffer_rsrc_t r = dd_rsrc(sp_uni(xrow + (long)c * W * SP_PXB), xlen);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(base + c * G::RX + pc * 1024), 16, lane * 16, pc * 1024, 0, 0);
      } else if (j < G::NP) {
        const int jj = j - G::NCX * G::PX, c = jj / G::PG, pc = jj - c * G::PG;
        const __amdgpu_buffer_rsrc_t r = dd_rsrc(sp_uni(grow + (long)c * gw * SP_PXB), glen);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(base + G::NCX * G::RX + c * G::RG + pc * 1024), 16, lane * 16,
                                                 pc * 1024, 0, 0);
      }
    }
    if (++f_xt == nxt) {
      f_xt = 0;
      if (++f_iy == H) { f_iy = 0; ++f_b; }
    }
  };

  if (nsteps > 0) fill(0);
  sp_barrier();
  int par = 0;
  for (long s = 0; s < nsteps; ++s) {
    // all of the step's operand reads BEFORE the next step's DMA is issued (a transpose read behind an LDS-DMA in program order gets an
    // s_waitcnt vmcnt(0) in front of it: the fill would not overlap the multiplies at all), the MFMAs behind it
    bf16x8 A[2][G::MT][3], Bv[2][G::NTO][3];
    if (wave < K) {
      const char* lb = lds + par * G::BUF;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {      // the piece's 32 pixels = two k16 blocks
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const char* a = lb + a_lane + mt * (2 * G::RX) + (16 * kb) * SP_PXB + pl * 32;
            A[kb][mt][pl] = sp_join(sp_tr_read(a), sp_tr_read(a + 4 * SP_PXB));
          }
#pragma unroll
        for (int nt = 0; nt < G::NTO; ++nt)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const char* bq = lb + b_lane + nt * (2 * G::RG) + (16 * kb) * SP_PXB + pl * 32;
            Bv[kb][nt][pl] = sp_join(sp_tr_read(bq), sp_tr_read(bq + 4 * SP_PXB));
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nsteps) fill(par ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < K) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < G::NTO; ++nt) {
            f32x16& c = acc[mt * G::NTO + nt];
            c = SP_MFMA(A[kb][mt][2], Bv[kb][nt][0], c);      // hi = [0], mid = [1], lo = [2]; smallest products first
            c = SP_MFMA(A[kb][mt][0], Bv[kb][nt][2], c);
            c = SP_MFMA(A[kb][mt][1], Bv[kb][nt][1], c);
            c = SP_MFMA(A[kb][mt][1], Bv[kb][nt][0], c);
            c = SP_MFMA(A[kb][mt][0], Bv[kb][nt][1], c);
            c = SP_MFMA(A[kb][mt][0], Bv[kb][nt][0], c);
          }
    }
    sp_barrier();
    par ^= 1;
  }
  if (wave < K) {
#pragma unroll
    for (int t = 0; t < G::TPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[((((long)blockIdx.x * K + wave) * G::TPW + t) * 16 + r) * 64 + lane] = acc[t][r];
  }
}

// One block = one accumulator tile of one tap (1024 threads = 16 registers x 64 lanes): sums the tap row's workgroups in fp64, in a fixed
// order, and scatters to the ConvTranspose2d weight layout [Cin][Cout][K][K].
template <int C, int O>
__global__ __launch_bounds__(1024) void dconv_swgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int wg_per_ky, int accumulate) {
  using G = SwGeom<C, O>;
  constexpr int K = G::K;
  const int t = blockIdx.x % G::TPW, kx = (blockIdx.x / G::TPW) % K, ky = blockIdx.x / (G::TPW * K);
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int w = 0; w < wg_per_ky; ++w) s += (double)part[((((long)(ky * wg_per_ky + w) * K + kx) * G::TPW + t) * 16 + r) * 64 + lane];
  const int mt = t / G::NTO, nt = t - mt * G::NTO;
  const int c = mt * 32 + dd_acc_row(r, lane), o = nt * 32 + (lane & 31);
  const long wi = (((long)c * O + o) * K + ky) * K + kx;
  dw[wi] = accumulate ? dw[wi] + (float)s : (float)s;
}

bool sp_common_ok(const dd_gconv_desc* d) {
  if (!dd_dconv_desc_ok(d)) return false;
  if (d->kh != 7 || d->kw != 7 || d->dil_h != 7 || d->dil_w != 7) return false;
  if (d->cin % 16 || d->cout <= 16 || d->cout % 4 || d->out_coff % 4 || d->out_cstore % 4 || d->cout > 96) return false;
  if ((long)d->in_w * SP_PXB * (d->cin / 16) * d->in_h * d->batch >= (1L << 40)) return false;
  return true;
}

// the forward: the full transposed form (pad = d(k-1), out = in + d(k-1))
bool sp_fwd_ok(const dd_gconv_desc* d) {
  if (!sp_common_ok(d)) return false;
  if (d->pad_h != 42 || d->pad_w != 42) return false;
  if (d->out_h < d->in_h + 42 || d->out_w < d->in_w + 42) return false;
  if (d->in_w > 320 || d->out_w > ((d->in_w + 31) / 32) * 32 + 42) return false;
  return true;
}

// the data gradient: a plain dilated convolution (pad 0, out = in - d(k-1)), rows of at most 352 output pixels
bool sp_dgrad_ok(const dd_gconv_desc* d) {
  if (!sp_common_ok(d)) return false;
  if (d->pad_h != 0 || d->pad_w != 0) return false;
  if (d->out_h > d->in_h - 42 || d->out_w > d->in_w - 42) return false;
  if (d->out_w > 352 || d->in_w > 448) return false;
  if ((d->out_w > 256 || d->in_w > 320) && d->cout > 64) return false;      // the 10-wave form holds at most two column tiles in 170 registers
  if ((d->out_w > 320 || d->in_w > 384) && d->cout > 32) return false;      // the 11-wave form (up_conv_3: 340 output pixels) one
  return true;
}

bool sp_layer_ok(const dd_gconv_desc* d) { return sp_fwd_ok(d) || sp_dgrad_ok(d); }

}  // namespace

extern "C" {

int32_t dd_dconv_split_supported(const dd_gconv_desc* d) { return d && sp_layer_ok(d) ? 1 : 0; }

int64_t dd_dconv_split_input_bytes(const dd_gconv_desc* d) {
  if (!d || !sp_layer_ok(d)) return dd_fail(DD_ERR_UNSUPPORTED, "dconv_split_input_bytes: unsupported layer"), -1;
  return (int64_t)d->batch * d->in_h * (d->cin / 16) * d->in_w * SP_PXB;
}

int64_t dd_dconv_split_packed_bytes(const dd_gconv_desc* d) {
  if (!d || !sp_layer_ok(d)) return dd_fail(DD_ERR_UNSUPPORTED, "dconv_split_packed_bytes: unsupported layer"), -1;
  return (int64_t)(d->cin / 16) * 7 * ((d->cout + 31) / 32) * 7 * 3 * 1024;
}

int dd_dconv_split_input(const float* x, void* xs, const dd_gconv_desc* d, void* stream) {
  DD_REQUIRE(d && sp_layer_ok(d), DD_ERR_UNSUPPORTED, "dconv_split_input: unsupported layer");
  DD_REQUIRE(x && xs, DD_ERR_BAD_ARG, "dconv_split_input: NULL pointer");
  DD_REQUIRE(d->in_coff % 4 == 0 && d->in_cstore % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)xs & 15) == 0, DD_ERR_BAD_ARG,
             "dconv_split_input: 16-byte alignment");
  const long rows = (long)d->batch * d->in_h;
  const long total = rows * (d->cin / 16) * d->in_w * 2;
  hipLaunchKernelGGL(split_input_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0, (hipStream_t)stream, x,
                     (char*)xs, rows, d->in_w, d->in_cstore, d->in_coff, d->cin / 16);
  DD_LAUNCH_CHECK("dconv_split_input");
  return 0;
}

int dd_dconv_split_pack(const float* w, void* packed, const dd_gconv_desc* d, int64_t w_off, int64_t sn, int64_t sc, int32_t flip,
                        int32_t n_real, int32_t c_real, void* stream) {
  DD_REQUIRE(d && sp_layer_ok(d), DD_ERR_UNSUPPORTED, "dconv_split_pack: unsupported layer");
  DD_REQUIRE(w && packed, DD_ERR_BAD_ARG, "dconv_split_pack: NULL pointer");
  DD_REQUIRE(n_real > 0 && n_real <= d->cout && c_real > 0 && c_real <= d->cin, DD_ERR_BAD_ARG, "dconv_split_pack: n_real/c_real");
  const int NT = (d->cout + 31) / 32;
  const long total = (long)(d->cin / 16) * 7 * NT * 7 * 512;
  hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed,
                     d->cin / 16, NT, 7, (long)w_off, (long)sn, (long)sc, flip, n_real, c_real);
  DD_LAUNCH_CHECK("dconv_split_pack");
  return 0;
}

/* fp32 NHWC rows -> the three-plane bf16 image, without a layer descriptor: `rows` image rows of `w` pixels, channels [coff, coff + c)
 * of `cstore`; xs = rows * (c / 16) * w * 112 bytes.  (dd_dconv_split_input is this for a descriptor's input.) */
int dd_dconv_split_rows(const float* x, void* xs, int64_t rows, int32_t w, int32_t cstore, int32_t coff, int32_t c, void* stream) {
  DD_REQUIRE(x && xs && rows > 0 && w > 0 && c > 0 && c % 16 == 0 && coff >= 0 && coff % 4 == 0 && cstore % 4 == 0 && coff + c <= cstore,
             DD_ERR_BAD_ARG, "dconv_split_rows: bad argument");
  DD_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)xs & 15) == 0, DD_ERR_BAD_ARG, "dconv_split_rows: 16-byte alignment");
  const long total = rows * (c / 16) * w * 2;
  hipLaunchKernelGGL(split_input_kernel, dim3((unsigned)min((total + 255) / 256, (long)DD_NUM_CU * 16)), dim3(256), 0, (hipStream_t)stream, x,
                     (char*)xs, (long)rows, w, cstore, coff, c / 16);
  DD_LAUNCH_CHECK("dconv_split_rows");
  return 0;
}

int32_t dd_dconv_wgrad_split_supported(int32_t k, int32_t dil, int32_t cin, int32_t cout) {
  return (k == 7 && dil == 7 && ((cin == 96 && cout == 64) || (cin == 64 && cout == 32))) ? 1 : 0;
}

// workgroups per tap row: one resident round.  96 -> 64 holds 120 KB of LDS: one workgroup per CU.  64 -> 32 (70 KB, 88 registers) fits TWO per
// CU, and needs them: with two accumulator tiles per wave a step is 24 MFMAs behind 36 transpose reads and a barrier -- one workgroup per CU
// ties the exact kernel (4.82 against 4.85 ms), two take 3.34
static int sp_wg_per_ky(int cin) { return max(1, dd_cu_budget_internal() / 7) * (cin == 64 ? 2 : 1); }

int64_t dd_dconv_wgrad_split_workspace_bytes(int32_t cin, int32_t cout) {
  if (!dd_dconv_wgrad_split_supported(7, 7, cin, cout)) return dd_fail(DD_ERR_UNSUPPORTED, "dconv_wgrad_split_workspace_bytes: unsupported layer"), -1;
  return (int64_t)sp_wg_per_ky(cin) * 7 * 7 * (cin / 32) * (cout / 32) * 16 * 64 * 4;
}

/* xs: split image of the layer's input x [batch, h, w, cin]; gs: split image of g = dL/dy [batch, gh, gw, cout] (gh >= h + 42,
 * gw >= w + 42); dw [cin][cout][7][7] (ConvTranspose2d's layout), accumulate != 0 adds. */
int dd_dconv_wgrad_split(const void* xs, const void* gs, float* dw, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t gh, int32_t gw,
                         int32_t cout, int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  DD_REQUIRE(dd_dconv_wgrad_split_supported(7, 7, cin, cout), DD_ERR_UNSUPPORTED, "dconv_wgrad_split: unsupported layer %d -> %d", cin, cout);
  DD_REQUIRE(xs && gs && dw && workspace, DD_ERR_BAD_ARG, "dconv_wgrad_split: NULL pointer");
  DD_REQUIRE(batch > 0 && h > 0 && w > 0 && gh >= h + 42 && gw >= w + 42, DD_ERR_BAD_ARG, "dconv_wgrad_split: sizes");
  const int wg = sp_wg_per_ky(cin);
  DD_REQUIRE(workspace_bytes >= (int64_t)wg * 7 * 7 * (cin / 32) * (cout / 32) * 16 * 64 * 4, DD_ERR_WORKSPACE, "dconv_wgrad_split: workspace too small");
  hipStream_t st = (hipStream_t)stream;
#define SW_LAUNCH(CC, OO)                                                                                                                  \
  do {                                                                                                                                     \
    hipLaunchKernelGGL((dconv_swgrad_kernel<CC, OO>), dim3(wg * 7), dim3(SP_THREADS), 0, st, (const char*)xs, (const char*)gs, (float*)workspace, wg, \
                       batch, h, w, gh, gw);                                                                                               \
    hipLaunchKernelGGL((dconv_swgrad_reduce<CC, OO>), dim3(7 * 7 * SwGeom<CC, OO>::TPW), dim3(1024), 0, st, (const float*)workspace, dw, wg, accumulate); \
  } while (0)
  if (cin == 96) SW_LAUNCH(96, 64); else SW_LAUNCH(64, 32);
#undef SW_LAUNCH
  DD_LAUNCH_CHECK("dconv_wgrad_split");
  return 0;
}

int dd_dconv_fwd_split(const void* xs, const void* packed, const float* bias, const float* mask, float* y, void* y_planes,
                       const dd_gconv_desc* d, int32_t epilogue, void* stream) {
  DD_REQUIRE(d && sp_layer_ok(d), DD_ERR_UNSUPPORTED, "dconv_fwd_split: unsupported layer");
  DD_REQUIRE(xs && packed && y, DD_ERR_BAD_ARG, "dconv_fwd_split: NULL pointer");
  DD_REQUIRE(!y_planes || (d->cout % 16 == 0 && ((uintptr_t)y_planes & 15) == 0 && (long)d->out_h * (d->cout / 16) * d->out_w * SP_PXB < (1L << 31)),
             DD_ERR_UNSUPPORTED, "dconv_fwd_split: y_planes needs cout % 16 == 0 and an image below 2 GB per batch element");
  const int grid = dd_cu_budget_internal() & ~7;
  DD_REQUIRE(grid >= 8, DD_ERR_UNSUPPORTED, "dconv_fwd_split: CU budget below 8");
  hipStream_t st = (hipStream_t)stream;
  if (sp_dgrad_ok(d)) {      // gather form: the data gradient (no bias; NONE or the ReLU mask of the layer's input)
    DD_REQUIRE(epilogue == DD_EPI_NONE || epilogue == DD_EPI_RELU_MASK, DD_ERR_BAD_ARG, "dconv_fwd_split: data-gradient epilogue %d", epilogue);
    DD_REQUIRE(epilogue != DD_EPI_RELU_MASK || mask, DD_ERR_BAD_ARG, "dconv_fwd_split: RELU_MASK needs a mask");
    const int ntc = (d->cout + 31) / 32;
#define SG_LAUNCH(NTC_, NW_, IWP_)                                                                                                   \
  hipLaunchKernelGGL((dconv_sgfwd_kernel<7, 7, NTC_, NW_, IWP_>), dim3(grid), dim3(NW_ * 64), 0, st, (const char*)xs, (const char*)packed, \
                     mask, y, (char*)y_planes, *d, epilogue)
    if (d->out_w <= 256 && d->in_w <= 320) {
      if (ntc == 1) SG_LAUNCH(1, 8, 320); else if (ntc == 2) SG_LAUNCH(2, 8, 320); else SG_LAUNCH(3, 8, 320);
    } else if (d->out_w <= 320 && d->in_w <= 384) {
      if (ntc == 1) SG_LAUNCH(1, 10, 384); else SG_LAUNCH(2, 10, 384);
    } else {
      SG_LAUNCH(1, 11, 448);
    }
#undef SG_LAUNCH
    DD_LAUNCH_CHECK("dconv_fwd_split (data gradient)");
    return 0;
  }
  DD_REQUIRE(epilogue == DD_EPI_NONE || epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU, DD_ERR_BAD_ARG, "dconv_fwd_split: epilogue %d",
             epilogue);
  DD_REQUIRE(!(epilogue == DD_EPI_BIAS || epilogue == DD_EPI_BIAS_RELU) || (bias && ((uintptr_t)bias & 15) == 0), DD_ERR_BAD_ARG,
             "dconv_fwd_split: bias epilogue needs a 16-byte aligned bias");
#define SP_LAUNCH_(ABL, NBUF_)                                                                                                            \
  do {                                                                                                                                    \
    if (d->in_w <= 256)                                                                                                                   \
      hipLaunchKernelGGL((dconv_stfwd_kernel<7, 7, 7, 0, 256, ABL, NBUF_>), dim3(grid), dim3(SP_THREADS), 0, st, (const char*)xs, (const char*)packed, \
                         bias, y, (char*)y_planes, *d, epilogue);                                                                         \
    else                                                                                                                                  \
      hipLaunchKernelGGL((dconv_stfwd_kernel<7, 7, 9, 1, 320, ABL, 2>), dim3(grid), dim3(SP_THREADS), 0, st, (const char*)xs, (const char*)packed, \
                         bias, y, (char*)y_planes, *d, epilogue);                                                                         \
  } while (0)
  // (three LDS buffers for the 256-wide form -- a fill has two stages to land -- measured 5.63-5.67 ms against 5.57-5.58 with two: the fills'
  // latency is not what the kernel waits for, it runs power-limited at 1.88 GHz with the matrix pipe 64 % busy; the arm is gone)
#define SP_LAUNCH(ABL) SP_LAUNCH_(ABL, 2)
#ifdef DD_TIMING_DIAG      // diagnostic builds only (build.py --diag): DD_SPLIT_ABL selects a timing ablation, the results are then wrong
  switch (getenv("DD_SPLIT_ABL") ? atoi(getenv("DD_SPLIT_ABL")) : 0) {
    case 1: SP_LAUNCH(1); break;
    case 2: SP_LAUNCH(2); break;
    case 3: SP_LAUNCH(3); break;
    case 4: SP_LAUNCH(4); break;
    case 8: SP_LAUNCH(8); break;
    case 9: SP_LAUNCH(9); break;
    case 11: SP_LAUNCH(11); break;
    case 15: SP_LAUNCH(15); break;
    default: SP_LAUNCH(0); break;
  }
#else
  SP_LAUNCH(0);
#endif
#undef SP_LAUNCH
#undef SP_LAUNCH_
  DD_LAUNCH_CHECK("dconv_fwd_split");
  return 0;
}

}  // extern "C"
