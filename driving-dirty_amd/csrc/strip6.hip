// The six strip convolutions of SpatialMappingCNN in ONE launch each way (reference spatial_bb/components.py:18-24 the layers,
// :34-65 their inputs, :70-73 the mosaic): forward (+ bias + ReLU, written straight into the 3 x 2 mosaic) and weight / bias gradient.
//
// What the reference does per camera view: bl / fl take the raw view through Conv2d(3, 32, (1, 50), stride (3, 2)); br / fr the view
// flipped in H and W through the same shape; b / f the view rotated by +-90 degrees through Conv2d(3, 32, (52, 1), stride (3, 2),
// padding 1).  Every one of them is a 1-D convolution ALONG A ROW OF THE RAW IMAGE: a (1, 50) kernel touches one input row per output
// row, and a (52, 1) kernel on a rotated image walks along what is a row of the unrotated one.  So the rotations and flips are index
// arithmetic, not data movement:
//
//   kind 0  bl, fl   raw row 3 L            out (oy = L, ox = m) = sum_t w[t] row[ 2 m + t ]                       t < 50
//   kind 1  br, fr   raw row H - 1 - 3 L    out (oy = L, ox = m) = sum_t w[t] row[ W - 1 - 2 m - t ]
//   kind 2  b        raw row 2 L - 1        out (oy = m, ox = L) = sum_t w[t] row[ W - 3 m - t ]                   t < 52, zero outside
//   kind 3  f        raw row H - 2 L        out (oy = m, ox = L) = sum_t w[t] row[ 3 m + t - 1 ]
//
// (L = line, m = position along it; rows / columns outside the image are the layers' zero padding).  Only 86 of 256 rows (kinds 0, 1)
// or 128 of 256 (kinds 2, 3) of a view are ever read -- 70 MB of the batch's 180 -- and nothing is re-laid: round 4 wrote the six views
// out as NHWC4 images (240 MB, six launches) and ran six launches of the generic gather engine over them each way, 1.9 ms per step at a
// quarter of the HBM roof.  Cin = 3: the layer is HBM-bound (452 MB per pass at bs 32), the matrix work (150 / 156 products per output)
// is what has to stay out of the way.
//
// Forward: a wave owns one line at a time.  The raw row (3 channel planes of W floats; uint8 frames: W x 3 bytes, ToTensor's /255 on
// the way) goes into the wave's own LDS strip, zero-padded on both sides so that padding taps and the positions past the end of a
// ragged last tile read zeros; A = positions (32 per MFMA tile, lane m reads row[x0 + sx m + st t]: one ds_read_b32, conflict-free for
// stride 2), B = the layer's weights, resident in 75 / 78 registers per lane for the wave's whole life (k-pair = two adjacent taps of
// one channel on the two half-waves); bias + ReLU in the epilogue, 128 contiguous bytes per pixel into the mosaic tile.  The next
// line's row is in flight in registers while the current one is multiplied.  No workgroup barrier: the strips are per wave.
// Weight gradient: the transpose -- M = output channels (A = dL/dy of the line, straight from HBM), N = taps (B from the same LDS
// strip), K = positions; 6 accumulator tiles per wave (3 channels x 2 tap halves) live across all its lines, one spare tap column
// multiplies by 1.0 and so collects the bias gradient; per-wave partials, summed in a fixed order by a second kernel (deterministic).
#include "dd_common.h"

namespace {

constexpr int S6_LPAD = 64;        // floats of zeros in front of x = 0
constexpr int S6_PITCH = 448;      // floats per channel plane of a wave's strip: x in [-64, 384)
constexpr int S6_WAVE = 3 * S6_PITCH;
constexpr int S6_MAXW = 320;       // 5 x 64 lanes per row load

struct S6Ptrs {
  const void* p[64];
};
struct S6Args {
  S6Ptrs samples;                  // per-sample base pointers: fp32 [6][3][H][W] or uint8 [6][H][W][3]
  const float* w[6];               // tile order bl, fl, b, f, br, fr: [32][3][T]
  const float* bias[6];
  float* mosaic;                   // forward: [nb][3 th][2 tw][32]
  unsigned* bits;                  // forward, optional: the mosaic's sign words [nb][3 th][2 tw] (bit c = channel c > 0)
  const float* g;                  // weight gradient: dL/d(mosaic), ReLU-masked, same shape
  float* part;                     // weight gradient: [waves][6][16][64] partial accumulators
  int nb, H, W, th, tw, waves_per_tile;
};

// tile -> (view, kind, tile row, tile column): mosaic  BL FL / B F / BR FR  (components.py:9-13,70-73)
__device__ __forceinline__ void s6_tile(int tile, int& view, int& kind, int& tr, int& tc) {
  constexpr int V[6] = {3, 0, 4, 1, 5, 2}, K[6] = {0, 0, 2, 3, 1, 1};
  view = V[tile];
  kind = K[tile];
  tr = tile >> 1;
  tc = tile & 1;
}

template <int KIND>
struct S6Kind {
  static constexpr int T = KIND < 2 ? 50 : 52;
  static constexpr int SX = KIND == 0 ? 2 : KIND == 1 ? -2 : KIND == 2 ? -3 : 3;
  static constexpr int ST = (KIND == 0 || KIND == 3) ? 1 : -1;
  __device__ static int x0(int W) { return KIND == 0 ? 0 : KIND == 1 ? W - 1 : KIND == 2 ? W : -1; }
  __device__ static int ys(int L, int H) { return KIND == 0 ? 3 * L : KIND == 1 ? H - 1 - 3 * L : KIND == 2 ? 2 * L - 1 : H - 2 * L; }
};

// One raw row (3 channels) in flight in registers: x = lane + 64 i.
template <bool U8>
struct S6Row {
  float v[3][5];
  __device__ __forceinline__ void fetch(const S6Args& a, int b, int view, int ys, int lane) {
    const bool rowok = ys >= 0 && ys < a.H;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int x = lane + 64 * i;
      const bool ok = rowok && x < a.W;
      if (U8) {
        const unsigned char* src = (const unsigned char*)a.samples.p[b] + (((long)view * a.H + (rowok ? ys : 0)) * a.W + (ok ? x : 0)) * 3;
        // a true division, correctly rounded: bit for bit ToTensor's img.float().div(255) (layout_pool.hip)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][i] = ok ? (float)src[c] / 255.0f : 0.f;
      } else {
        const float* src = (const float*)a.samples.p[b] + ((long)view * 3 * a.H + (rowok ? ys : 0)) * a.W + (ok ? x : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][i] = ok ? src[(long)c * a.H * a.W] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void to_lds(float* strip, int W, int lane) const {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int x = lane + 64 * i;
      if (x < W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) strip[c * S6_PITCH + S6_LPAD + x] = v[c][i];
      }
    }
  }
};

__device__ __forceinline__ void s6_zero_strip(float* strip, int lane) {
  for (int i = lane; i < S6_WAVE; i += 64) strip[i] = 0.f;
}

// ---------------------------------------------------------------------------------------------- forward
template <int KIND, bool U8>
__device__ __forceinline__ void s6_fwd_tile(const S6Args& a, int tile, int view, int tr, int tc, int wt, float* strip) {
  using KD = S6Kind<KIND>;
  constexpr int T = KD::T, NP = T / 2;
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int nl = KIND < 2 ? a.th : a.tw, np = KIND < 2 ? a.tw : a.th;
  const int ntm = (np + 31) / 32;
  const long total = (long)a.nb * nl;
  // the layer's weights for the wave's whole life: k-pair (j, c) = taps 2j, 2j + 1 of channel c on the two half-waves
  float wreg[NP * 3];
  {
    const float* wsrc = a.w[tile] + (long)m * 3 * T + h;      // lane = output channel m
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) wreg[j * 3 + c] = wsrc[c * T + 2 * j];
  }
  const float bias = a.bias[tile][m];
  const float* base = strip + S6_LPAD + KD::x0(a.W) + KD::SX * m + KD::ST * h;
  S6Row<U8> row;
  long line = wt;
  if (line < total) row.fetch(a, (int)(line / nl), view, KD::ys((int)(line % nl), a.H), lane);
  for (; line < total; line += a.waves_per_tile) {
    const int b = (int)(line / nl), L = (int)(line % nl);
    __builtin_amdgcn_wave_barrier();                          // the previous line's reads are issued before this line's writes (same wave: in order)
    row.to_lds(strip, a.W, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long nxt = line + a.waves_per_tile;
    if (nxt < total) row.fetch(a, (int)(nxt / nl), view, KD::ys((int)(nxt % nl), a.H), lane);      // in flight under this line's MFMAs
    for (int mt = 0; mt < ntm; ++mt) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* p = base + KD::SX * 32 * mt;
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc = DD_MFMA(p[c * S6_PITCH + KD::ST * 2 * j], wreg[j * 3 + c], acc);
      unsigned word = 0;                                      // sign word of position 32 mt + (lane & 31), kept by lanes 0 .. 31
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pm = 32 * mt + dd_acc_row(r, lane);
        const float v = acc[r] + bias;
        if (pm < np) {
          const int oy = KIND < 2 ? L : pm, ox = KIND < 2 ? pm : L;
          a.mosaic[((((long)b * 3 + tr) * a.th + oy) * (2 * a.tw) + tc * a.tw + ox) * 32 + m] = v > 0.f ? v : 0.f;
        }
        if (a.bits) {      // lanes 0-31 hold the 32 channels of position i0 = (r & 3) + 8 (r >> 2), lanes 32-63 those of i0 + 4: one ballot = two words
          const unsigned long long bal = __ballot(v > 0.f);
          const int i0 = (r & 3) + 8 * (r >> 2);
          if (lane == i0) word = (unsigned)bal;
          if (lane == i0 + 4) word = (unsigned)(bal >> 32);
        }
      }
      if (a.bits) {
        const int pm = 32 * mt + lane;
        if (lane < 32 && pm < np) {
          const int oy = KIND < 2 ? L : pm, ox = KIND < 2 ? pm : L;
          a.bits[(((long)b * 3 + tr) * a.th + oy) * (2 * a.tw) + tc * a.tw + ox] = word;
        }
      }
    }
  }
}

template <bool U8>
__global__ __launch_bounds__(256) void strip6_fwd_kernel(const S6Args a) {
  __shared__ float lds[4 * S6_WAVE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* strip = lds + wave * S6_WAVE;
  s6_zero_strip(strip, threadIdx.x & 63);
  const int gw = blockIdx.x * 4 + wave;
  const int tile = gw % 6, wt = gw / 6;
  if (wt >= a.waves_per_tile) return;
  int view, kind, tr, tc;
  s6_tile(tile, view, kind, tr, tc);
  switch (kind) {
    case 0: s6_fwd_tile<0, U8>(a, tile, view, tr, tc, wt, strip); break;
    case 1: s6_fwd_tile<1, U8>(a, tile, view, tr, tc, wt, strip); break;
    case 2: s6_fwd_tile<2, U8>(a, tile, view, tr, tc, wt, strip); break;
    default: s6_fwd_tile<3, U8>(a, tile, view, tr, tc, wt, strip); break;
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// accumulator nt = 2 c + half holds dW[co][c][32 half + t'] (row = co from A = dL/dy, column = t' from B = the row); column T of channel 0
// (half 1, t' = T - 32) is fed 1.0 and so holds the bias gradient.
template <int KIND, bool U8>
__device__ __forceinline__ void s6_wgrad_tile(const S6Args& a, int tile, int view, int tr, int tc, int wt, int gwave, float* strip) {
  using KD = S6Kind<KIND>;
  constexpr int T = KD::T;
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int nl = KIND < 2 ? a.th : a.tw, np = KIND < 2 ? a.tw : a.th;
  const int npair = (np + 1) / 2;
  const long total = (long)a.nb * nl;
  f32x16 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const float* base = strip + S6_LPAD + KD::x0(a.W) + KD::SX * h + KD::ST * n;      // position 2 i + h, tap n (+ 32 for the second half)
  const bool bias_lane = (n == T - 32);
  S6Row<U8> row;
  long line = wt;
  if (line < total) row.fetch(a, (int)(line / nl), view, KD::ys((int)(line % nl), a.H), lane);
  for (; line < total; line += a.waves_per_tile) {
    const int b = (int)(line / nl), L = (int)(line % nl);
    __builtin_amdgcn_wave_barrier();
    row.to_lds(strip, a.W, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long nxt = line + a.waves_per_tile;
    if (nxt < total) row.fetch(a, (int)(nxt / nl), view, KD::ys((int)(nxt % nl), a.H), lane);
    // dL/dy of the line: pixel (oy, ox) = (L, pm) for kinds 0 / 1, (pm, L) for kinds 2 / 3; lane = output channel n, position 2 i + h
    const long pix0 = KIND < 2 ? ((((long)b * 3 + tr) * a.th + L) * (2 * a.tw) + tc * a.tw) : ((((long)b * 3 + tr) * a.th) * (2 * a.tw) + tc * a.tw + L);
    const long pstride = KIND < 2 ? 1 : 2 * a.tw;
    const float* gp = a.g + pix0 * 32 + n;
    constexpr int CH = 8;                                     // k-pairs per chunk: the next chunk's dL/dy is in flight under this chunk's 48 MFMAs
    float ga[CH], gb[CH];
    auto load = [&](float (&dst)[CH], int i0) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int pm = 2 * (i0 + q) + h;
        dst[q] = pm < np ? gp[(long)pm * pstride * 32] : 0.f;
      }
    };
    auto mul = [&](const float (&src)[CH], int i0) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        if (i0 + q >= npair) break;                           // wave-uniform: no multiply by strip cells past the positions the strip is padded for
        const float* p = base + KD::SX * 2 * (i0 + q);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float b0 = p[c * S6_PITCH], b1r = p[c * S6_PITCH + KD::ST * 32];
          const float b1 = (c == 0 && bias_lane) ? 1.0f : b1r;
          acc[2 * c] = DD_MFMA(src[q], b0, acc[2 * c]);
          acc[2 * c + 1] = DD_MFMA(src[q], b1, acc[2 * c + 1]);
        }
      }
    };
    load(ga, 0);
    for (int i0 = 0; i0 < npair; i0 += 2 * CH) {
      load(gb, i0 + CH);                                      // positions past np: zeros (and their B operands read the strip's padding)
      mul(ga, i0);
      load(ga, i0 + 2 * CH);
      mul(gb, i0 + CH);
    }
  }
  float* out = a.part + (long)gwave * 6 * 1024;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) out[t * 1024 + i * 64 + lane] = acc[t][i];
}

template <bool U8>
__global__ __launch_bounds__(256) void strip6_wgrad_kernel(const S6Args a) {
  __shared__ float lds[4 * S6_WAVE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* strip = lds + wave * S6_WAVE;
  s6_zero_strip(strip, threadIdx.x & 63);
  const int gw = blockIdx.x * 4 + wave;
  const int tile = gw % 6, wt = gw / 6;
  if (wt >= a.waves_per_tile) return;
  int view, kind, tr, tc;
  s6_tile(tile, view, kind, tr, tc);
  switch (kind) {
    case 0: s6_wgrad_tile<0, U8>(a, tile, view, tr, tc, wt, gw, strip); break;
    case 1: s6_wgrad_tile<1, U8>(a, tile, view, tr, tc, wt, gw, strip); break;
    case 2: s6_wgrad_tile<2, U8>(a, tile, view, tr, tc, wt, gw, strip); break;
    default: s6_wgrad_tile<3, U8>(a, tile, view, tr, tc, wt, gw, strip); break;
  }
}

struct S6Out {
  float* dw[6];
  float* db[6];
};

// second stage: a workgroup owns 64 consecutive accumulator elements (tile, nt, r, lane): a whole register row, 256 contiguous bytes of
// every partial; its 16 thread groups each add the waves w = g, g + 16, ... of the tile (sixteen loads in flight), LDS adds the groups in
// order: fixed order, deterministic.  (One thread per element walking all 341 waves alone was a latency chain: 82 us; eight elements per
// workgroup read 32-byte pieces: 33 us.)
__global__ __launch_bounds__(1024) void strip6_wgrad_reduce(const float* __restrict__ part, const S6Out out, int waves_per_tile, int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;                       // [tile][nt][r][lane], 6 x 6144 elements
  const int tile = e / 6144, rem = e % 6144, nt = rem / 1024, r = (rem % 1024) / 64;
  float s = 0.f;
  if (g < waves_per_tile)
    dd_sum_strided(s, part + ((long)(g * 6 + tile) * 6 + nt) * 1024 + r * 64 + lane, 16L * 36 * 1024, (waves_per_tile - g + 15) / 16);
  red[g][lane] = s;
  __syncthreads();
  if (g != 0) return;
  s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i][lane];
  const int T = (tile == 2 || tile == 3) ? 52 : 50;
  const int co = dd_acc_row(r, lane), c = nt >> 1, t = 32 * (nt & 1) + (lane & 31);
  if (t < T) {
    float* d = out.dw[tile] + ((long)co * 3 + c) * T + t;
    *d = accumulate ? *d + s : s;
  } else if (t == T && c == 0) {
    float* d = out.db[tile] + co;
    *d = accumulate ? *d + s : s;
  }
}

// Host-side validation: every LDS index the kernels can form stays inside a wave's strip.
bool s6_shape_ok(int H, int W, int& th, int& tw) {
  if (H < 52 || W < 52 || W > S6_MAXW) return false;
  th = (H - 1) / 3 + 1;                                       // kinds 0 / 1: rows; must equal kinds 2 / 3: (W + 2 - 52) / 3 + 1
  tw = (W - 50) / 2 + 1;                                      // kinds 0 / 1: columns; must equal kinds 2 / 3: (H + 2 - 1) / 2 + 1
  if (th != (W + 2 - 52) / 3 + 1 || tw != (H + 1) / 2 + 1) return false;      // the six tiles must be equal (components.py:70-73 cat)
  if (tw > 160 || th > 96) return false;
  // forward: positions up to the end of the last 32-wide tile, taps < T; weight gradient: positions up to the odd member of the last pair,
  // taps < 64 (the second tap half is read whole)
  const int lo = -S6_LPAD, hi = S6_PITCH - S6_LPAD - 1;
  auto ok = [&](int v) { return v >= lo && v <= hi; };
  const int pf0 = 32 * ((tw + 31) / 32) - 1, pf2 = 32 * ((th + 31) / 32) - 1, pw0 = 2 * ((tw + 1) / 2) - 1, pw2 = 2 * ((th + 1) / 2) - 1;
  bool good = ok(W) && ok(-1);
  for (int pass = 0; pass < 2; ++pass) {
    const int p0 = pass ? pw0 : pf0, p2 = pass ? pw2 : pf2, t0 = pass ? 63 : 49, t2 = pass ? 63 : 51;
    good = good && ok(2 * p0 + t0) && ok(W - 1 - 2 * p0 - t0) && ok(W - 3 * p2 - t2) && ok(-1 + 3 * p2 + t2);
  }
  return good;
}

int s6_fill(S6Args& a, const void* const* sample_ptrs, int b0, int nb, const float* const* weights, const float* const* biases, int H, int W,
            int th, int tw, int waves_per_tile) {
  for (int i = 0; i < 64; ++i) a.samples.p[i] = i < nb ? sample_ptrs[b0 + i] : nullptr;
  for (int i = 0; i < nb; ++i) DD_REQUIRE(a.samples.p[i] != nullptr, DD_ERR_BAD_ARG, "strip6: null sample pointer");
  for (int t = 0; t < 6; ++t) {
    a.w[t] = weights ? weights[t] : nullptr;
    a.bias[t] = biases ? biases[t] : nullptr;
  }
  a.nb = nb; a.H = H; a.W = W; a.th = th; a.tw = tw; a.waves_per_tile = waves_per_tile;
  return 0;
}

constexpr int S6_FWD_BLOCKS = DD_NUM_CU * 2;       // 2048 waves, 341 per tile: two workgroups per CU are resident (210 registers per wave)
constexpr int S6_WG_BLOCKS = DD_NUM_CU * 2;        // 2048 waves, 341 per tile

}  // namespace

extern "C" {

int32_t dd_strip6_supported(int32_t height, int32_t width) {
  int th, tw;
  return s6_shape_ok(height, width, th, tw) ? 1 : 0;
}

int dd_strip6_fwd(const void* const* sample_ptrs, int32_t u8, const float* const* weights, const float* const* biases, float* mosaic,
                  uint32_t* relu_bits, int32_t batch, int32_t height, int32_t width, void* stream) {
  DD_REQUIRE(sample_ptrs && weights && biases && mosaic && batch > 0, DD_ERR_BAD_ARG, "strip6_fwd: bad argument");
  int th, tw;
  DD_REQUIRE(s6_shape_ok(height, width, th, tw), DD_ERR_UNSUPPORTED, "strip6_fwd: %d x %d views are not served (six equal tiles, W <= %d)", height, width, S6_MAXW);
  for (int t = 0; t < 6; ++t) DD_REQUIRE(weights[t] && biases[t], DD_ERR_BAD_ARG, "strip6_fwd: null weight / bias pointer");
  const int wpt = S6_FWD_BLOCKS * 4 / 6;
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    S6Args a;
    if (int rc = s6_fill(a, sample_ptrs, b0, nb, weights, biases, height, width, th, tw, wpt)) return rc;
    a.mosaic = mosaic + (long)b0 * 3 * th * 2 * tw * 32;
    a.bits = relu_bits ? relu_bits + (long)b0 * 3 * th * 2 * tw : nullptr;
    a.g = nullptr; a.part = nullptr;
    if (u8) hipLaunchKernelGGL(strip6_fwd_kernel<true>, dim3(S6_FWD_BLOCKS), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(strip6_fwd_kernel<false>, dim3(S6_FWD_BLOCKS), dim3(256), 0, (hipStream_t)stream, a);
    DD_LAUNCH_CHECK("strip6_fwd");
  }
  return 0;
}

int64_t dd_strip6_wgrad_workspace_bytes(void) { return (int64_t)S6_WG_BLOCKS * 4 * 6 * 1024 * sizeof(float); }

int dd_strip6_wgrad(const void* const* sample_ptrs, int32_t u8, const float* g, float* const* dweights, float* const* dbiases, int32_t batch,
                    int32_t height, int32_t width, void* workspace, int64_t workspace_bytes, void* stream) {
  DD_REQUIRE(sample_ptrs && g && dweights && dbiases && workspace && batch > 0, DD_ERR_BAD_ARG, "strip6_wgrad: bad argument");
  DD_REQUIRE(workspace_bytes >= dd_strip6_wgrad_workspace_bytes(), DD_ERR_WORKSPACE, "strip6_wgrad: workspace too small");
  int th, tw;
  DD_REQUIRE(s6_shape_ok(height, width, th, tw), DD_ERR_UNSUPPORTED, "strip6_wgrad: %d x %d views are not served", height, width);
  S6Out out;
  for (int t = 0; t < 6; ++t) {
    DD_REQUIRE(dweights[t] && dbiases[t], DD_ERR_BAD_ARG, "strip6_wgrad: null gradient pointer");
    out.dw[t] = dweights[t];
    out.db[t] = dbiases[t];
  }
  const int wpt = S6_WG_BLOCKS * 4 / 6;
  for (int b0 = 0; b0 < batch; b0 += 64) {
    const int nb = min(64, batch - b0);
    S6Args a;
    if (int rc = s6_fill(a, sample_ptrs, b0, nb, nullptr, nullptr, height, width, th, tw, wpt)) return rc;
    a.mosaic = nullptr;
    a.bits = nullptr;
    a.g = g + (long)b0 * 3 * th * 2 * tw * 32;
    a.part = (float*)workspace;
    if (u8) hipLaunchKernelGGL(strip6_wgrad_kernel<true>, dim3(S6_WG_BLOCKS), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(strip6_wgrad_kernel<false>, dim3(S6_WG_BLOCKS), dim3(256), 0, (hipStream_t)stream, a);
    DD_LAUNCH_CHECK("strip6_wgrad");
    hipLaunchKernelGGL(strip6_wgrad_reduce, dim3(6 * 6 * 1024 / 64), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, out, wpt,
                       b0 > 0 ? 1 : 0);
    DD_LAUNCH_CHECK("strip6_wgrad_reduce");
  }
  return 0;
}

}  // extern "C"
