// Issue model of fp32 MFMA next to other instructions on gfx950, one or two waves per SIMD (tools only: not part of the library).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2p __attribute__((ext_vector_type(2)));

template <int NV, int NDS, int NS, bool PK>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  f32x4v acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4v{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x2p t[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  float4 dsv = {0, 0, 0, 0};
  int sacc = 0;
  lds[threadIdx.x] = a;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t[v & 3]) : "v"(t[(v + 1) & 3]));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[v & 3].x) : "v"(t[(v + 1) & 3].y));
      }
#pragma unroll
      for (int d = 0; d < NDS; ++d) asm volatile("ds_read_b128 %0, %1" : "=v"(dsv) : "v"((int)(threadIdx.x & 63) * 16));
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
    }
    if (NDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + t[0].x + t[1].x + t[2].y + t[3].y + dsv.x + sacc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// G MFMAs back to back, then G * NV vector instructions: how much does each switch between the two kinds cost?
template <int G, int NV, bool PK>
__global__ __launch_bounds__(512) void kg(float* out, long long* cyc, int iters) {
  f32x4v acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4v{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x2p t[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m0 = 0; m0 < 32; m0 += G) {
#pragma unroll
      for (int m = m0; m < m0 + G; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 15]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < G * NV; ++v) {
        if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t[v & 3]) : "v"(t[(v + 1) & 3]));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[v & 3].x) : "v"(t[(v + 1) & 3].y));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + t[0].x + t[1].x + t[2].y + t[3].y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int G, int NV, bool PK>
void rung(int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 1000;
  kg<G, NV, PK><<<256, threads>>>(out, cyc, 10);
  kg<G, NV, PK><<<256, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("groups of %2d MFMAs + %d x %s each: waves/SIMD %d  %.1f cycles per MFMA (32 = matrix pipe alone)\n", G, NV, PK ? "v_pk_add_f32" : "v_add_f32", threads / 256,
         (double)c / (iters * 32.0));
  hipFree(out); hipFree(cyc);
}

// Two waves on every SIMD: waves 0-3 issue MFMAs only, waves 4-7 vector adds only (MODE 1), MFMAs too (MODE 2) or exit (MODE 0).
// Does another wave's vector work take matrix time?
template <int MODE, bool PK>
__global__ __launch_bounds__(512) void kmix(float* out, long long* cyc, int iters) {
  f32x4v acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4v{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x2p t[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4 || MODE == 2;
  if (!mf && MODE == 0) return;
  long long t0 = __builtin_readcyclecounter();
  if (mf) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 15]) : "v"(a), "v"(b));
  } else {
    for (int it = 0; it < iters * 8; ++it)
#pragma unroll
      for (int v = 0; v < 32; ++v) {
        if (PK) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t[v & 3]) : "v"(t[(v + 1) & 3]));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[v & 3].x) : "v"(t[(v + 1) & 3].y));
      }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + t[0].x + t[1].x + t[2].y + t[3].y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

template <int MODE, bool PK>
void runmix(const char* what) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  const int iters = 1000;
  kmix<MODE, PK><<<256, 512>>>(out, cyc, 10);
  kmix<MODE, PK><<<256, 512>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
  printf("two waves per SIMD, wave A MFMAs only, wave B %s: %.1f cycles per MFMA of wave A", what, (double)c[0] / (iters * 32.0));
  if (MODE == 1) printf("; wave B: %.1f cycles per vector instruction", (double)c[1] / (iters * 8 * 32.0));
  printf("\n");
  hipFree(out); hipFree(cyc);
}

// Every wave runs [NVB packed adds][32 MFMAs] per stage, like a stage of the c2 kernels: with two waves per SIMD, does one wave's
// vector block fall under the other's MFMA block by itself?
template <int NVB>
__global__ __launch_bounds__(512) void kstage(float* out, long long* cyc, int iters) {
  f32x4v acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4v{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x2p t[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int v = 0; v < NVB; ++v) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t[v & 3]) : "v"(t[(v + 1) & 3]));
#pragma unroll
    for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 15]) : "v"(a), "v"(b));
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + t[0].x + t[1].x + t[2].y + t[3].y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

// The same with the alternation FORCED: waves 0-3 run [adds] barrier [MFMAs] barrier, waves 4-7 [MFMAs] barrier [adds] barrier.
template <int NVB>
__global__ __launch_bounds__(512) void kphase(float* out, long long* cyc, int iters) {
  f32x4v acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4v{0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f32x2p t[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
  const bool second = (threadIdx.x >> 8) != 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      if ((ph == 0) != second) {
#pragma unroll
        for (int v = 0; v < NVB; ++v) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t[v & 3]) : "v"(t[(v + 1) & 3]));
      } else {
#pragma unroll
        for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 15]) : "v"(a), "v"(b));
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + t[0].x + t[1].x + t[2].y + t[3].y;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NVB>
void runphase() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  const int iters = 2000;
  kphase<NVB><<<256, 512>>>(out, cyc, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kphase<NVB><<<256, 512>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("stage = %3d packed adds + 32 MFMAs, 2 waves per SIMD in FORCED alternation (two barriers a stage): %.3f ms wall = %.0f ns per stage and SIMD-wave\n", NVB, ms,
         ms * 1e6 / iters / 2);
  hipFree(out); hipFree(cyc);
}

template <int NVB>
void runstage() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
  const int iters = 2000;
  for (int th = 256; th <= 512; th += 256) {
    kstage<NVB><<<256, th>>>(out, cyc, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kstage<NVB><<<256, th>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("stage = %3d packed adds + 32 MFMAs, %d wave(s) per SIMD: %.0f cycles per stage of wave 0 (counter), %.3f ms wall = %.0f ns per stage and SIMD-wave\n", NVB,
           th / 256, (double)c[0] / iters, ms, ms * 1e6 / iters / (th / 256));
  }
  hipFree(out); hipFree(cyc);
}

template <int NV, int NDS, int NS, bool PK>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, NDS, NS, PK><<<256, threads>>>(out, cyc, 10);
  hipEventRecord(e0);
  k<NV, NDS, NS, PK><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // readcyclecounter = s_memtime (shader clock? on gfx950 a constant 100 MHz counter is s_memrealtime); report both
  printf("%-34s waves/SIMD %d  %.3f ms  -> %.1f ns per MFMA slot (%.1f cycles at 2.4 GHz), counter %.1f / MFMA\n", name, threads / 256, ms,
         ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) * 2.4, (double)c / (iters * 16.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int th = 256; th <= 512; th += 256) {
    run<0, 0, 0, false>("mfma only", th);
    run<1, 0, 0, false>("mfma + 1 v_add_f32", th);
    run<2, 0, 0, false>("mfma + 2 v_add_f32", th);
    run<4, 0, 0, false>("mfma + 4 v_add_f32", th);
    run<8, 0, 0, false>("mfma + 8 v_add_f32", th);
    run<2, 0, 0, true>("mfma + 2 v_pk_add_f32", th);
    run<4, 0, 0, true>("mfma + 4 v_pk_add_f32", th);
    run<8, 0, 0, true>("mfma + 8 v_pk_add_f32", th);
    run<0, 1, 0, false>("mfma + 1 ds_read_b128", th);
    run<0, 0, 4, false>("mfma + 4 s_add", th);
    run<0, 0, 16, false>("mfma + 16 s_add", th);
    run<2, 1, 4, true>("mfma + 2 pk + 1 ds + 4 salu", th);
  }
  for (int th = 256; th <= 512; th += 256) {
    rung<1, 2, true>(th); rung<2, 2, true>(th); rung<4, 2, true>(th); rung<8, 2, true>(th); rung<16, 2, true>(th); rung<32, 2, true>(th);
    rung<1, 2, false>(th); rung<4, 2, false>(th); rung<32, 2, false>(th);
  }
  runstage<50>(); runstage<100>(); runstage<150>();
  runphase<50>(); runphase<100>(); runphase<150>();
  runmix<0, false>("absent");
  runmix<1, false>("v_add_f32 only");
  runmix<1, true>("v_pk_add_f32 only");
  runmix<2, false>("MFMAs too");
  return 0;
}
