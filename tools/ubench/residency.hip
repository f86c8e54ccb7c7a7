// Who becomes resident beside whom?  (tools only)  A "conv-like" persistent kernel A (256 workgroups x 4 waves, one wave per SIMD,
// R registers a lane, L bytes of LDS) spins for ~1 ms; a streaming kernel B (1024 blocks x 256 threads, <= 48 registers, no LDS) copies
// 1 GB.  B is launched on a second stream 100 us after A (or A after B): how long do both take?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/residency.hip -o /tmp/residency && /tmp/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAD>
__global__ __launch_bounds__(256) void spin_kernel(long long cycles, float* out) {
  extern __shared__ char lds[];
  if (PAD == 1) asm volatile("; pad" ::: "v199", "a255");      // 200 + 256 = 456
  if (PAD == 2) asm volatile("; pad" ::: "v207", "a255");      // 208 + 256 = 464
  if (PAD == 3) asm volatile("; pad" ::: "v231", "a255");      // 232 + 256 = 488
  if (PAD == 4) asm volatile("; pad" ::: "v255", "a183");      // 256 + 184 = 440
  if (PAD == 5) asm volatile("; pad" ::: "v255", "a227");      // 256 + 228 = 484
  const long long t0 = __builtin_readcyclecounter();
  float s = 0.f;
  while (__builtin_readcyclecounter() - t0 < cycles) s += 1.f;
  if (s < 0.f) out[threadIdx.x] = s + lds[threadIdx.x];
}

__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ b, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) b[i] = a[i] * 1.0001f;
}

template <int PAD>
void run(const char* what, size_t ldsbytes, bool b_first) {
  const long n4 = (1L << 30) / 16;
  f32x4 *a, *b; float* o;
  (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16); (void)hipMalloc(&o, 4096);
  (void)hipMemset(a, 0, n4 * 16);
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  hipEvent_t a0, a1, b0, b1; (void)hipEventCreate(&a0); (void)hipEventCreate(&a1); (void)hipEventCreate(&b0); (void)hipEventCreate(&b1);
  (void)hipFuncSetAttribute((const void*)spin_kernel<PAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsbytes);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipDeviceSynchronize();
    if (!b_first) {
      (void)hipEventRecord(a0, s1); spin_kernel<PAD><<<256, 256, ldsbytes, s1>>>(2400000, o); (void)hipEventRecord(a1, s1);
      spin_kernel<0><<<1, 64, 0, s2>>>(240000, o);      // ~100 us later
      (void)hipEventRecord(b0, s2); stream_kernel<<<1024, 256, 0, s2>>>(a, b, n4); (void)hipEventRecord(b1, s2);
    } else {
      (void)hipEventRecord(b0, s2); stream_kernel<<<1024, 256, 0, s2>>>(a, b, n4); (void)hipEventRecord(b1, s2);
      spin_kernel<0><<<1, 64, 0, s1>>>(24000, o);       // ~10 us later
      (void)hipEventRecord(a0, s1); spin_kernel<PAD><<<256, 256, ldsbytes, s1>>>(2400000, o); (void)hipEventRecord(a1, s1);
    }
    (void)hipDeviceSynchronize();
  }
  float ta, tb, tab;
  (void)hipEventElapsedTime(&ta, a0, a1); (void)hipEventElapsedTime(&tb, b0, b1); (void)hipEventElapsedTime(&tab, b_first ? b0 : a0, b_first ? a1 : b1);
  printf("%-34s LDS %3zu KB, %s first: A %.3f ms (1.0 alone), B %.3f ms, first start -> last end %.3f ms\n", what, ldsbytes >> 10, b_first ? "B" : "A", ta, tb, tab);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(o);
}

int main() {
  {      // B alone
    const long n4 = (1L << 30) / 16; f32x4 *a, *b; (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    stream_kernel<<<1024, 256>>>(a, b, n4); (void)hipEventRecord(e0); stream_kernel<<<1024, 256>>>(a, b, n4); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float t; (void)hipEventElapsedTime(&t, e0, e1); printf("B alone: %.3f ms (2 GB of traffic)\n", t); (void)hipFree(a); (void)hipFree(b);
  }
  for (int bf = 0; bf < 2; ++bf) {
    run<4>("A = 440 registers", 130 << 10, bf);
    run<1>("A = 456 registers", 148 << 10, bf);
    run<2>("A = 464 registers", 75 << 10, bf);
    run<5>("A = 484 registers", 130 << 10, bf);
    run<3>("A = 488 registers", 148 << 10, bf);
  }
  return 0;
}
