// Shared helpers for the gfx950 kernels of the hot path (internal; the public ABI is include/dd_hotpath.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dd_hotpath.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D(32x32) += A(32x2) * B(2x32), exact fp32 (v_mfma_f32_32x32x2_f32, 64 cycles / SIMD).
// lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31];
// D register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
#define DD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define DD_NUM_CU 256

int dd_fail(int code, const char* fmt, ...);
int dd_adam_blocks_internal();  // persistent workgroups per CU of dd_adam_step / dd_adam_step_rankb: dd_set_adam_blocks_per_cu
int dd_adam_spare_internal();   // compute units dd_adam_step_rankb leaves free of its workgroups: dd_set_adam_spare_cus
int dd_cu_budget_internal();   // compute units the resident-grid (persistent) kernels may fill: dd_set_cu_budget

#define DD_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) return dd_fail(code, __VA_ARGS__); \
  } while (0)

#define DD_LAUNCH_CHECK(what)                                                        \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) return dd_fail(DD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

static inline int dd_conv_out(int in, int stride) { return (in + 2 - 3) / stride + 1; }

__device__ __forceinline__ int dd_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The lane id, recomputed where it is needed: a `volatile` asm is neither hoisted nor shared, so code after a long MFMA loop
// (an epilogue's addresses, the next tile's fill plan) does not keep lane-derived registers alive across that loop -- which is
// what the register allocator otherwise spills to scratch in kernels that use the whole register file.
__device__ __forceinline__ int dd_fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Raw buffer access: the descriptor (wave-uniform base + byte count) makes the hardware range-check every lane:
// an out-of-range load returns zeros, an out-of-range store is dropped.  A negative offset is a huge unsigned
// one, i.e. out of range.  Used for zero padding and ragged edges without branches, and as a guard against faults.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dd_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 dd_bload4(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float dd_bload1(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void dd_bstore1(__amdgpu_buffer_rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// s += p[0] + p[stride] + ... (n terms), left to right -- the order, and so the bits, of the plain loop -- with SIXTEEN loads in flight:
// the second stages of the weight / bias gradients add a few hundred per-workgroup partials per output element, and written as
// `for (w...) s += part[w * stride]` each add waited for its own load: 50-60 us of memory latency per launch for a few KB of output
// (round 5: 0.49 ms of the box-head step in ten such launches).
template <typename Acc>
__device__ __forceinline__ void dd_sum_strided(Acc& s, const float* __restrict__ p, long stride, int n) {
  for (int i = 0; i < n; i += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = p[(long)min(i + j, n - 1) * stride];      // unconditional (a guarded load is a branch per load): past the end, the last term again, not added
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (i + j < n) s += (Acc)v[j];
  }
}

// The contiguous range [idx, end) of `total` work items owned by piece `i` of `n` equal pieces.
__device__ __forceinline__ void dd_range(long total, int i, int n, long& idx, long& end) {
  const long per = (total + n - 1) / n;
  idx = (long)i * per;
  end = idx + per < total ? idx + per : total;
  if (idx > end) idx = end;
}

// dconv_t.hip: the input-aligned forward of the dilated transposed layers; false = not one of its layers, nothing launched.
bool dd_dconv_desc_ok(const dd_gconv_desc* d);      // dconv.hip: the eligibility test of dd_dconv_fwd, for its two specialised launchers
bool dd_dconv_tfwd_launch(const float* x, const float* packed, const float* bias, float* y, const dd_gconv_desc* d, int epilogue,
                          int wp_bytes, hipStream_t st);
bool dd_dconv_tfwd8_launch(const float* x, const float* packed, const float* bias, float* y, const dd_gconv_desc* d, int epilogue,
                           hipStream_t st);      // dconv_t.hip: up_conv_4's forward, four tap columns per column tile
bool dd_dconv_gfwd_launch(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                          int epilogue, int wp_bytes, hipStream_t st);
// dconv_m.hip: gather form with several output rows per workgroup (rows that are not a whole number of 8 m-tiles); false = not one of its layers.
bool dd_dconv_mfwd_launch(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                          int epilogue, int wp_bytes, hipStream_t st);
bool dd_dconv_mfwd_launch_colsum(const float* x, const float* packed, const float* bias, const float* mask, float* y, const dd_gconv_desc* d,
                                 int epilogue, int wp_bytes, hipStream_t st, float* colsum_part);
void dd_dconv_colsum_reduce_launch(const float* part, float* out, int nblocks, int cout, hipStream_t st);
bool dd_dconv_mwin_takes(const dd_gconv_desc* d, int epilogue, bool has_mask, bool has_bias);
