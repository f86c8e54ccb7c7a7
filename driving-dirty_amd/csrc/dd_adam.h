// The one definition of the Adam element update shared by every optimizer kernel (dense.hip: dd_adam_step / dd_adam_step_multi;
// adam_rankb.hip: dd_adam_step_rankb): the same inputs give the same bits whichever kernel a tensor meets.
#pragma once
#include "dd_common.h"

// One element of torch.optim.Adam, written with explicit fused multiply-adds and contraction off: the vector kernel, its scalar
// tail and the multi-tensor kernel must produce the SAME bits for the same inputs (left to the compiler, `a * b + c` was
// contracted differently in the float4 loop and in the scalar kernels, so a tensor updated by one kernel or the other -- the
// overlapped optimizer picks by size and timing -- ended a last bit apart).
// Hardware square root and reciprocal (v_sqrt_f32 / v_rcp_f32, 1 ulp each) instead of the IEEE sequences, and packed fp32 for
// the rest: 34 vector instructions per four elements instead of 175.  The pass usually runs beside the conv backward, where its
// vector instructions compete with the conv kernels' own (tools/ubench/mfma_issue.hip: they do overlap the MFMAs of another
// wave but not its vector work): with the IEEE form the register-row data gradient took 1.95-1.97 ms in the step, with this one
// 1.70-1.78.  m and v are exact (multiplies and fmas only); p differs from torch.optim.Adam's by at most the
// rounding of an update term that is 2e-7 relative off (tests: 1e-6).
typedef float f32x2a __attribute__((ext_vector_type(2)));
// omb1 = 1.f - b1, omb2 = 1.f - b2 (fp32 subtractions: the same bits on the host and on the device).  A kernel that passes them as
// arguments keeps them in scalar registers; computed in the kernel they occupy vector registers (no scalar float ALU on gfx950).
__device__ __forceinline__ void adam_elem2(f32x2a& p, f32x2a& m, f32x2a& v, f32x2a g, float gscale, float b1, float b2, float omb1,
                                           float omb2, float eps, float step_size, float inv_bc2) {
#pragma clang fp contract(off)
  const f32x2a gg = g * gscale;
  const f32x2a mm = __builtin_elementwise_fma(f32x2a{b1, b1}, m, gg * omb1);
  const f32x2a vv = __builtin_elementwise_fma(f32x2a{b2, b2}, v, (gg * omb2) * gg);
  m = mm;
  v = vv;
  const f32x2a s = {__builtin_amdgcn_sqrtf(vv.x), __builtin_amdgcn_sqrtf(vv.y)};
  const f32x2a denom = __builtin_elementwise_fma(s, f32x2a{inv_bc2, inv_bc2}, f32x2a{eps, eps});
  const f32x2a r = {__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
  p = __builtin_elementwise_fma(f32x2a{-step_size, -step_size}, mm * r, p);
}
__device__ __forceinline__ void adam_elem2(f32x2a& p, f32x2a& m, f32x2a& v, f32x2a g, float gscale, float b1, float b2, float eps,
                                           float step_size, float inv_bc2) {
  adam_elem2(p, m, v, g, gscale, b1, b2, 1.f - b1, 1.f - b2, eps, step_size, inv_bc2);
}
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float gscale, float b1, float b2, float eps,
                                          float step_size, float bc2_sqrt) {      // the same operations on one element (tails, small tensors)
  f32x2a pp = {p, 0.f}, mm = {m, 0.f}, vv = {v, 0.f};
  adam_elem2(pp, mm, vv, f32x2a{g, 0.f}, gscale, b1, b2, eps, step_size, 1.f / bc2_sqrt);
  p = pp.x; m = mm.x; v = vv.x;
}

