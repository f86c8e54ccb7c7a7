// Does the range check of a raw buffer access include the scalar offset on gfx950?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* out, int rowbytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, rowbytes, 0x00020000);
  // row 3, element threadIdx.x: voffset inside [0, rowbytes), soffset = 3 rows
  float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, 3 * rowbytes, 0));
  // voffset out of range, soffset 0
  float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, rowbytes + threadIdx.x * 4, 0, 0));
  out[threadIdx.x] = a;
  out[64 + threadIdx.x] = b;
}
int main() {
  const int W = 64, H = 8;
  float h[W * H];
  for (int i = 0; i < W * H; ++i) h[i] = (float)i;
  float *x, *o;
  (void)hipMalloc(&x, sizeof(h)); (void)hipMalloc(&o, 128 * 4);
  (void)hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, o, W * 4);
  float r[128];
  (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("soffset = 3 rows, voffset in range : lane 5 reads %g (row 3 element 5 = %g; 0 = the scalar offset is range-checked)\n", r[5], h[3 * W + 5]);
  printf("voffset one row out of range       : lane 5 reads %g (0 expected)\n", r[64 + 5]);
  return 0;
}
