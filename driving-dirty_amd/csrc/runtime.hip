// Error reporting of the C ABI (thread-local message; see include/dd_hotpath.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "dd_common.h"

static thread_local char g_err[512] = "";

int dd_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int g_cu_budget = DD_NUM_CU;
static int g_adam_blocks = 1;      // persistent workgroups per CU of the optimizer kernels (dd_set_adam_blocks_per_cu)
static int g_adam_spare = 0;       // compute units dd_adam_step_rankb leaves without a workgroup of its own (dd_set_adam_spare_cus)

int dd_adam_blocks_internal() { return g_adam_blocks; }
int dd_adam_spare_internal() { return g_adam_spare; }
int dd_cu_budget_internal() { return g_cu_budget; }

// One wave that samples the shader-clock counter (s_memtime: counts at the clock the CUs actually run at) against the constant
// 100 MHz reference counter (s_memrealtime) every ~`spin` sleeps: the clock the part delivers WHILE other kernels run, which no
// per-kernel profiler counter shows for kernels that overlap on different streams.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ samples, int nsamples, int spin) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < nsamples; ++i) {
    samples[2 * i] = __builtin_readcyclecounter();
    samples[2 * i + 1] = wall_clock64();
    for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(127);
  }
}

extern "C" {
int dd_abi_version(void) { return DD_ABI_VERSION; }

int dd_clock_probe(uint64_t* samples, int32_t nsamples, int32_t spin, void* stream) {
  DD_REQUIRE(samples && nsamples > 0 && spin >= 0, DD_ERR_BAD_ARG, "clock_probe: bad argument");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)samples, nsamples, spin);
  DD_LAUNCH_CHECK("clock_probe");
  return 0;
}

int dd_set_cu_budget(int32_t compute_units) {
  DD_REQUIRE(compute_units >= 1 && compute_units <= DD_NUM_CU, DD_ERR_BAD_ARG, "set_cu_budget: %d not in 1..%d", compute_units, DD_NUM_CU);
  g_cu_budget = compute_units;
  return 0;
}
int dd_get_cu_budget(void) { return g_cu_budget; }

int dd_set_adam_blocks_per_cu(int32_t blocks) {
  DD_REQUIRE(blocks >= 1 && blocks <= 8, DD_ERR_BAD_ARG, "set_adam_blocks_per_cu: %d not in 1..8", blocks);
  g_adam_blocks = blocks;
  return 0;
}
int dd_set_adam_spare_cus(int32_t compute_units) {
  DD_REQUIRE(compute_units >= 0 && compute_units <= DD_NUM_CU / 2, DD_ERR_BAD_ARG, "set_adam_spare_cus: %d not in 0..%d", compute_units,
             DD_NUM_CU / 2);
  g_adam_spare = compute_units;
  return 0;
}
const char* dd_last_error(void) { return g_err; }
}
