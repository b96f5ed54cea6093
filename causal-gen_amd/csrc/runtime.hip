// Error plumbing, version and small utility kernels (Philox fill, rng advance).
#include <stdarg.h>

#include "common.h"

namespace cgen {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CGEN_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
  return CGEN_OK;
}

__global__ __launch_bounds__(256) void philox_fill_kernel(float* out, int64_t count, const uint64_t* rng, uint32_t stream_id) {
  const uint64_t seed = rng[0], off = rng[1];
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g * 4 < count; g += (int64_t)gridDim.x * 256) {
    float z[4];
    Philox::normal4(seed, off, stream_id, (uint64_t)g, z);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (g * 4 + e < count) out[g * 4 + e] = z[e];
  }
}

__global__ void rng_advance_kernel(uint64_t* rng, uint64_t inc) { rng[1] += inc; }
}  // namespace cgen

using namespace cgen;

extern "C" int cgen_version(void) { return CGEN_ABI_VERSION; }  // ABI version (include/cgen_hip.h CGEN_ABI_VERSION): bumped whenever a struct, an enum value or a signature moves
extern "C" int cgen_h16_format(void) {
#ifdef CGEN_H16_BF16
  return 1;
#else
  return 0;
#endif
}
extern "C" const char* cgen_last_error(void) { return g_err; }

extern "C" int cgen_philox_normal(float* out, int64_t count, const uint64_t* rng, uint32_t stream_id, cgen_stream_t stream) {
  CGEN_REQUIRE(out && rng && count >= 0, "cgen_philox_normal: bad args");
  if (count == 0) return CGEN_OK;
  int blocks = ceil_div(ceil_div(count, 4), 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(philox_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, count, rng, stream_id);
  return check_launch("cgen_philox_normal");
}

extern "C" int cgen_rng_advance(uint64_t* rng, uint64_t inc, cgen_stream_t stream) {
  CGEN_REQUIRE(rng, "cgen_rng_advance: null");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng, inc);
  return check_launch("cgen_rng_advance");
}
