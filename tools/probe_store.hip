// Store-throughput probe (gfx950): plain coalesced stores / copies at several sizes and store widths.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int W>  // bytes per lane per store: 16, 8, 4
__global__ __launch_bounds__(256) void wr(char* dst, size_t n16, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    if (W == 16) ((uint4*)dst)[i] = make_uint4(v, v, v, v);
    if (W == 8) { ((uint2*)dst)[2 * i] = make_uint2(v, v); ((uint2*)dst)[2 * i + 1] = make_uint2(v, v); }
    if (W == 4) { for (int j = 0; j < 4; ++j) ((uint32_t*)dst)[4 * i + j] = v; }
  }
}
__global__ __launch_bounds__(256) void cp(const uint4* src, uint4* dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void rd(const uint4* src, uint32_t* out, size_t n16) {
  uint32_t a = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 t = src[i]; a ^= t.x ^ t.y ^ t.z ^ t.w; }
  if (a == 0x12345) out[0] = a;
}
// one 16-byte chunk per lane, each lane writes ONE chunk then exits (like a conv epilogue): grid covers everything
__global__ __launch_bounds__(256) void wr_once(uint4* dst, size_t n16, uint32_t v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = make_uint4(v, v, v, v);
}
template <typename F> static float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
int main() {
  char *a, *b; uint32_t* o;
  hipMalloc(&a, (size_t)1 << 30); hipMalloc(&b, (size_t)1 << 30); hipMalloc(&o, 64);
  hipMemset(a, 1, (size_t)1 << 30); hipMemset(b, 2, (size_t)1 << 30);
  for (size_t mb : {8, 16, 38, 64, 256, 1024}) {
    size_t bytes = mb << 20, n16 = bytes / 16;
    for (int grid : {256, 1024, 4096}) {
      float t16 = timeit([&] { hipLaunchKernelGGL(wr<16>, dim3(grid), dim3(256), 0, 0, b, n16, 7u); }, 20);
      float t8 = timeit([&] { hipLaunchKernelGGL(wr<8>, dim3(grid), dim3(256), 0, 0, b, n16, 7u); }, 20);
      float t4 = timeit([&] { hipLaunchKernelGGL(wr<4>, dim3(grid), dim3(256), 0, 0, b, n16, 7u); }, 20);
      float tc = timeit([&] { hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); }, 20);
      float tr = timeit([&] { hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, (const uint4*)a, o, n16); }, 20);
      printf("%5zu MB grid %5d: write16 %7.1f us %6.0f GB/s | write8 %7.1f us %6.0f | write4 %7.1f us %6.0f | copy %7.1f us %6.0f GB/s(r+w) | read %7.1f us %6.0f GB/s\n", mb, grid,
             t16 * 1e3, bytes / t16 / 1e6, t8 * 1e3, bytes / t8 / 1e6, t4 * 1e3, bytes / t4 / 1e6, tc * 1e3, 2.0 * bytes / tc / 1e6, tr * 1e3, bytes / tr / 1e6);
    }
    float to = timeit([&] { hipLaunchKernelGGL(wr_once, dim3((n16 + 255) / 256), dim3(256), 0, 0, (uint4*)b, n16, 7u); }, 20);
    printf("%5zu MB write-once (1 chunk per lane, %zu WGs): %7.1f us %6.0f GB/s\n", mb, (n16 + 255) / 256, to * 1e3, bytes / to / 1e6);
  }
  float te = timeit([&] { hipLaunchKernelGGL(wr_once, dim3(1), dim3(256), 0, 0, (uint4*)b, (size_t)0, 7u); }, 200);
  printf("empty kernel back-to-back: %.2f us\n", te * 1e3);
  return 0;
}
