#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint4* src, uint4* dst, int n) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 16, 0x00020000);
  int off = threadIdx.x * 16;
  if (threadIdx.x & 1) off = 0x7fffffff;
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  dst[threadIdx.x] = make_uint4(v.x, v.y, v.z, v.w);
}
int main() {
  uint4 *s, *d; hipMalloc(&s, 64 * 16); hipMalloc(&d, 64 * 16); hipMemset(s, 0x11, 64 * 16); hipMemset(d, 0xff, 64 * 16);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, s, d, 64);
  uint4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%08x %08x %08x %08x\n", h[0].x, h[1].x, h[2].w, h[3].w);
  return 0;
}
