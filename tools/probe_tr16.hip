#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int l = threadIdx.x;
  int t = l & 15, g = l >> 4;
  // lane t=4r+q of group g supplies row (g*4 + r), columns 4q..4q+3 ; row stride = stride_elems
  int r = t >> 2, q = t & 3;
  uint16_t* p = lds + (g * 4 + r) * stride_elems + q * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 40}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d(r%d,c%d)", h[l*4+j], h[l*4+j]/stride, h[l*4+j]%stride); printf("\n"); }
  }
  return 0;
}
