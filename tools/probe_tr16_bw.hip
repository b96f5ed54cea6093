// Throughput of ds_read_b64_tr_b16 under the address patterns of the streaming weight-gradient kernel (csrc/wgrad3.hip):
// lane (g = l >> 4, r = (l >> 2) & 3, q = l & 3) reads 8 bytes at  (g >> 1) * A + r * SP + (g & 1) * B + q * 8 + (second read: + 4 SP).
// Prints cycles per wave-instruction with 4 / 8 waves of one workgroup hammering the LDS (conflict-free = 2 LDS cycles per read
// per wave => 8 / 16 cycles per read at 4 / 8 waves when LDS-bound).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_tr tools/probe_tr16_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lp;
__global__ void k(unsigned long long* out, int A, int SP, int B, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, g = l >> 4, r = (l >> 2) & 3, q = l & 3;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint32_t a0 = base + (g >> 1) * A + r * SP + (g & 1) * B + q * 8;
  const uint32_t a1 = a0 + 4 * SP;
  s16x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    s16x4 v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[2 * u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a0 + u * 64));
      v[2 * u + 1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a1 + u * 64));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (l == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (acc[0] == 12345 && acc[1] == 77) out[63] = 1;
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 2000;
  struct Pat { const char* name; int A, SP, B; };
  Pat pats[] = {
      {"P sp 64 (32 ch)", 8 * 64, 64, 32},   {"P sp 128 (64 ch, unpadded)", 8 * 128, 128, 32}, {"P sp 192 (96 ch)", 8 * 192, 192, 32},
      {"P sp 256 (128 ch, unpadded)", 8 * 256, 256, 32}, {"P sp 320 (128 ch padded)", 8 * 320, 320, 32}, {"P sp 448", 8 * 448, 448, 32},
      {"P sp 80", 8 * 80, 80, 32},  {"P sp 144", 8 * 144, 144, 32}, {"P sp 208", 8 * 208, 208, 32}, {"P sp 272", 8 * 272, 272, 32},
      {"S sp 48 (24 ch) B 32", 8 * 48, 48, 32}, {"S sp 48, B = +1 px (48)", 8 * 48, 48, 48}, {"S sp 80 (40 ch) B 32", 8 * 80, 80, 32},
      {"S sp 16 (8 ch) B 16", 8 * 16, 16, 16}, {"S sp 32 B 32", 8 * 32, 32, 32}, {"S sp 64 B 32", 8 * 64, 64, 32},
      {"good T10 layout: 128 B per group, groups 512 B apart", 1024, 32, 512},
      {"all lanes same 128 B", 0, 32, 0},
  };
  for (auto& p : pats) {
    for (int waves : {1, 2, 4, 8, 16}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 64 * 1024, 0, d, p.A, p.SP, p.B, iters);
      unsigned long long h[16];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      unsigned long long mx = 0;
      for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
      printf("%-55s %d waves: %6.1f cycles per read per wave\n", p.name, waves, (double)mx / (iters * 8.0));
    }
  }
  return 0;
}
