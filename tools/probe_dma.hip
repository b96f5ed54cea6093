// Staging-throughput probe (gfx950): how fast can one CU move global memory into LDS?
//   mode 0: global_load_lds b128 (LDS-DMA), NB pieces issued back to back per wave, then one wait
//   mode 1: global_load_dwordx4 into registers (NB in flight), then ds_write_b128
//   mode 2: global_load_lds b32
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_dma tools/probe_dma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

template <int MODE, int NB>
__global__ __launch_bounds__(256) void probe(const uint4* src, size_t nvec, int iters, uint32_t* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* mine = smem + wave * (NB * 1024);
  const size_t stride = (size_t)gridDim.x * 256 * NB;  // vectors per iteration over the grid
  size_t base = ((size_t)blockIdx.x * 4 + wave) * 64 * NB + lane;
  uint32_t accx = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    size_t b = (base + (size_t)it * stride) % nvec;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr)(src + b + i * 64), (lds_ptr)(mine + i * 1024), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70 & 0);  // vmcnt(0) (full wait)
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_global_load_lds((gbl_ptr)((const uint32_t*)(src + b + i * 64) + j * 64 - lane * 3), (lds_ptr)(mine + i * 1024 + j * 256), 4, 0, 0);
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_s_barrier();
    } else {
      uint4 r[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) r[i] = src[b + i * 64];
#pragma unroll
      for (int i = 0; i < NB; ++i) *(uint4*)(mine + i * 1024 + lane * 16) = r[i];
      __syncthreads();
    }
    accx ^= *(const uint32_t*)(mine + ((lane * 7 + it) & 255) * 4);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + tid] = accx;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NB>
static void run(const char* name, const uint4* src, size_t nvec, int wgs_per_cu, uint32_t* out, unsigned long long* cyc) {
  const int grid = 256 * wgs_per_cu, iters = 200;
  const size_t lds = 4 * NB * 1024;
  hipFuncSetAttribute((const void*)probe<MODE, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NB>), dim3(grid), dim3(256), lds, 0, src, nvec, 10, out, cyc);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<MODE, NB>), dim3(grid), dim3(256), lds, 0, src, nvec, iters, out, cyc);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double bytes = (double)grid * 256 * NB * 16 * iters;
  printf("%-28s NB=%2d wg/cu=%d ws=%6.0fMB  %8.1f GB/s  %6.3f ms  cyc/iter(wg0)=%llu (=%.0f cyc per KiB per wave)\n", name, NB, wgs_per_cu,
         nvec * 16 / 1e6, bytes / ms / 1e6, ms, h[0] / iters, (double)h[0] / iters / NB);
}

int main() {
  const size_t big = (size_t)1 << 30, small = (size_t)2 << 20;
  uint4* src; hipMalloc(&src, big); hipMemset(src, 1, big);
  uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 8 * 8);
  for (size_t ws : {big, small}) {
    const size_t nvec = ws / 16;
    for (int occ : {1, 2, 4}) {
      run<0, 4>("dma_b128", src, nvec, occ, out, cyc);
      run<0, 8>("dma_b128", src, nvec, occ, out, cyc);
      run<0, 16>("dma_b128", src, nvec, occ, out, cyc);
      run<1, 4>("reg_x4+ds_write", src, nvec, occ, out, cyc);
      run<1, 8>("reg_x4+ds_write", src, nvec, occ, out, cyc);
      run<1, 16>("reg_x4+ds_write", src, nvec, occ, out, cyc);
      run<2, 4>("dma_b32", src, nvec, occ, out, cyc);
    }
  }
  return 0;
}
