// Shared device/host helpers for libcgen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cgen_hip.h"

namespace cgen {

// Kernels take their parameter structs by value: 400-700 bytes of kernarg segment that hipcc reads lazily, one scalar
// load (and one s_waitcnt) per use site.  On a launch-latency-bound kernel every such load that misses the scalar cache is
// a serial ~250 ns round trip, ten cache lines deep.  Touching one dword of every 64-byte line up front turns that into
// ONE round trip; the later loads hit the scalar cache.
template <int NBYTES>
__device__ __forceinline__ void warm_kernargs() {
  typedef const uint32_t __attribute__((address_space(4))) * kptr;
  kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t acc = 0;
#pragma unroll
  for (int o = 0; o < NBYTES; o += 64) acc |= ka[o / 4];
  asm volatile("" ::"s"(acc));
}

// ----------------------------------------------------------------------------- errors
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
#define CGEN_REQUIRE(cond, ...)                       \
  do {                                                \
    if (!(cond)) return cgen::fail(CGEN_EINVAL, __VA_ARGS__); \
  } while (0)

// ----------------------------------------------------------------------------- dtypes
// The 16-bit storage format of the throughput path is IEEE binary16 (11-bit significand): at identical weights and noise its
// ELBO / counterfactual pixels sit 8x closer to the f32 path than bf16's (8-bit significand) did -- tools/bf16_trunk_sim.py,
// DESIGN section 1a -- at the same MFMA rate (v_mfma_f32_16x16x32_f16).  Activations are O(1..1e3), far from 65504; GRADIENTS
// are carried times a power-of-two loss scale (engine.py `loss_scale`) so that they stay in binary16's normal range.
// -DCGEN_H16_BF16 rebuilds the library with bfloat16 storage for an A/B measurement (one format per build, never both).
typedef uint16_t h16_t;  // raw bits
#ifdef CGEN_H16_BF16
typedef __bf16 h16n_t;  // the compiler's native type of the format
#else
typedef _Float16 h16n_t;
#endif
typedef h16n_t h16x2_t __attribute__((ext_vector_type(2)));
typedef h16n_t h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float h2f(h16_t v) {
  union { h16_t u; h16n_t h; } c;
  c.u = v;
  return (float)c.h;
}
// low / high half of a packed pair -> f32 (fp16: v_cvt_f32_f16 with a word select; bf16: a shift / a mask)
__device__ __forceinline__ float h_lo(uint32_t w) {
#ifdef CGEN_H16_BF16
  return __uint_as_float(w << 16);
#else
  union { uint32_t u; h16x2_t h; } c;
  c.u = w;
  return (float)c.h[0];
#endif
}
__device__ __forceinline__ float h_hi(uint32_t w) {
#ifdef CGEN_H16_BF16
  return __uint_as_float(w & 0xffff0000u);
#else
  union { uint32_t u; h16x2_t h; } c;
  c.u = w;
  return (float)c.h[1];
#endif
}
__device__ __forceinline__ h16_t f2h(float f) {  // round-to-nearest-even
  union { h16n_t h; h16_t u; } c;
  c.h = (h16n_t)f;
  return c.u;
}
__device__ __forceinline__ uint32_t f2h_pk(float lo, float hi) {  // two floats -> packed pair: one v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32
  union { h16x2_t h; uint32_t u; } c;
  const f32x2_t v = {lo, hi};
  c.h = __builtin_convertvector(v, h16x2_t);
  return c.u;
}
__device__ __forceinline__ f32x4_t mfma_h16(h16x8 a, h16x8 b, f32x4_t c, int, int, int) {
#ifdef CGEN_H16_BF16
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float to(float v) { return v; }
};
template <> struct Elem<h16_t> {
  static __device__ __forceinline__ float ld(const h16_t* p) { return h2f(*p); }
  static __device__ __forceinline__ void st(h16_t* p, float v) { *p = f2h(v); }
  static __device__ __forceinline__ h16_t to(float v) { return f2h(v); }
};

// ----------------------------------------------------------------------------- views
struct View {  // device-side mirror of cgen_view with typed access
  char* p;
  int64_t sn, sh, sw;
  int c;
  int cpad;
};
static inline View mk(const cgen_view& v) {
  View o;
  o.p = (char*)v.p; o.sn = v.sn; o.sh = v.sh; o.sw = v.sw; o.c = v.c; o.cpad = v.cpad;
  return o;
}

template <typename T>
__device__ __forceinline__ T* vptr(const View& v, int n, int y, int x) {
  return (T*)v.p + (n * v.sn + y * v.sh + x * v.sw);
}
// the same with 32-bit offset arithmetic (5 scalar instructions instead of ~30): the HOST must have checked that every
// element offset of the view fits in 31 bits (view_fits_i32)
template <typename T>
__device__ __forceinline__ T* vptr32(const View& v, int n, int y, int x) {
  return (T*)v.p + (n * (int)v.sn + y * (int)v.sh + x * (int)v.sw);
}
// 16-byte vector access is legal for a view iff base and all strides are multiples of 16 bytes
static inline bool vec16_ok(const cgen_view& v, int esz) {
  if (!v.p) return true;
  return (((uintptr_t)v.p) % 16 == 0) && ((v.sn * esz) % 16 == 0) && ((v.sh * esz) % 16 == 0) && ((v.sw * esz) % 16 == 0);
}
// every 16-byte channel group of the view can be fetched whole (LDS-DMA): aligned, and the channel count is a multiple
// of the group or the caller vouches for zero padding up to the next multiple
static inline bool dma_clean(const cgen_view& v, int esz) {
  const int G = 16 / esz;
  const int cg = (v.c + G - 1) / G * G;
  // the tiled kernels form per-lane offsets inside a tile with 24-bit multiplies (row stride, 64-pixel stride)
  const bool small_strides = v.sh >= 0 && v.sw >= 0 && v.sh < (1 << 24) && v.sw * 64 < (1 << 24);
  return vec16_ok(v, esz) && (v.c % G == 0 || v.cpad >= cg) && small_strides;
}

// ----------------------------------------------------------------------------- activations (vae.py:50,59)
#define CGEN_SQRT1_2 0.70710678118654752440f
#define CGEN_INV_SQRT_2PI 0.39894228040143267794f
// The erf-GELU bodies are deliberately NOT inlinable: as inline code hipcc if-converts the activation switch and
// evaluates the erf polynomial for every element even when the layer is a ReLU (measured: half the wgrad kernel's
// time).  Behind a call the GELU math sits on a real, wave-uniform branch.
__device__ __noinline__ float gelu_fwd_slow(float x) { return 0.5f * x * (1.f + erff(x * CGEN_SQRT1_2)); }  // nn.GELU() default
__device__ __noinline__ float gelu_bwd_slow(float x) {
  return 0.5f * (1.f + erff(x * CGEN_SQRT1_2)) + x * CGEN_INV_SQRT_2PI * __expf(-0.5f * x * x);
}
// bf16 path: eight elements per call (one call per 16-byte group instead of one per element) and the erf of Abramowitz &
// Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution) that shares its exponential with the density term:
// erf(x/sqrt2) = sign(x) (1 - poly(t) exp(-x^2/2)), t = 1/(1 + p |x|/sqrt2).  The f32 path keeps erff.
__device__ __forceinline__ void gelu_terms_fast(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * CGEN_SQRT1_2;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.f - poly * e;
  cdf = 0.5f * (1.f + copysignf(erf_abs, x));
  pdf = CGEN_INV_SQRT_2PI * e;
}
struct F8 { float v[8]; };
__device__ __noinline__ uint4 gelu8_fwd_h16(uint4 x) {
  uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = h_lo(w[i]), b = h_hi(w[i]);
    float ca, cb, pa, pb;
    gelu_terms_fast(a, ca, pa);
    gelu_terms_fast(b, cb, pb);
    w[i] = f2h_pk(a * ca, b * cb);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
// gelu'(x) for the eight bf16 pre-activations of a group
__device__ __noinline__ F8 gelu8_bwd_h16(uint4 x) {
  const uint32_t w[4] = {x.x, x.y, x.z, x.w};
  F8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = h_lo(w[i]), b = h_hi(w[i]);
    float ca, cb, pa, pb;
    gelu_terms_fast(a, ca, pa);
    gelu_terms_fast(b, cb, pb);
    r.v[2 * i] = ca + a * pa;
    r.v[2 * i + 1] = cb + b * pb;
  }
  return r;
}
__device__ __forceinline__ float act_fwd(int act, float x) {
  if (act == CGEN_ACT_GELU) return gelu_fwd_slow(x);
  return (act == CGEN_ACT_RELU && x <= 0.f) ? 0.f : x;  // NaN stays NaN, as torch.relu
}
__device__ __forceinline__ float act_bwd(int act, float x) {
  if (act == CGEN_ACT_GELU) return gelu_bwd_slow(x);
  return (act == CGEN_ACT_RELU && !(x > 0.f)) ? 0.f : 1.f;
}

// ----------------------------------------------------------------------------- wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// deterministic block sum for blockDim.x == 256; result valid in thread 0; `sm` >= 4 floats
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
}

// ----------------------------------------------------------------------------- Philox4x32-10 + Box-Muller
struct Philox {
  static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  // 4 x 32 random bits for (seed, offset+stream, index)
  static __device__ __forceinline__ void gen(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset ^ (stream * 0x9E3779B9u), (uint32_t)(offset >> 32) + stream};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, k0, k1);
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
  static __device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
  // 4 standard normals for element group `idx`
  static __device__ __forceinline__ void normal4(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx, float (&z)[4]) {
    uint32_t r[4];
    gen(seed, offset, stream, idx, r);
    float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
    float ra = sqrtf(-2.f * __logf(u0)), rb = sqrtf(-2.f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.28318530717958647692f * u1, &s0, &c0);
    __sincosf(6.28318530717958647692f * u3, &s1, &c1);
    z[0] = ra * c0; z[1] = ra * s0; z[2] = rb * c1; z[3] = rb * s1;
  }
  // one normal for a flat element index (uses lane idx&3 of group idx>>2)
  static __device__ __forceinline__ float normal1(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t i) {
    float z[4];
    normal4(seed, offset, stream, i >> 2, z);
    int k = (int)(i & 3);
    return k == 0 ? z[0] : k == 1 ? z[1] : k == 2 ? z[2] : z[3];
  }
};

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace cgen
