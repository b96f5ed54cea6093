// Implicit-GEMM convolution for gfx950 (MI355X): forward / data-gradient (cgen_conv2d) and weight-gradient
// (cgen_conv2d_wgrad, cgen_conv2d_wgrad_batch_*), plus the multi-tensor weight-image prep and split-K reduction.
//
// GEMM view (operands swapped so a lane owns consecutive output channels of one pixel):
//   D[co][px] = sum_k  Wimg[co][k] * A[k][px],   k = (tap, segment, channel) with channels 8-granular
//   A[k][px]  = act( seg_s[n, y+dy, x+dx, c] )   (zero outside the image; act(0) == 0)
// f32 path : v_mfma_f32_16x16x4_f32  (exact f32 fmaf chains -> the 1e-4 ELBO parity path)
// bf16 path: v_mfma_f32_16x16x32_bf16 (f32 accumulate)
//
// Kernels in this file (DESIGN.md section 3 has the when/why and the measurements):
//   conv_kernel<T,NTC>            generic gather kernel (any view, f32 on tiny images)
//   conv_tile_kernel<T,NTC,KS>    one 8x16 halo tile + weight slab in LDS per workgroup (f32; bf16 multi-window shapes)
//   conv_px_kernel<NP,KS>         lean persistent kernel for short-K convs: slab in LDS once, 16-byte epilogue from registers
//   conv_ws_kernel<NTC,NKW>       weight-stationary persistent kernel for long-K / narrow-output convs (weights in registers)
//   conv_smallp_kernel<KS,..>     K split over the waves, operands straight from global memory (<= ~6000 pixels per batch)
//   wgrad_kernel<T,NTC>           generic weight gradient (f32 MFMA)
//   wgrad_tile_kernel / wgrad_tile_batched_kernel<NCF,NJW,KS>   bf16 weight gradient with LDS transpose reads;
//                                 the batched form runs the workgroups of many problems in one launch
//   wprep_kernel / wred_kernel    OIHW f32 -> weight images; split-K partials -> flat OIHW gradient
// What bounds them on MI355X is instruction issue (VALU/SALU per MFMA), not MFMA or HBM: see DESIGN.md 3.4.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "wgrad3.h"

#ifdef CGEN_NO_SETPRIO
#define CGEN_SETPRIO() do {} while (0)
#else
#define CGEN_SETPRIO() __builtin_amdgcn_s_setprio(3)
#endif

namespace cgen {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Division by a run-time constant without v_rcp/loops: q = umulhi(n, mul) >> shift, exact for 0 <= n < 2^31
// (round-up method: mul = ceil(2^(32+shift) / d)).  Built on the host, used in the per-lane address generation.
struct FastDiv {
  uint32_t mul, shift, d, pad;
};
static inline FastDiv mk_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d; f.pad = 0;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t sh = 0;
  while ((1u << sh) < d) ++sh;  // ceil(log2 d)
  f.shift = sh;
  f.mul = (uint32_t)((((uint64_t)1 << (32 + sh)) + d - 1) / d - ((uint64_t)1 << 32));
  return f;
}
// q = (umulhi(n, mul) + n) >> shift   (the "add" variant keeps mul in 32 bits)
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {  // d == 1: mul = shift = 0 -> n (no branch)
  return (int)(((uint64_t)__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shift);
}

#define CONV_PT 128  // pixels per workgroup
#define CONV_BK 32   // K per step
#define CONV_LDK 40  // LDS row stride in elements (bank-conflict-free 16B fragment reads)

struct ConvP {
  int N, H, W, KS, pad, nseg, act, dact, Co, P, taps, ctot8, krow;
  View seg[CGEN_MAX_SEG];
  int seg_koff[CGEN_MAX_SEG];  // offset of the segment inside the 8-granular concatenated channel axis
  int seg_vec[CGEN_MAX_SEG];
  const void* w;
  const float* bias;
  View out, aux, res1, res2;
  long long out_rem, r1_rem;  // byte offsets of the remainder planes of out / res1 (f16 residual trunk: value = hi + rem); 0 = none
  int epi_vec, force_generic, dma_ok, epi_vec16;
  int tap0, tap1;  // taps that can touch the image (a 3x3 conv on a 1x1 image only ever sees its centre tap)
  int split, pad_s;  // CGEN_F32S: f32 tensors, products as three binary16 MFMAs (hi*hi + hi*lo + lo*hi), f32 accumulate
};

// 4-element (16B f32 / 8B bf16) vector access
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&v)[4]) {
  const float4 t = *(const float4*)p;
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<h16_t>(const h16_t* p, float (&v)[4]) {
  const uint2 t = *(const uint2*)p;
  v[0] = h_lo(t.x); v[1] = h_hi(t.x);
  v[2] = h_lo(t.y); v[3] = h_hi(t.y);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void st4<h16_t>(h16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = f2h_pk(v[0], v[1]);
  t.y = f2h_pk(v[2], v[3]);
  *(uint2*)p = t;
}
template <typename T, int N> union Pack {
  T e[N];
  uint4 v4;
};

// Remainder plane of the f16 residual trunk: out = rn16(v) goes to the tensor, rn16(v - out) to the plane `rem` bytes further
// (include/cgen_hip.h, cgen_conv_args): hi + rem carries ~22 significant bits through the ~100 residual updates of a pass.
__device__ __forceinline__ float h16_rem(float v) { return v - h2f(f2h(v)); }

// (bias + acc) * act'(aux) + res1 + res2 for 4 consecutive output channels of one pixel
template <typename T, bool REM = true>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, const f32x4& a, int pn, int py, int px, int co) {
  if (co >= p.Co) {  // padding channels [Co, out.cpad) are written as zeros so that consumers may fetch whole 16-byte groups
    if (co < p.out.cpad) {
      T* z = vptr<T>(p.out, pn, py, px);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e < p.out.cpad) z[co + e] = (T)0;
    }
    return;
  }
  T* optr = vptr<T>(p.out, pn, py, px);
  const T* aptr = p.aux.p ? vptr<T>(p.aux, pn, py, px) : nullptr;
  const T* r1 = p.res1.p ? vptr<T>(p.res1, pn, py, px) : nullptr;
  const T* r2 = p.res2.p ? vptr<T>(p.res2, pn, py, px) : nullptr;
  const T* r1l = (REM && r1 && p.r1_rem) ? (const T*)((const char*)r1 + p.r1_rem) : nullptr;
  T* orem = (REM && p.out_rem) ? (T*)((char*)optr + p.out_rem) : nullptr;
  float v[4] = {a[0], a[1], a[2], a[3]};
  if ((co + 4 <= p.Co) && p.epi_vec) {
    if (p.bias) {
      const float4 bb = *(const float4*)(p.bias + co);
      v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
    }
    float t4[4];
    if (aptr) {
      ld4<T>(aptr + co, t4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= act_bwd(p.dact, t4[e]);
    }
    if (r1) {
      ld4<T>(r1 + co, t4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += t4[e];
    }
    if (r2) {
      ld4<T>(r2 + co, t4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += t4[e];
    }
    if constexpr (REM && sizeof(T) == 2) {
      if (r1l) {
        ld4<T>(r1l + co, t4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += t4[e];
      }
      if (orem) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[e] = h16_rem(v[e]);
        st4<T>(orem + co, t4);
      }
    }
    st4<T>(optr + co, v);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (co + e < p.Co) {
        float u = v[e] + (p.bias ? p.bias[co + e] : 0.f);
        if (aptr) u *= act_bwd(p.dact, Elem<T>::ld(aptr + co + e));
        if (r1) u += Elem<T>::ld(r1 + co + e);
        if (r2) u += Elem<T>::ld(r2 + co + e);
        if constexpr (REM && sizeof(T) == 2) {
          if (r1l) u += Elem<T>::ld(r1l + co + e);
          if (orem) Elem<T>::st(orem + co + e, h16_rem(u));
        }
        Elem<T>::st(optr + co + e, u);
      } else if (co + e < p.out.cpad) {
        optr[co + e] = (T)0;
      }
    }
  }
}

// Split form of the 8-channel epilogue: on gfx950 loads and stores share one in-order counter (vmcnt), so a load issued
// after a store cannot be waited for without also waiting for the store's acknowledgement.  An epilogue that interleaves
// "load aux/residual -> store" per chunk therefore serialises a full load + store round trip per chunk.  The kernels
// issue ALL operand loads of a tile first (Epi8, before or during the MFMA loop) and finish with pure math + stores.
struct Epi8 {
  uint4 a, r1, r2, r1l;
};
struct Bias8 {
  float4 b0, b1;
};
__device__ __forceinline__ void bias8_load(const ConvP& p, int co, Bias8& l) {
  l.b0 = l.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) { l.b0 = *(const float4*)(p.bias + co); l.b1 = *(const float4*)(p.bias + co + 4); }
}
template <bool REM>
__device__ __forceinline__ void epi8_load(const ConvP& p, int pn, int py, int px, int co, Epi8& l) {
  typedef h16_t T;
  l.a = l.r1 = l.r2 = l.r1l = make_uint4(0, 0, 0, 0);
  if (p.aux.p) l.a = *(const uint4*)(vptr<T>(p.aux, pn, py, px) + co);
  if (p.res1.p) l.r1 = *(const uint4*)(vptr<T>(p.res1, pn, py, px) + co);
  if constexpr (REM) {
    if (p.r1_rem) l.r1l = *(const uint4*)((const char*)(vptr<T>(p.res1, pn, py, px) + co) + p.r1_rem);
  }
  if (p.res2.p) l.r2 = *(const uint4*)(vptr<T>(p.res2, pn, py, px) + co);
}
// v = (bias + v) * act'(aux) + res1 + res2, rounded once, stored as one 16-byte chunk (fast path only: whole aligned chunk)
template <bool REM>
__device__ __forceinline__ void epi8_finish(const ConvP& p, float (&v)[8], const Epi8& l, const Bias8& bb, int pn, int py, int px, int co) {
  typedef h16_t T;
  v[0] += bb.b0.x; v[1] += bb.b0.y; v[2] += bb.b0.z; v[3] += bb.b0.w; v[4] += bb.b1.x; v[5] += bb.b1.y; v[6] += bb.b1.z; v[7] += bb.b1.w;
  Pack<T, 8> t;
  if (p.aux.p) {
    t.v4 = l.a;
    if (p.dact == CGEN_ACT_GELU) {
      const F8 gp = gelu8_bwd_h16(l.a);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gp.v[e];
    } else if (p.dact == CGEN_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = h2f(t.e[e]) > 0.f ? v[e] : 0.f;
    }
  }
  if (p.res1.p) {
    t.v4 = l.r1;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += h2f(t.e[e]);
  }
  if (p.res2.p) {
    t.v4 = l.r2;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += h2f(t.e[e]);
  }
  if (REM && p.r1_rem) {
    t.v4 = l.r1l;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += h2f(t.e[e]);
  }
  T* optr = vptr<T>(p.out, pn, py, px) + co;
  if (REM && p.out_rem) {
#pragma unroll
    for (int e = 0; e < 8; ++e) t.e[e] = f2h(h16_rem(v[e]));
    *(uint4*)((char*)optr + p.out_rem) = t.v4;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) t.e[e] = f2h(v[e]);
  *(uint4*)optr = t.v4;
}

// 8 consecutive output channels of one pixel (16-byte bf16 I/O): same math as conv_epilogue, used by the LDS-staged
// epilogues where consecutive lanes own consecutive 16-byte chunks of a pixel row (fully coalesced stores / loads)
template <bool REM>
__device__ __forceinline__ void conv_epilogue8_h16(const ConvP& p, float (&v)[8], int pn, int py, int px, int co) {
  typedef h16_t T;
  if (co >= p.Co) {
    if (co < p.out.cpad) {
      T* z = vptr<T>(p.out, pn, py, px);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (co + e < p.out.cpad) z[co + e] = 0;
    }
    return;
  }
  T* optr = vptr<T>(p.out, pn, py, px) + co;
  const T* aptr = p.aux.p ? vptr<T>(p.aux, pn, py, px) + co : nullptr;
  const T* r1 = p.res1.p ? vptr<T>(p.res1, pn, py, px) + co : nullptr;
  const T* r2 = p.res2.p ? vptr<T>(p.res2, pn, py, px) + co : nullptr;
  const T* r1l = (REM && r1 && p.r1_rem) ? (const T*)((const char*)r1 + p.r1_rem) : nullptr;
  T* orem = (REM && p.out_rem) ? (T*)((char*)optr + p.out_rem) : nullptr;
  if (co + 8 <= p.Co && p.epi_vec16) {
    if (p.bias) {
      const float4 b0 = *(const float4*)(p.bias + co), b1 = *(const float4*)(p.bias + co + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    Pack<T, 8> t;
    if (aptr) {
      t.v4 = *(const uint4*)aptr;
      if (p.dact == CGEN_ACT_GELU) {
        const F8 gp = gelu8_bwd_h16(t.v4);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gp.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= act_bwd(p.dact, h2f(t.e[e]));
      }
    }
    if (r1) {
      t.v4 = *(const uint4*)r1;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += h2f(t.e[e]);
    }
    if (r2) {
      t.v4 = *(const uint4*)r2;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += h2f(t.e[e]);
    }
    if (r1l) {
      t.v4 = *(const uint4*)r1l;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += h2f(t.e[e]);
    }
    if (orem) {
#pragma unroll
      for (int e = 0; e < 8; ++e) t.e[e] = f2h(h16_rem(v[e]));
      *(uint4*)orem = t.v4;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) t.e[e] = f2h(v[e]);
    *(uint4*)optr = t.v4;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (co + e < p.Co) {
        float u = v[e] + (p.bias ? p.bias[co + e] : 0.f);
        if (aptr) u *= act_bwd(p.dact, h2f(aptr[e]));
        if (r1) u += h2f(r1[e]);
        if (r2) u += h2f(r2[e]);
        if (r1l) u += h2f(r1l[e]);
        if (orem) orem[e] = f2h(h16_rem(u));
        optr[e] = f2h(u);
      } else if (co + e < p.out.cpad) {
        optr[e] = 0;
      }
    }
  }
}

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; };
template <> struct Frag<h16_t> { typedef h16x8 type; };

template <typename T, int NTC>
__global__ __launch_bounds__(256) void conv_kernel(ConvP p) {
  constexpr int G = 16 / sizeof(T);   // elements per 16-byte group
  constexpr int NG = 16 / G;          // groups per thread for the activation tile (16 elements per thread)
  constexpr int WROWS = NTC * 16;
  constexpr int WGROUPS = WROWS * (CONV_BK / G);
  constexpr int WPT = (WGROUPS + 255) / 256;
  __shared__ __attribute__((aligned(16))) T Xs[CONV_PT * CONV_LDK];
  __shared__ __attribute__((aligned(16))) T Ws[WROWS * CONV_LDK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = tid >> 1, half = tid & 1;
  const int m = blockIdx.x * CONV_PT + row;
  const bool mvalid = m < p.P;
  int n = 0, y = 0, x = 0;
  if (mvalid) {
    n = m / (p.H * p.W);
    int r = m - n * p.H * p.W;
    y = r / p.W;
    x = r - y * p.W;
  }
  const int co_base = blockIdx.y * WROWS;

  f32x4 acc[NTC][2];
#pragma unroll
  for (int t = 0; t < NTC; ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f) acc[t][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- K-step iterator state: (tap, seg, c0)
  int tap = p.tap0, s = 0, c0 = 0;
  uint4 xr[NG];   // staged activation groups (raw 16 bytes each)
  uint4 wr[WPT];  // staged weight groups

  auto load_step = [&](int tap_, int s_, int c0_) {
    const int dy = tap_ / p.KS - p.pad, dx = tap_ % p.KS - p.pad;
    const int yy = y + dy, xx = x + dx;
    const bool inb = mvalid && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    const View& sv = p.seg[s_];
    const T* src = inb ? vptr<T>(sv, n, yy, xx) : nullptr;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int cs = c0_ + half * 16 + g * G;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (inb) {
        if (p.seg_vec[s_] && cs + G <= sv.c) {
          v = *(const uint4*)(src + cs);
        } else if (cs < sv.c) {
          Pack<T, G> tmp;
#pragma unroll
          for (int e = 0; e < G; ++e) tmp.e[e] = (cs + e < sv.c) ? src[cs + e] : (T)0;
          v = tmp.v4;
        }
      }
      if (p.act != CGEN_ACT_NONE) {
        Pack<T, G> tv;
        tv.v4 = v;
#pragma unroll
        for (int e = 0; e < G; ++e) tv.e[e] = Elem<T>::to(act_fwd(p.act, Elem<T>::ld(&tv.e[e])));
        v = tv.v4;
      }
      xr[g] = v;
    }
    const T* wbase = (const T*)p.w + (size_t)tap_ * p.ctot8 + p.seg_koff[s_] + c0_;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int gi = tid + i * 256;
      if (gi < WGROUPS) {
        const int r = gi / (CONV_BK / G), kg = gi % (CONV_BK / G);
        wr[i] = *(const uint4*)(wbase + (size_t)min(co_base + r, ((p.Co + 15) & ~15) - 1) * p.krow + kg * G);  // rows past the image: clamped (never stored)
      }
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int g = 0; g < NG; ++g) *(uint4*)(Xs + row * CONV_LDK + half * 16 + g * G) = xr[g];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int gi = tid + i * 256;
      if (gi < WGROUPS) {
        const int r = gi / (CONV_BK / G), kg = gi % (CONV_BK / G);
        *(uint4*)(Ws + r * CONV_LDK + kg * G) = wr[i];
      }
    }
  };
  auto advance = [&]() -> bool {  // returns false when the K loop is exhausted
    c0 += CONV_BK;
    if (c0 >= p.seg[s].c) {
      c0 = 0;
      ++s;
      if (s >= p.nseg) {
        s = 0;
        ++tap;
      }
    }
    return tap < p.tap1;
  };

  load_step(tap, s, c0);
  bool more = true;
  while (more) {
    __syncthreads();  // previous step's fragment reads are done
    store_step();
    __syncthreads();
    more = advance();
    if (more) load_step(tap, s, c0);  // in flight during the MFMAs below
    const int fr = lane & 15, fg = lane >> 4;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f32x4 b[2], a[NTC];
#pragma unroll
        for (int f = 0; f < 2; ++f) b[f] = *(const f32x4*)(Xs + (wave * 32 + f * 16 + fr) * CONV_LDK + kk * 16 + fg * 4);
#pragma unroll
        for (int t = 0; t < NTC; ++t) a[t] = *(const f32x4*)(Ws + (t * 16 + fr) * CONV_LDK + kk * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][j], b[f][j], acc[t][f], 0, 0, 0);
      }
    } else {
      h16x8 b[2], a[NTC];
#pragma unroll
      for (int f = 0; f < 2; ++f) b[f] = *(const h16x8*)(Xs + (wave * 32 + f * 16 + fr) * CONV_LDK + fg * 8);
#pragma unroll
      for (int t = 0; t < NTC; ++t) a[t] = *(const h16x8*)(Ws + (t * 16 + fr) * CONV_LDK + fg * 8);
#pragma unroll
      for (int t = 0; t < NTC; ++t)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[t][f] = mfma_h16(a[t], b[f], acc[t][f], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns pixel (lane&15) of fragment f and channels (lane>>4)*4 .. +4 of fragment t
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int pm = blockIdx.x * CONV_PT + wave * 32 + f * 16 + (lane & 15);
    if (pm >= p.P) continue;
    const int pn = pm / (p.H * p.W);
    const int pr = pm - pn * p.H * p.W;
    const int py = pr / p.W, px = pr - py * p.W;
#pragma unroll
    for (int t = 0; t < NTC; ++t) conv_epilogue<T>(p, acc[t][f], pn, py, px, co_base + t * 16 + (lane >> 4) * 4);
  }
}

static inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// ----------------------------------------------------------------------------- halo-tiled kernel (H >= 8, W >= 16)
// One workgroup = one 8x16 spatial tile of one image (4 waves, each 2 rows of 16 pixels).
// The (8+2h)x(16+2h) input halo tile -- all segments concatenated, i.e. the reference's torch.cat materialised only in
// LDS -- and the matching weight slab are staged ONCE per channel window; then the whole K axis
//     K = (tap, channel) flattened with channels at 8-element granularity
// runs out of LDS with no further barrier.  Each input element is fetched from HBM once per (tile, co-tile) instead of
// KS*KS times, narrow inputs (Ci = 8, 16, 24) pack several taps into one 32-wide MFMA K-step instead of padding every tap
// to 32, and LDS is sized at run time to the window so narrow layers keep 4-6 workgroups per CU in flight.
#define TILE_H 8
#define TILE_W 16

struct TileP {
  int tiles_x, tiles_y;
  int cw;    // channel window per pass (multiple of 8; == ctot8 when everything fits in one pass)
  int ldc;   // LDS row stride of the halo tile (elements)
  int kp;    // K length of a pass, padded to the MFMA K-step
  int ldw;   // LDS row stride of the weight slab (elements)
  FastDiv d_gprw, d_gprx, d_cw;
  int dbg, pad0;  // ablation mask (CGEN_TILE_DBG): 1 skip weight DMA, 2 skip halo DMA, 4 skip act pass, 8 skip K loop, 16 skip epilogue
};

__device__ uint4 g_zero16[4];  // 64 bytes of zeros: DMA source for out-of-image / padding groups

// in-register activation of one 16-byte group
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ uint4 act_group(uint4 v, int act) {
  constexpr int G = 16 / sizeof(T);
  Pack<T, G> tv;
  tv.v4 = v;
  if constexpr (sizeof(T) == 2) {
    if (act == CGEN_ACT_RELU) {
      // bf16 ReLU == signed 16-bit max(x, 0) on the raw bits (negative floats have the sign bit set): one v_pk_max_i16
      // per channel pair.  (-0.0 -> +0.0; a NaN with the sign bit set becomes 0, torch keeps it -- not reachable here)
      union { uint32_t u; s16x2 s; } c;
      uint32_t* w = (uint32_t*)&tv.v4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c.u = w[e];
        c.s = __builtin_elementwise_max(c.s, (s16x2){0, 0});
        w[e] = c.u;
      }
      return tv.v4;
    }
    if (act == CGEN_ACT_GELU) {
      return gelu8_fwd_h16(v);
    }
    return v;
  }
#pragma unroll
  for (int e = 0; e < G; ++e) tv.e[e] = Elem<T>::to(act_fwd(act, Elem<T>::ld(&tv.e[e])));
  return tv.v4;
}


// In-place activation of the 1-KiB DMA pieces this wave fetched itself (pieces wave, wave + 4, ...; one 16-byte group per lane).
// Four pieces per round: all four reads are issued before the first write -- the plain `*ptr = act(*ptr)` loop is a chain of
// LDS read -> wait -> math -> write round trips (hipcc may not move a later piece's read above an earlier piece's write: it
// cannot prove they do not alias), ~250 cycles per piece where the weight-gradient kernel's cycle stamps showed 1.3-2.4 k per
// tile in this pass.  Lanes that carry no data hold zeros (or sit in never-read padding slots): act(0) = 0, so no lane mask.
template <typename T>
__device__ __forceinline__ void act_pieces(char* buf, const int wave, const int lane, const int npieces, const int act) {
  char* base = buf + lane * 16;
  int pi = wave;
  for (; pi + 12 < npieces; pi += 16) {
    uint4* p0 = (uint4*)(base + pi * 1024);
    uint4* p1 = (uint4*)(base + (pi + 4) * 1024);
    uint4* p2 = (uint4*)(base + (pi + 8) * 1024);
    uint4* p3 = (uint4*)(base + (pi + 12) * 1024);
    const uint4 v0 = *p0, v1 = *p1, v2 = *p2, v3 = *p3;
    const uint4 r0 = act_group<T>(v0, act), r1 = act_group<T>(v1, act), r2 = act_group<T>(v2, act), r3 = act_group<T>(v3, act);
    *p0 = r0; *p1 = r1; *p2 = r2; *p3 = r3;
  }
  if (pi + 4 < npieces) {  // two left (or three: the third goes below)
    uint4* p0 = (uint4*)(base + pi * 1024);
    uint4* p1 = (uint4*)(base + (pi + 4) * 1024);
    const uint4 v0 = *p0, v1 = *p1;
    const uint4 r0 = act_group<T>(v0, act), r1 = act_group<T>(v1, act);
    *p0 = r0; *p1 = r1;
    pi += 8;
  }
  for (; pi < npieces; pi += 4) {
    uint4* p0 = (uint4*)(base + pi * 1024);
    *p0 = act_group<T>(*p0, act);
  }
}

// Split-operand form of an f32 value for the binary16 MFMAs of the CGEN_F32S path: v = hi + lo' * 2^-11 (+ O(2^-22 |v|)),
// hi = rn16(v), lo' = rn16((v - hi) * 2^11).  The scaled remainder has the magnitude of hi's last place times 2^11, i.e. of hi
// itself: it is a NORMAL binary16 number whenever hi is (an unscaled remainder of |v| < 0.125 would be subnormal), and the cross
// products are summed in an accumulator of their own that joins the main one scaled by 2^-11 at the end.
// In: 8 consecutive f32 (two 16-byte groups).  Out: the same 32 bytes as [8 x hi][8 x lo'].
__device__ __forceinline__ void split8_f32(const uint4 a, const uint4 b, const int act, uint4& hi, uint4& lo) {
  float v[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
  _Float16 h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (act != CGEN_ACT_NONE) v[e] = act_fwd(act, v[e]);
    h[e] = (_Float16)v[e];
    l[e] = (_Float16)((v[e] - (float)h[e]) * 2048.f);
  }
  union { _Float16 f[8]; uint4 q; } uh, ul;
#pragma unroll
  for (int e = 0; e < 8; ++e) { uh.f[e] = h[e]; ul.f[e] = l[e]; }
  hi = uh.q; lo = ul.q;
}
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <typename T, int NTC, int KS, bool REM, bool SPLIT = false>
__global__ __launch_bounds__(256, NTC == 4 ? 3 : 4) void conv_tile_kernel(ConvP p, TileP q) {  // REM: remainder planes of the f16 trunk (own instance: the plain one keeps its registers)
  CGEN_SETPRIO();  // the chain's waves win issue arbitration over background weight-gradient waves on the same SIMD
  static_assert(!SPLIT || sizeof(T) == 4, "split operands are the f32-storage path");
  constexpr int G = 16 / sizeof(T);
  constexpr int HALO = KS / 2, HH = TILE_H + 2 * HALO, HW = TILE_W + 2 * HALO, HPX = HH * HW;
  constexpr int TAPS = KS * KS;
  constexpr int WROWS = NTC * 16;
  constexpr int KSTEP = (sizeof(T) == 4 && !SPLIT) ? 16 : 32;
  constexpr int GK = KSTEP / 4;  // K elements one lane contributes per K-step (f32: 4 via the j-trick, 16-bit and split: 8)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS image: [weight slab: WROWS rows of ldw] [halo tile: HPX rows of ldc], each region padded to whole 1-KiB pieces
  const int wgroups = WROWS * (q.ldw / G), wpieces = (wgroups + 63) >> 6;
  const int xgroups = HPX * (q.ldc / G), xpieces = (xgroups + 63) >> 6;
  T* Ws = (T*)smem;
  T* Xs = (T*)(smem + (size_t)wpieces * 1024);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % q.tiles_x; b /= q.tiles_x;
  const int ty = b % q.tiles_y;
  const int n = b / q.tiles_y;
  const int y0 = ty * TILE_H, x0 = tx * TILE_W;
  const int co_base = blockIdx.y * WROWS;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[NTC][2], accx[SPLIT ? NTC : 1][2];  // (accx: the cross products hi * lo' + lo' * hi of the split form, in units of 2^-11)
#pragma unroll
  for (int t = 0; t < NTC; ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f) acc[t][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < (SPLIT ? NTC : 1); ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f) accx[t][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int gpr_w = q.ldw / G, gpr_x = q.ldc / G;  // 16-byte groups per LDS row (incl. padding)
  const bool single = q.cw == p.ctot8;

  // epilogue operands (bf16): issued first, in flight with the DMAs and the K loop (see Epi8)
  Epi8 epl[NTC];
  Bias8 ebias;
  bool efast[NTC];
  if constexpr (sizeof(T) == 2) {
    constexpr int CPP = NTC * 2;  // 16-byte chunks per pixel; divides 64, so a lane's chunk (=> bias) is the same for every k
    ebias.b0 = ebias.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.epi_vec16 && co_base + (lane % CPP) * 8 + 8 <= p.Co) bias8_load(p, co_base + (lane % CPP) * 8, ebias);
#pragma unroll
    for (int k = 0; k < NTC; ++k) {
      const int idx = lane + 64 * k;
      const int pl = idx / CPP, ch = idx % CPP;
      const int py = y0 + wave * 2 + (pl >> 4), px = x0 + (pl & 15);
      const int co = co_base + ch * 8;
      efast[k] = py < p.H && px < p.W && p.epi_vec16 && co + 8 <= p.Co && !(q.dbg & 16);
      if (efast[k]) epi8_load<REM>(p, n, py, px, co, epl[k]);
    }
  }

  for (int cA = 0; cA < p.ctot8; cA += q.cw) {
    const int cw = min(q.cw, p.ctot8 - cA);  // last window may be narrower (still a multiple of 8)
    const int kend = TAPS * cw;
    const int kreal = (kend + KSTEP - 1) / KSTEP * KSTEP;
    if (cA > 0) __syncthreads();  // the previous pass's fragment reads are done
    // ---- weights: LDS row [co][tap*cw + c'] <- image row [co][tap*ctot8 + cA + c']   (global -> LDS DMA, no VGPR staging)
    if (!(q.dbg & 1))
    for (int piece = wave; piece < wpieces; piece += 4) {
      const int pu = __builtin_amdgcn_readfirstlane(piece);
      const int gi = pu * 64 + lane;
      const int r = fdiv(gi, q.d_gprw), k = (gi - r * gpr_w) * G;
      if (gi < wgroups && k < kreal) {
        // rows past the image (the last output-channel tile of a ragged Co) are clamped: their outputs are never stored, and
        // reading past the LAST image of the buffer faulted once in ~2 500 fuzz cases (tools/fuzz_conv.py 150 24)
        const T* row = (const T*)p.w + (size_t)min(co_base + r, ((p.Co + 15) & ~15) - 1) * p.krow;
        const T* src;
        if (single) {
          src = row + k;
        } else {
          const int tap = cw == q.cw ? fdiv(k, q.d_cw) : k / cw, c = k - tap * cw;
          src = tap < TAPS ? row + tap * p.ctot8 + cA + c : (const T*)g_zero16;
        }
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(smem + (size_t)pu * 1024), 16, 0, 0);
      }
    }
    // ---- halo tile: virtual concat of the segments; zeros outside the image / past a segment's channels
    if (!(q.dbg & 2))
    for (int piece = wave; piece < xpieces; piece += 4) {
      const int pu = __builtin_amdgcn_readfirstlane(piece);
      const int gi = pu * 64 + lane;
      const int hp = fdiv(gi, q.d_gprx), cl = (gi - hp * gpr_x) * G;  // channel inside the window
      if (gi < xgroups && cl < cw) {
        const int c = cA + cl;
        const int yy = y0 + hp / HW - HALO, xx = x0 + hp % HW - HALO;
        int sidx = 0;
#pragma unroll
        for (int k = 1; k < CGEN_MAX_SEG; ++k) sidx += (k < p.nseg && c >= p.seg_koff[k]) ? 1 : 0;
        View sv = p.seg[0];
        int koff = p.seg_koff[0];
#pragma unroll
        for (int k = 1; k < CGEN_MAX_SEG; ++k)
          if (sidx == k) { sv = p.seg[k]; koff = p.seg_koff[k]; }
        const int cs = c - koff;
        const bool inside = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && cs < sv.c;
        char* dst = (char*)Xs + (size_t)pu * 1024;
        // no LDS store may sit between the DMAs (hipcc would drain vmcnt before it and serialise them): the host only
        // selects this kernel for dma_clean() segments, so every group is a whole 16-byte fetch or zeros
        const T* src = inside ? vptr<T>(sv, n, yy, xx) + cs : (const T*)g_zero16;
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)dst, 16, 0, 0);
      }
    }
    __syncthreads();  // hipcc drains vmcnt (incl. the LDS DMA) before the barrier
    if constexpr (SPLIT) {
      // activation (in f32) and the split, once per element, in place: every 32-byte run of 8 f32 becomes [8 x hi][8 x lo'] -- the
      // halo tile (cw / 8 runs per pixel) and the weight slab (kreal / 8 runs per row; zero padding splits to zeros)
      const int xr8 = cw >> 3, nx = HPX * xr8, wr8 = kreal >> 3, nw = WROWS * wr8;
#pragma unroll 1
      for (int i = tid; i < nx + nw; i += 256) {
        char* ptr;
        int act;
        if (i < nx) {
          const int rr = i / xr8, cc = i - rr * xr8;
          ptr = (char*)(Xs + rr * q.ldc + cc * 8); act = p.act;
        } else {
          const int j = i - nx, rr = j / wr8, cc = j - rr * wr8;
          ptr = (char*)(Ws + rr * q.ldw + cc * 8); act = CGEN_ACT_NONE;
        }
        uint4 hi, lo;
        split8_f32(*(const uint4*)ptr, *(const uint4*)(ptr + 16), act, hi, lo);
        *(uint4*)ptr = hi;
        *(uint4*)(ptr + 16) = lo;
      }
      __syncthreads();
    } else
    if (p.act != CGEN_ACT_NONE && !(q.dbg & 4)) {  // activation once per element, in place (not once per tap at fragment-read time)
      const int rstep = fdiv(256, q.d_gprx), cstep = 256 - rstep * gpr_x;
      int rr = fdiv(tid, q.d_gprx), cc = tid - rr * gpr_x;
#pragma unroll 1
      while (rr < HPX) {
        if (cc * G < cw) {
          uint4* ptr = (uint4*)(Xs + rr * q.ldc + cc * G);
          *ptr = act_group<T>(*ptr, p.act);
        }
        rr += rstep; cc += cstep;
        if (cc >= gpr_x) { cc -= gpr_x; ++rr; }
      }
      __syncthreads();
    }

    // ---- K loop over (tap, channel) of this window, incremental decode per lane group
    const int kq = KSTEP / cw, krem = KSTEP - kq * cw;
    int tap = (fg * GK) / cw, c = fg * GK - tap * cw;
    const T* xb = Xs + ((wave * 2) * HW + fr) * q.ldc;
    const T* wb = Ws + fr * q.ldw + fg * GK;
    if (!(q.dbg & 8))
#pragma unroll 3
    for (int k0 = 0; k0 < kend; k0 += KSTEP) {
      const int tp = tap < TAPS ? tap : TAPS - 1;  // lanes past the last tap multiply zero weights; keep the address legal
      const int off = ((tp / KS) * HW + (tp % KS)) * q.ldc + c;
      if constexpr (SPLIT) {
        f16x8_t bh[2], bl[2], ah[NTC], al[NTC];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const char* bp = (const char*)(xb + f * HW * q.ldc + off);
          bh[f] = *(const f16x8_t*)bp; bl[f] = *(const f16x8_t*)(bp + 16);
        }
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
          const char* ap = (const char*)(wb + t * 16 * q.ldw + k0);
          ah[t] = *(const f16x8_t*)ap; al[t] = *(const f16x8_t*)(ap + 16);
        }
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh[f], acc[t][f], 0, 0, 0);
            accx[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl[f], accx[t][f], 0, 0, 0);
            accx[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh[f], accx[t][f], 0, 0, 0);
          }
      } else if constexpr (sizeof(T) == 4) {
        f32x4 bq[2], aq[NTC];
#pragma unroll
        for (int f = 0; f < 2; ++f) bq[f] = *(const f32x4*)(xb + f * HW * q.ldc + off);
#pragma unroll
        for (int t = 0; t < NTC; ++t) aq[t] = *(const f32x4*)(wb + t * 16 * q.ldw + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t][j], bq[f][j], acc[t][f], 0, 0, 0);
      } else {
        h16x8 bq[2], aq[NTC];
#pragma unroll
        for (int f = 0; f < 2; ++f) bq[f] = *(const h16x8*)(xb + f * HW * q.ldc + off);
#pragma unroll
        for (int t = 0; t < NTC; ++t) aq[t] = *(const h16x8*)(wb + t * 16 * q.ldw + k0);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
          for (int f = 0; f < 2; ++f) acc[t][f] = mfma_h16(aq[t], bq[f], acc[t][f], 0, 0, 0);
      }
      tap += kq; c += krem;
      if (c >= cw) { c -= cw; ++tap; }
    }
  }
  if (q.dbg & 16) return;
  if constexpr (SPLIT) {
#pragma unroll
    for (int t = 0; t < NTC; ++t)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][f][e] = fmaf(accx[t][f][e], 1.f / 2048.f, acc[t][f][e]);
  }
  if constexpr (sizeof(T) == 2) {
    // LDS-staged epilogue: the MFMA layout gives a lane 4 channels of one pixel (8-byte pieces, 32 B runs per pixel);
    // re-reading the wave's 32 pixels x COT channels from LDS as 16-byte chunks makes consecutive lanes cover consecutive
    // chunks of a pixel row, so the aux / residual loads and the output store are whole 128-byte+ runs.
    constexpr int COT = NTC * 16, LDE = COT + 4, CPP = COT / 8;  // chunks per pixel
    __syncthreads();  // everyone is done with the weight slab / halo tile
    float* es = (float*)smem + wave * (32 * LDE);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int t = 0; t < NTC; ++t) *(f32x4*)(es + (f * 16 + fr) * LDE + t * 16 + fg * 4) = acc[t][f];
    // wave-private region: LDS operations of one wave complete in order, no block barrier needed
#pragma unroll
    for (int k = 0; k < NTC; ++k) {
      const int idx = lane + 64 * k;
      const int pl = idx / CPP, ch = idx % CPP;
      const int py = y0 + wave * 2 + (pl >> 4), px = x0 + (pl & 15);
      if (py < p.H && px < p.W) {
        const f32x4 lo = *(const f32x4*)(es + pl * LDE + ch * 8), hi = *(const f32x4*)(es + pl * LDE + ch * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (efast[k]) epi8_finish<REM>(p, v, epl[k], ebias, n, py, px, co_base + ch * 8);
        else conv_epilogue8_h16<REM>(p, v, n, py, px, co_base + ch * 8);
      }
    }
  } else {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int py = y0 + wave * 2 + f, px = x0 + fr;
      if (py >= p.H || px >= p.W) continue;
#pragma unroll
      for (int t = 0; t < NTC; ++t) conv_epilogue<T, REM>(p, acc[t][f], n, py, px, co_base + t * 16 + fg * 4);
    }
  }
}

// LDS row strides: +8 elements (16 B bf16 / 32 B f32) keeps consecutive rows on distinct 16-byte bank slots
static inline int lds_stride(int width, int esz) {
  int s = width + 8;
  if (((s * esz / 16) & 1) == 0) s += 16 / esz;  // make the stride an odd number of 16-byte slots
  return s;
}

template <typename T, int KS>
static bool launch_conv_tile(const ConvP& p, hipStream_t st) {
  constexpr int HALO = KS / 2, HPX = (TILE_H + 2 * HALO) * (TILE_W + 2 * HALO), TAPS = KS * KS;
  constexpr int esz = sizeof(T), G = 16 / esz;
  const int KSTEP = (esz == 4 && !p.split) ? 16 : 32;
  TileP q;
  q.tiles_x = ceil_div(p.W, TILE_W); q.tiles_y = ceil_div(p.H, TILE_H);
  const int ntiles = p.N * q.tiles_y * q.tiles_x;
  int ntc = p.Co <= 16 ? 1 : (p.Co <= 32 ? 2 : 4);
  if (ntiles < 512 && ntc > 1) ntc = ntiles < 192 ? 1 : 2;  // few tiles -> spread output channels over more workgroups
  const int wrows = ntc * 16;
  auto lds_bytes = [&](int cw, int& kp, int& ldc, int& ldw) {
    kp = pad_to(TAPS * cw, KSTEP);
    ldc = lds_stride(cw, esz); ldw = lds_stride(kp, esz);
    const long wp = ((long)wrows * (ldw / G) + 63) / 64, xp = ((long)HPX * (ldc / G) + 63) / 64;
    return (wp + xp) * 1024;
  };
  // widest channel window whose weight slab + halo tile fit the LDS budget (budget keeps >= 2 workgroups per CU)
  const long budget = 64 * 1024;  // default dynamic-LDS limit; also keeps >= 2 workgroups per CU
  int cw = p.ctot8;
  long lds = 0;
  for (;; cw -= 8) {
    if (cw < 8) return false;
    lds = lds_bytes(cw, q.kp, q.ldc, q.ldw);
    if (lds <= budget) { q.cw = cw; break; }
  }
  { const long epi = 4L * 32 * (wrows + 4) * 4; if (lds < epi) lds = epi; }
  q.d_gprw = mk_fastdiv(q.ldw / G); q.d_gprx = mk_fastdiv(q.ldc / G); q.d_cw = mk_fastdiv(q.cw);
  { const char* e = getenv("CGEN_TILE_DBG"); q.dbg = e ? atoi(e) : 0; q.pad0 = 0; }
  dim3 block(256);
  const bool rem = sizeof(T) == 2 && (p.out_rem || p.r1_rem);  // (remainder planes: f16 only, checked by cgen_conv2d)
  if constexpr (sizeof(T) == 2) {
    if (rem) {
      if (ntc == 1) hipLaunchKernelGGL((conv_tile_kernel<T, 1, KS, true>), dim3(ntiles, ceil_div(p.Co, 16)), block, lds, st, p, q);
      else if (ntc == 2) hipLaunchKernelGGL((conv_tile_kernel<T, 2, KS, true>), dim3(ntiles, ceil_div(p.Co, 32)), block, lds, st, p, q);
      else hipLaunchKernelGGL((conv_tile_kernel<T, 4, KS, true>), dim3(ntiles, ceil_div(p.Co, 64)), block, lds, st, p, q);
      return true;
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (p.split) {
      if (ntc == 1) hipLaunchKernelGGL((conv_tile_kernel<T, 1, KS, false, true>), dim3(ntiles, ceil_div(p.Co, 16)), block, lds, st, p, q);
      else if (ntc == 2) hipLaunchKernelGGL((conv_tile_kernel<T, 2, KS, false, true>), dim3(ntiles, ceil_div(p.Co, 32)), block, lds, st, p, q);
      else hipLaunchKernelGGL((conv_tile_kernel<T, 4, KS, false, true>), dim3(ntiles, ceil_div(p.Co, 64)), block, lds, st, p, q);
      return true;
    }
  }
  if (ntc == 1) hipLaunchKernelGGL((conv_tile_kernel<T, 1, KS, false>), dim3(ntiles, ceil_div(p.Co, 16)), block, lds, st, p, q);
  else if (ntc == 2) hipLaunchKernelGGL((conv_tile_kernel<T, 2, KS, false>), dim3(ntiles, ceil_div(p.Co, 32)), block, lds, st, p, q);
  else hipLaunchKernelGGL((conv_tile_kernel<T, 4, KS, false>), dim3(ntiles, ceil_div(p.Co, 64)), block, lds, st, p, q);
  return true;
}

static bool launch_conv_ws(const ConvP& p, hipStream_t st);  // weight-stationary persistent kernel (bf16), defined below
static bool launch_conv_px(const ConvP& p, hipStream_t st);  // lean persistent kernel for short-K convs (bf16), defined below
#define PX_MAXKS 16

// ============================================================================= small-image conv with the K axis split (bf16)
// Layers with few pixels in the batch (1x1 ... 12x12 images at batch 32: the top of both hierarchies) have long K
// (512 channels x 9 taps = 144 K-steps) and almost no pixels: the generic kernel walks K serially in a handful of
// workgroups, one exposed global-memory round trip per step (measured 40-270 us per launch).  Here the K-steps are
// dealt round-robin to the 4 waves of a workgroup; each wave loads its MFMA operands straight from global memory (16 B per
// lane: a weight-image row slice and an im2col slice of one pixel) with four K-steps in flight, and the four partial sums
// are added through LDS in a fixed order (deterministic).  One workgroup = 16*SP_NCO output channels x 32 pixels.
template <int KS, int SP_NCO, int KU, bool ONESEG>
__device__ __forceinline__ void conv_smallp_body(const ConvP& p, const int nks, const FastDiv d_ctot8, const FastDiv d_hw, const FastDiv d_w, unsigned long long* stamps,
                                                 const int bx, const int by, const int gx) {
  CGEN_SETPRIO();
  unsigned long long* stamp = (stamps != nullptr && threadIdx.x == 0) ? stamps + 8 * (by * gx + bx) : nullptr;
  if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();  // the chain's waves win issue arbitration over background weight-gradient waves on the same SIMD
  typedef h16_t T;
  constexpr int HALO = KS / 2, TAPS = KS * KS;  // KU K-steps are issued together per wave
  __shared__ __attribute__((aligned(16))) float red[4 * SP_NCO * 2 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int co_base = by * (SP_NCO * 16);
  const int HWp = p.H * p.W;
  int pn[2], py[2], px[2];
  bool pv[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int m = bx * 32 + f * 16 + fr;
    pv[f] = m < p.P;
    const int mm = pv[f] ? m : p.P - 1;
    pn[f] = fdiv(mm, d_hw);
    const int r = mm - pn[f] * HWp;
    py[f] = fdiv(r, d_w);
    px[f] = r - py[f] * p.W;
  }
  // 32-bit element offsets (the host checked that every view spans < 2^31 bytes): this lane's two pixels in every segment,
  // and its weight rows (rows past the image are clamped: their outputs are never stored)
  int poff[CGEN_MAX_SEG][2];
#pragma unroll
  for (int u = 0; u < CGEN_MAX_SEG; ++u)
#pragma unroll
    for (int f = 0; f < 2; ++f)
      poff[u][f] = (ONESEG && u > 0) ? 0 : pn[f] * (int)p.seg[u].sn + py[f] * (int)p.seg[u].sh + px[f] * (int)p.seg[u].sw;
  const int rows_pad = (p.Co + 15) & ~15;
  int woff[SP_NCO];
#pragma unroll
  for (int t = 0; t < SP_NCO; ++t) woff[t] = __umul24(min(co_base + t * 16 + fr, rows_pad - 1), p.krow) + fg * 8;
  f32x4 acc[SP_NCO][2];
#pragma unroll
  for (int t = 0; t < SP_NCO; ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f) acc[t][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // K-step ks = k0 + wave + 4*j ; image column = tap0*ctot8 + ks*32 + fg*8 (1x1 images only ever see their centre tap).
  // The loop is instruction-issue bound (~150 instructions per K-step for 4 MFMAs before this form), so:
  //   * it is instantiated per activation (the dispatch on p.act happens once, outside);
  //   * operands come through BUFFER loads: the weight address is (per-lane row offset) + (scalar column offset) with no
  //     vector arithmetic at all, and an out-of-image / out-of-range activation group is a lane whose offset lies past
  //     num_records -- the hardware returns zeros, no pointer select, no branch (one-segment inputs; concatenated inputs
  //     pick their segment per lane and keep the pointer form).
  const int kcol0 = p.tap0 * p.ctot8;
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].p, 0, 0x7ffffff0, 0x00020000);
  if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
  auto kloop = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    for (int kb = wave; kb < nks; kb += 4 * KU) {
      u32x4 aq[KU][SP_NCO], bq[KU][2];
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        const int ks = min(kb + 4 * j, nks);  // K-steps past the end read the 32 zero columns that close every weight row
        const int kc = kcol0 + ks * 32;
#pragma unroll
        for (int t = 0; t < SP_NCO; ++t) aq[j][t] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[t] * 2, kc * 2, 0);
        const int k = kc + fg * 8;
        const int tap = fdiv(k, d_ctot8), c = k - tap * p.ctot8;
        const int dy = (KS == 3 ? (min(tap, 8) * 11) >> 5 : 0), dx = (KS == 3 ? min(tap, 8) - dy * 3 : 0);
        if (ONESEG) {
          const bool kv = tap < TAPS && c < p.seg[0].c;
          const int delta = __mul24(dy - HALO, (int)p.seg[0].sh) + __mul24(dx - HALO, (int)p.seg[0].sw) + c;  // strides < 2^24 (dma_clean)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const int yy = py[f] + dy - HALO, xx = px[f] + dx - HALO;
            const bool ok = kv && pv[f] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            bq[j][f] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? (poff[0][f] + delta) * 2 : 0x7fffffff, 0, 0);
          }
        } else {
          const T* sbase = (const T*)p.seg[0].p;
          int cs = c, segc = p.seg[0].c, sh = (int)p.seg[0].sh, sw = (int)p.seg[0].sw, o0 = poff[0][0], o1 = poff[0][1];
#pragma unroll
          for (int u = 1; u < CGEN_MAX_SEG; ++u)
            if (u < p.nseg && c >= p.seg_koff[u]) {
              sbase = (const T*)p.seg[u].p; cs = c - p.seg_koff[u]; segc = p.seg[u].c; sh = (int)p.seg[u].sh; sw = (int)p.seg[u].sw;
              o0 = poff[u][0]; o1 = poff[u][1];
            }
          const bool kv = tap < TAPS && cs < segc;
          const int delta = __mul24(dy - HALO, sh) + __mul24(dx - HALO, sw) + cs;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const int yy = py[f] + dy - HALO, xx = px[f] + dx - HALO;
            const bool ok = kv && pv[f] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            const u32x4* src = ok ? (const u32x4*)(sbase + ((f == 0 ? o0 : o1) + delta)) : (const u32x4*)g_zero16;
            bq[j][f] = *src;  // unconditional load from a selected address (no early wait)
          }
        }
      }
#pragma unroll
      for (int j = 0; j < KU; ++j) {
        union { u32x4 u; uint4 q; h16x8 v; } a0, b0;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          b0.u = bq[j][f];
          if (ACT != CGEN_ACT_NONE) b0.q = act_group<T>(b0.q, ACT);
#pragma unroll
          for (int t = 0; t < SP_NCO; ++t) {
            a0.u = aq[j][t];
            acc[t][f] = mfma_h16(a0.v, b0.v, acc[t][f], 0, 0, 0);
          }
        }
      }
    }
  };
  if (p.act == CGEN_ACT_NONE) kloop(std::integral_constant<int, CGEN_ACT_NONE>{});
  else if (p.act == CGEN_ACT_RELU) kloop(std::integral_constant<int, CGEN_ACT_RELU>{});
  else kloop(std::integral_constant<int, CGEN_ACT_GELU>{});
  if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
  // ---- fixed-order reduction over the four waves, then the generic fused epilogue
#pragma unroll
  for (int t = 0; t < SP_NCO; ++t)
#pragma unroll
    for (int f = 0; f < 2; ++f) *(f32x4*)(red + ((wave * SP_NCO + t) * 2 + f) * 256 + lane * 4) = acc[t][f];
  __syncthreads();
  if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
  // the SP_NCO * 2 fragments are finalised round-robin by the four waves
  for (int fi = wave; fi < SP_NCO * 2; fi += 4) {
    const int t = fi >> 1, f = fi & 1;
    f32x4 v = *(const f32x4*)(red + ((0 * SP_NCO + t) * 2 + f) * 256 + lane * 4);
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const f32x4 o = *(const f32x4*)(red + ((w2 * SP_NCO + t) * 2 + f) * 256 + lane * 4);
      v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    // the pixel of fragment f owned by this lane (all lanes hold identical pn/py/px tables for both f)
    const int n2 = f == 0 ? pn[0] : pn[1], y2 = f == 0 ? py[0] : py[1], x2 = f == 0 ? px[0] : px[1];
    const bool v2 = f == 0 ? pv[0] : pv[1];
    if (v2) conv_epilogue<T>(p, v, n2, y2, x2, co_base + t * 16 + fg * 4);
  }
  if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
}

template <int KS, int SP_NCO, int KU, bool ONESEG>
__global__ __launch_bounds__(256) void conv_smallp_kernel(ConvP p, int nks, FastDiv d_ctot8, FastDiv d_hw, FastDiv d_w, unsigned long long* stamps) {
  conv_smallp_body<KS, SP_NCO, KU, ONESEG>(p, nks, d_ctot8, d_hw, d_w, stamps, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// Two independent small-image convs of the same instance in ONE launch (cgen_conv2d_pair): the data gradients of the posterior and the
// prior Block's convs on the <= 12x12 layers, whose launches are latency-bound whatever their size (see launch_conv_smallp) -- two of
// them side by side cost what one costs.  Workgroups with blockIdx.x < a.gx take the first problem; the record is picked by address
// inside the kernarg segment (scalar loads), rows of the grid a problem does not need leave at once.
struct SpArgs { ConvP p; int nks, gx, gy, pad; FastDiv d_ctot8, d_hw, d_w; };
template <int KS, int SP_NCO, int KU, bool ONESEG>
__global__ __launch_bounds__(256) void conv_smallp_pair_kernel(SpArgs a, SpArgs b) {
  (void)a; (void)b;
  typedef const SpArgs __attribute__((address_space(4)))* ka_ptr;
  const int gxa = ((ka_ptr)__builtin_amdgcn_kernarg_segment_ptr())->gx;
  const bool second = (int)blockIdx.x >= gxa;
  constexpr size_t off_b = (sizeof(SpArgs) + alignof(SpArgs) - 1) / alignof(SpArgs) * alignof(SpArgs);
  const SpArgs& s = *(const SpArgs*)(ka_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + (second ? off_b : 0));
  if ((int)blockIdx.y >= s.gy) return;
  conv_smallp_body<KS, SP_NCO, KU, ONESEG>(s.p, s.nks, s.d_ctot8, s.d_hw, s.d_w, nullptr, second ? (int)blockIdx.x - gxa : (int)blockIdx.x, (int)blockIdx.y, s.gx);
}

// every element offset of the view (extent n x h x w) fits 32-bit byte arithmetic
static inline bool fits_i32(const View& v, int n, int h, int w) {
  if (!v.p) return true;
  const int64_t ext = (int64_t)n * v.sn + (int64_t)h * v.sh + (int64_t)w * v.sw + v.c;
  return ext * 2 < ((int64_t)1 << 31);
}

// does this f16 conv go to conv_smallp_kernel?  THE decision: launch_conv and the pair launch (cgen_conv2d_pair) both ask here, so a
// paired and an unpaired backward pass cannot pick different kernels
static bool smallp_takes(const ConvP& p) {
  static const int smallp_maxp = [] { const char* e = getenv("CGEN_SMALLP_MAXP"); return e ? atoi(e) : 6000; }();
  static const int smallp_maxp_longk = [] { const char* e = getenv("CGEN_SMALLP_MAXP_LONGK"); return e ? atoi(e) : 19000; }();
  static const int smallp_longk = [] { const char* e = getenv("CGEN_SMALLP_LONGK"); return e ? atoi(e) : 60; }();
  const bool longk = ceil_div(p.taps * p.ctot8, 32) >= smallp_longk;
  const bool by_size = (p.P <= smallp_maxp || (longk && p.P <= smallp_maxp_longk)) && (p.KS == 1 || p.KS == 3);
  static const bool no_smallp = getenv("CGEN_CONV_NO_SMALLP") != nullptr;
  const bool by_side = (p.H < 5 || p.W < 5) && (p.KS == 1 || p.KS == 3) && !no_smallp;
  if (p.force_generic || !(by_size || by_side)) return false;
  for (int s = 0; s < p.nseg; ++s)
    if (!p.seg_vec[s] || !fits_i32(p.seg[s], p.N, p.H + 2, p.W + 2)) return false;
  return p.dma_ok != 0;
}

static bool launch_conv_smallp(const ConvP& p, hipStream_t st) {
  for (int s = 0; s < p.nseg; ++s)
    if (!p.seg_vec[s]) return false;  // 16-byte fragment loads
  if (!p.dma_ok) return false;        // ragged channel counts must be zero padded to 8 (cpad), as for the tiled kernels
  const int ntap = p.tap1 - p.tap0;
  const int nks = ceil_div(ntap * p.ctot8, 32);
  // measured on MI355X: 1 / 2 output fragments and 2 / 4 K-steps in flight are equivalent, 6 / 8 are slower.  The K loop's time
  // does not depend on the number of workgroups (4.2 us for 12 K-steps per wave from batch 4 to 32) nor on its instruction
  // count (150 -> 60 per K-step changed nothing): it is ~22 ns per 1-KiB load instruction per CU -- every load touches 16
  // half-used cache lines -- plus ~0.6 us latency per round of KU K-steps
  constexpr int NCO = 2, KU = 4;
  dim3 grid(ceil_div(p.P, 32), ceil_div(p.Co, NCO * 16));
  const FastDiv d1 = mk_fastdiv(p.ctot8), d2 = mk_fastdiv(p.H * p.W), d3 = mk_fastdiv(p.W);
  // 32-bit address arithmetic in the kernel
  for (int sg = 0; sg < p.nseg; ++sg)
    if (!fits_i32(p.seg[sg], p.N, p.H + 2, p.W + 2)) return false;
  unsigned long long* stamps = nullptr;
  { const char* e = getenv("CGEN_PX_STAMPS"); stamps = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
#define SP_LAUNCH(KS_, KU_) do { if (p.nseg == 1) hipLaunchKernelGGL((conv_smallp_kernel<KS_, NCO, KU_, true>), grid, dim3(256), 0, st, p, nks, d1, d2, d3, stamps); \
    else hipLaunchKernelGGL((conv_smallp_kernel<KS_, NCO, KU_, false>), grid, dim3(256), 0, st, p, nks, d1, d2, d3, stamps); } while (0)
  if (p.KS == 1) SP_LAUNCH(1, KU);
  else if (p.KS == 3) SP_LAUNCH(3, KU);
  else return false;
#undef SP_LAUNCH
  return true;
}

static void conv_trace(const ConvP& p, const char* which) {  // CGEN_CONV_TRACE=1: which kernel served which shape
  static const bool on = getenv("CGEN_CONV_TRACE") != nullptr;
  if (on) fprintf(stderr, "conv %-5s ks%d ci8 %-4d co %-4d res %-3d nseg %d act %d aux %d res %d%d\n", which, p.KS, p.ctot8, p.Co, p.H, p.nseg, p.act,
                  p.aux.p != nullptr, p.res1.p != nullptr, p.res2.p != nullptr);
}

template <typename T>
static int launch_conv(const ConvP& p, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (smallp_takes(p) && launch_conv_smallp(p, st)) {  // (few pixels in the batch, or an image side below 5: by size or by side)
      conv_trace(p, "smlp");
      return check_launch("cgen_conv2d(smallp)");
    }
  }
  if constexpr (sizeof(T) == 2) {
    if ((p.KS == 1 || p.KS == 3) && p.H >= 5 && p.W >= 5 && p.dma_ok && !p.force_generic && !getenv("CGEN_CONV_NO_PX")) {
      const int nks = ceil_div(p.taps * p.ctot8, 32);
      const bool expanding = nks <= PX_MAXKS && p.Co >= 2 * p.ctot8;  // short K, wide output
      static const int px_mode = [] { const char* e = getenv("CGEN_PX_MODE"); return e ? atoi(e) : 2; }();  // 0: expanding only, 1: +1x1, 2: everything with a short K
      const bool take = expanding || (px_mode >= 1 && p.KS == 1 && nks <= PX_MAXKS) || (px_mode >= 2 && nks <= PX_MAXKS);
      if (take && launch_conv_px(p, st)) { conv_trace(p, "px"); return check_launch("cgen_conv2d(px)"); }
    }
    if ((p.KS == 1 || p.KS == 3) && p.H >= 5 && p.W >= 5 && p.dma_ok && !p.force_generic && !getenv("CGEN_CONV_NO_WS")) {
      if (!p.out_rem && !p.r1_rem && launch_conv_ws(p, st)) { conv_trace(p, "ws"); return check_launch("cgen_conv2d(ws)"); }
    }
  }
  if ((p.KS == 1 || p.KS == 3) && p.H >= 5 && p.W >= 5 && !p.force_generic && p.dma_ok) {
    const bool ok = p.KS == 3 ? launch_conv_tile<T, 3>(p, st) : launch_conv_tile<T, 1>(p, st);
    if (ok) { conv_trace(p, "tile"); return check_launch("cgen_conv2d(tile)"); }
  }
  dim3 block(256);
  const int px_tiles = ceil_div(p.P, CONV_PT);
  if (p.Co <= 16) {
    hipLaunchKernelGGL((conv_kernel<T, 1>), dim3(px_tiles, 1), block, 0, st, p);
  } else if (p.Co <= 32) {
    hipLaunchKernelGGL((conv_kernel<T, 2>), dim3(px_tiles, 1), block, 0, st, p);
  } else {
    hipLaunchKernelGGL((conv_kernel<T, 4>), dim3(px_tiles, ceil_div(p.Co, 64)), block, 0, st, p);
  }
  conv_trace(p, "gen");
  return check_launch("cgen_conv2d");
}

// ============================================================================= weight gradient
// dW[co][tap][ci] = sum_px G[px][co] * act(X)[px + tap][ci]          (f32 MFMA 16x16x4 for both storage dtypes)
// Workgroup tile: (16*NTC output channels) x (32 input channels) x (<= 9 taps) x one pixel split.
#define WG_PK 64
#define WG_MAXT 9

struct WgP {
  int N, H, W, KS, pad, nseg, act, Co, P, taps, ci_total, nsplit, pix_per_split, n_tapgroups, n_cichunks;
  View seg[CGEN_MAX_SEG];
  int seg_off[CGEN_MAX_SEG];
  int seg_vec[CGEN_MAX_SEG];
  int seg_chunk0[CGEN_MAX_SEG + 1];  // first 32-wide chunk index of each segment
  View gout;
  int gout_vec;
  float* pw;
  float* pb;
};

template <typename T, int NTC>
__global__ __launch_bounds__(256) void wgrad_kernel(WgP p) {
  constexpr int COT = NTC * 16;
  constexpr int LDG = COT + 16;  // stride == 16 (mod 32) -> conflict-free b32 fragment reads
  constexpr int LDX = 32 + 16;
  constexpr int NPAIR = NTC * 2;                 // (co-frag, ci-frag) pairs
  constexpr int KW = NPAIR >= 4 ? 1 : 4 / NPAIR; // K-ways across waves
  constexpr int PPW = NPAIR >= 4 ? NPAIR / 4 : 1;
  constexpr int G = 16 / sizeof(T);
  __shared__ float Gs[WG_PK * LDG];
  __shared__ float Xs[WG_PK * LDX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sp = blockIdx.x;
  const int chunk = blockIdx.y;  // global 32-wide ci chunk
  const int co_tile = blockIdx.z / p.n_tapgroups, tg = blockIdx.z % p.n_tapgroups;
  const int co_base = co_tile * COT;
  const int t0 = tg * WG_MAXT;
  const int ntap = min(WG_MAXT, p.taps - t0);
  int s = 0;
  while (s + 1 < p.nseg && chunk >= p.seg_chunk0[s + 1]) ++s;
  const int c0 = (chunk - p.seg_chunk0[s]) * 32;
  const View sv = p.seg[s];

  const int px0 = sp * p.pix_per_split;
  const int px1 = min(p.P, px0 + p.pix_per_split);

  f32x4 acc[WG_MAXT][PPW];
#pragma unroll
  for (int t = 0; t < WG_MAXT; ++t)
#pragma unroll
    for (int q = 0; q < PPW; ++q) acc[t][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const int kpart = (KW > 1) ? (wave / NPAIR) : 0;
  const int pair0 = (KW > 1) ? (wave % NPAIR) : wave * PPW;

  for (int pc = px0; pc < px1; pc += WG_PK) {
    __syncthreads();
    // ---- gradient tile Gs[64][COT] (f32)
    for (int g = tid; g < WG_PK * COT / 4; g += 256) {
      const int r = g / (COT / 4), cg = (g % (COT / 4)) * 4;
      const int m = pc + r, co = co_base + cg;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (m < px1 && co < p.Co) {
        const int n = m / (p.H * p.W);
        const int rr = m - n * p.H * p.W;
        const int y = rr / p.W, x = rr - y * p.W;
        const T* src = vptr<T>(p.gout, n, y, x) + co;
        if (p.gout_vec && co + 4 <= p.Co) {
          ld4<T>(src, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (co + e < p.Co) v[e] = Elem<T>::ld(src + e);
        }
      }
      *(float4*)(Gs + r * LDG + cg) = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int tt = 0; tt < WG_MAXT; ++tt) {
      if (tt < ntap) {
        const int tap = t0 + tt;
        const int dy = tap / p.KS - p.pad, dx = tap % p.KS - p.pad;
        if (tt > 0) __syncthreads();
        // ---- activation tile Xs[64][32] for this tap
        for (int g = tid; g < WG_PK * 32 / G; g += 256) {
          const int r = g / (32 / G), cg = (g % (32 / G)) * G;
          const int m = pc + r, cs = c0 + cg;
          float v[G];
#pragma unroll
          for (int e = 0; e < G; ++e) v[e] = 0.f;
          if (m < px1 && cs < sv.c) {
            const int n = m / (p.H * p.W);
            const int rr = m - n * p.H * p.W;
            const int y = rr / p.W + dy, x = rr % p.W + dx;
            if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
              const T* src = vptr<T>(sv, n, y, x) + cs;
              if (p.seg_vec[s] && cs + G <= sv.c) {
                Pack<T, G> tmp;
                tmp.v4 = *(const uint4*)src;
#pragma unroll
                for (int e = 0; e < G; ++e) v[e] = act_fwd(p.act, Elem<T>::ld(&tmp.e[e]));
              } else {
#pragma unroll
                for (int e = 0; e < G; ++e) if (cs + e < sv.c) v[e] = act_fwd(p.act, Elem<T>::ld(src + e));
              }
            }
          }
#pragma unroll
          for (int e = 0; e < G; e += 4) *(float4*)(Xs + r * LDX + cg + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
        }
        __syncthreads();
        if (tt == 0 && p.pb && chunk == 0 && tg == 0 && tid < COT) {
          float a = 0.f;
          for (int r = 0; r < WG_PK; ++r) a += Gs[r * LDG + tid];
          bsum += a;
        }
        // ---- MFMA: D[co][ci] += G^T[co][px] * X[px][ci]
        const int kq0 = kpart * (WG_PK / 4 / KW), kq1 = kq0 + WG_PK / 4 / KW;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
          const int pr = pair0 + q;
          const int cf = pr >> 1, jf = pr & 1;
          for (int k4 = kq0; k4 < kq1; ++k4) {
            const int prow = k4 * 4 + (lane >> 4);
            const float a = Gs[prow * LDG + cf * 16 + (lane & 15)];
            const float b = Xs[prow * LDX + jf * 16 + (lane & 15)];
            acc[tt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tt][q], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- write partials (K-ways are combined through LDS in a fixed order)
  __syncthreads();
  float* red = Gs;  // reuse (>= 64*32 floats)
#pragma unroll
  for (int tt = 0; tt < WG_MAXT; ++tt) {
    if (tt < ntap) {
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        f32x4 v = acc[tt][q];
        if (KW > 1) {
          // waves with kpart > 0 publish, kpart == 0 sums in kpart order
          for (int kp = 1; kp < KW; ++kp) {
            __syncthreads();
            if (kpart == kp) *(f32x4*)(red + ((wave % NPAIR) * 64 + lane) * 4) = v;
            __syncthreads();
            if (kpart == 0) {
              const f32x4 o = *(const f32x4*)(red + ((wave % NPAIR) * 64 + lane) * 4);
              v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
          }
        }
        if (kpart == 0) {
          const int pr = pair0 + q;
          const int cf = pr >> 1, jf = pr & 1;
          const int ci = c0 + jf * 16 + (lane & 15);
          if (ci < sv.c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int co = co_base + cf * 16 + (lane >> 4) * 4 + e;
              if (co < p.Co)
                p.pw[(((size_t)sp * p.Co + co) * p.taps + (t0 + tt)) * p.ci_total + p.seg_off[s] + ci] = v[e];
            }
          }
        }
      }
    }
  }
  if (p.pb && chunk == 0 && tg == 0 && tid < COT && co_base + tid < p.Co) p.pb[(size_t)sp * p.Co + co_base + tid] = bsum;
}

// ============================================================================= weight gradient, tiled / streaming (bf16)
// dW[co][tap][ci] = sum_px G[px][co] * act(X)[px + tap][ci] with the PIXEL axis as the MFMA K dimension.
// Both operands live in HBM with channels contiguous, so K (pixels) is strided: the fragments are assembled with the
// gfx950 LDS transpose read ds_read_b64_tr_b16 (4 pixels x 16 channels -> each lane gets 4 consecutive-K values of its
// channel) straight from NHWC tiles in LDS -- no explicit transposes.
// A persistent workgroup owns the whole [16*NCF co] x [taps x cwin ci] gradient slab in MFMA accumulators
// (output-stationary) and streams 8x16 pixel tiles through LDS by global->LDS DMA: every activation / gradient element
// is read from HBM exactly once per (co range, ci window).  2 workgroups per CU: one's DMA wait overlaps the other's MFMAs.
//
// LDS pixel-tile layout ("row pieces"): a tile row is cut into 1-KiB pieces of `ppp` pixels x `gpr` 16-byte groups
// (gpr odd => conflict-free fragment reads).  One wave-wide DMA instruction fills one piece, so everything about a piece
// except a per-lane constant is wave-uniform: the per-tile address generation is a few SALU ops + ~6 VALU per piece.
#define WG2_MAXACC 24  // accumulator fragments per wave (96 VGPRs)

struct PixTile {
  int gpr, ppp, ppr, rows, npx, rowbytes, bytes, pad;
  FastDiv d_gpr, d_ppr, d_ppp;
};
static inline PixTile mk_pixtile(int width_elems, int esz, int rows, int npx) {
  PixTile t;
  int gpr = (width_elems * esz + 15) / 16 + 1;
  if ((gpr & 1) == 0) ++gpr;
  t.gpr = gpr; t.ppp = 64 / gpr; t.rows = rows; t.npx = npx;
  if (t.ppp < 1) {  // pixel wider than one 1-KiB piece: caller must fall back (checks ppp < 1)
    t.ppr = 0; t.rowbytes = 0; t.bytes = 0; t.pad = 0;
    t.d_gpr = mk_fastdiv(1); t.d_ppr = mk_fastdiv(1); t.d_ppp = mk_fastdiv(1);
    return t;
  }
  t.ppr = (npx + t.ppp - 1) / t.ppp;
  t.rowbytes = t.ppr * 1024; t.bytes = rows * t.rowbytes; t.pad = 0;
  t.d_gpr = mk_fastdiv(gpr); t.d_ppr = mk_fastdiv(t.ppr); t.d_ppp = mk_fastdiv(t.ppp);
  return t;
}
// byte offset of pixel (row, x) inside a PixTile
__device__ __forceinline__ int pix_off(const PixTile& t, int row, int x) {
  const int pc = fdiv(x, t.d_ppp);
  return row * t.rowbytes + pc * 1024 + (x - pc * t.ppp) * t.gpr * 16;
}

// ---- lean row-piece DMA of one tile (shared by the persistent kernels).  Everything tile-invariant about a lane lives in
// LaneTile (built once per kernel); per piece the wave spends ~10 VALU + ~12 SALU: the piece walk (hy, pc) is scalar and
// incremental, offsets are 24-bit multiplies, the image box is given in TILE coordinates.
struct LaneTile {
  int xl;       // pixel of this lane inside a piece
  int sh, swp;  // element strides of this lane's source per tile row / per piece (ppp pixels); < 2^24 (dma_clean)
  bool lane;    // lane maps to a slot of the piece
  bool data;    // ... and the slot carries real channels
};
// `org`: this lane's source for (tile row 0, piece 0), lane pixel / channel offset included.  Rows [ry0, ry1) and columns
// [cx0, cx1) of the tile lie inside the image (and inside the tile's own pixel extent); everything else reads zeros.
// `wave` must be wave-uniform (readfirstlane).  No LDS store may sit between two DMAs (hipcc would drain vmcnt).
template <typename T>
__device__ __forceinline__ void dma_tile(const PixTile& xt, const LaneTile& L, const T* org, char* buf, int wave, int ry0, int ry1,
                                         int cx0, int cx1) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  const int npieces = xt.rows * xt.ppr;
  int hy = fdiv(wave, xt.d_ppr), pc = wave - hy * xt.ppr;
  for (int pi = wave; pi < npieces; pi += 4) {
    if (L.lane) {
      const int hx = pc * xt.ppp + L.xl;
      const bool ok = L.data && hy >= ry0 && hy < ry1 && hx >= cx0 && hx < cx1;
      const T* src = ok ? org + (int)(__umul24(hy, L.sh) + __umul24(pc, L.swp)) : (const T*)g_zero16;
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(buf + pi * 1024), 16, 0, 0);
    }
    pc += 4;
    while (pc >= xt.ppr) { pc -= xt.ppr; ++hy; }
  }
}

struct Wg2P {
  int N, H, W, KS, nseg, act, Co, taps, ci_total, ctot8;
  View seg[CGEN_MAX_SEG];
  int seg_koff[CGEN_MAX_SEG];  // 8-granular concat offsets
  int seg_off[CGEN_MAX_SEG];   // real channel offsets in the OIHW gradient
  View gout;
  float* pw;
  float* pb;
  int tiles_x, tiles_y, ntiles, nsplit, tiles_per_split;
  int cwin, cog;  // channel window, co columns staged per workgroup (16*NCF)
  int dbg;        // ablation mask (CGEN_WG2_DBG): 1 skip DMA, 2 skip activation pass, 4 skip MFMA loop
  unsigned long long* stamps;  // optional (CGEN_WG2_STAMPS): per-phase cycle stamps of workgroup 0
  PixTile xt, gt;
  FastDiv d_tx, d_ty;
  int variant, dbuf;  // ncf * 2 + (KS == 3): which body the all-variant kernel runs for this problem; dbuf: two tile buffers (the next tile's DMA flies under this tile's MFMAs)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h16x8 tr_pair(const char* p0, const char* p1, const bool plain = false) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  if (plain) {  // ablation (CGEN_WG2_DBG & 8): ordinary 8-byte LDS reads instead of the transposing ones (wrong math, same traffic)
    union { s16x4 h[2]; h16x8 v; } u;
    u.h[0] = *(const s16x4*)p0; u.h[1] = *(const s16x4*)p1;
    return u.v;
  }
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p1);
  union { s16x4 h[2]; h16x8 v; } u;
  u.h[0] = lo; u.h[1] = hi;
  return u.v;
}

template <int NCF, int NJW, int KS>
__device__ __forceinline__ void wgrad_tile_body(const Wg2P& p, const int bid_x, const int bid_y, const int bid_z) {
  typedef h16_t T;
  constexpr int G = 8;
  constexpr int HALO = KS / 2, HH = TILE_H + 2 * HALO, HW = TILE_W + 2 * HALO;
  constexpr int TAPS = KS * KS;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bufbytes = p.xt.bytes + p.gt.bytes;  // one tile buffer: activation halo tile, then the gradient tile
  const bool dbuf = p.dbuf != 0;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sp = bid_x;
  const int cA = bid_y * p.cwin;
  const int cw = min(p.cwin, p.ctot8 - cA);      // multiple of 8
  const int cw16 = (cw + 15) & ~15;
  const int cgrp = cw16 >> 4;
  const int njf = TAPS * cgrp;                   // (tap, 16-channel group) fragments of this window
  const int co_base = bid_z * (NCF * 16);
  const int t_begin = sp * p.tiles_per_split, t_end = min(p.ntiles, t_begin + p.tiles_per_split);

  f32x4 acc[NCF][NJW];
  f32x4 accb[NCF];
#pragma unroll
  for (int a = 0; a < NCF; ++a) {
    accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJW; ++j) acc[a][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.pb != nullptr && bid_y == 0 && wave == 0;
  const unsigned long long t_entry = __builtin_readcyclecounter();

  // ---- per-lane DMA constants (identical for every piece of every tile)
  // activation tile: lane -> (pixel xl inside the piece, channel group) -> segment / channel / element offset
  const int xl = fdiv(lane, p.xt.d_gpr), xcg = lane - xl * p.xt.gpr;
  int x_si = 0, x_off = 0;
  bool x_lane = xl < p.xt.ppp && xcg * G < cw16, x_data = false;
  {
    const int c = cA + xcg * G;
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k) x_si += (k < p.nseg && c >= p.seg_koff[k]) ? 1 : 0;
    View sv = p.seg[0];
    int koff = p.seg_koff[0];
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k)
      if (x_si == k) { sv = p.seg[k]; koff = p.seg_koff[k]; }
    const int cs = c - koff;
    x_data = x_lane && xcg * G < cw && cs < sv.c;
    x_off = (int)(xl * sv.sw) + cs;
  }
  // gradient tile
  const int gl = fdiv(lane, p.gt.d_gpr), gcg = lane - gl * p.gt.gpr;
  const bool g_lane = gl < p.gt.ppp && gcg * G < p.cog;
  const bool g_data = g_lane && co_base + gcg * G < p.Co;
  const int g_off = (int)(gl * p.gout.sw) + co_base + gcg * G;
  const int xpieces = p.xt.rows * p.xt.ppr, gpieces = p.gt.rows * p.gt.ppr;

  LaneTile LX, LG;
  {
    LX.xl = xl; LX.lane = x_lane; LX.data = x_data;
    LX.sh = (int)p.seg[0].sh; LX.swp = (int)(p.seg[0].sw * p.xt.ppp);
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k)
      if (x_si == k) { LX.sh = (int)p.seg[k].sh; LX.swp = (int)(p.seg[k].sw * p.xt.ppp); }
    LG.xl = gl; LG.lane = g_lane; LG.data = g_data;
    LG.sh = (int)p.gout.sh; LG.swp = (int)(p.gout.sw * p.gt.ppp);
  }
  auto issue_tile = [&](int t, char* Xb, char* Gb) {
    const int b1 = fdiv(t, p.d_tx), tx = t - b1 * p.tiles_x;
    const int n = fdiv(b1, p.d_ty), ty = b1 - n * p.tiles_y;
    const int y0 = ty * TILE_H, x0 = tx * TILE_W;
    const T* my_org = vptr32<T>(p.seg[0], n, y0 - HALO, x0 - HALO);  // this lane's segment
    if (p.nseg > 1) {
      if (x_si == 1) my_org = vptr32<T>(p.seg[1], n, y0 - HALO, x0 - HALO);
      if (x_si == 2) my_org = vptr32<T>(p.seg[2], n, y0 - HALO, x0 - HALO);
      if (x_si == 3) my_org = vptr32<T>(p.seg[3], n, y0 - HALO, x0 - HALO);
    }
    dma_tile<T>(p.xt, LX, my_org + x_off, Xb, wave, max(0, HALO - y0), min(HH, p.H + HALO - y0), max(0, HALO - x0), min(HW, p.W + HALO - x0));
    dma_tile<T>(p.gt, LG, vptr32<T>(p.gout, n, y0, x0) + g_off, Gb, wave, 0, min(TILE_H, p.H - y0), 0, min(TILE_W, p.W - x0));
  };
  // in-place activation of the staged halo tile: every lane re-visits the groups it DMA'd (same piece mapping)
  auto act_pass = [&](char* Xb) { act_pieces<T>(Xb, wave, lane, xpieces, p.act); };

  // lane geometry of the transpose reads: group g = k-block, t = 4r + q supplies (pixel r of the read, channels 4q..4q+3)
  const int g = lane >> 4, t16 = lane & 15, r = t16 >> 2, qd = t16 & 3;
  const int krow = g >> 1, kx = (g & 1) * 8 + r;  // this lane's FIRST read inside a 2-row K-step: pixel (row, x); second: x + 4
  // tile-independent LDS byte offsets of this lane's two reads per (tap, channel group) fragment
  int xo0[NJW], xo1[NJW];
  int jtap[NJW], jcb[NJW];  // (tap, first channel of the 16-channel group) of fragment j of this wave; wave-uniform
  {
    int tapw = wave / cgrp, gw = wave - tapw * cgrp;  // fragment jf = wave + 4j  ->  (tap, group), walked incrementally
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      const int tap = tapw < TAPS ? tapw : TAPS - 1;  // waves short of a fragment compute a duplicate that is never written out
      const int cb = tapw < TAPS ? gw * 16 : 0;
      jtap[j] = tap; jcb[j] = cb;
      const int dy = tap / KS, dx = tap % KS;
      xo0[j] = pix_off(p.xt, krow + dy, kx + dx) + (cb + qd * 4) * 2;
      xo1[j] = pix_off(p.xt, krow + dy, kx + dx + 4) + (cb + qd * 4) * 2;
      gw += 4;
      while (gw >= cgrp) { gw -= cgrp; ++tapw; }
    }
  }
  const int nj_eff = (njf + 3) >> 2;  // fragments per wave, the same for all four waves
  const int go0 = pix_off(p.gt, krow, kx) + qd * 8, go1 = pix_off(p.gt, krow, kx + 4) + qd * 8;
  h16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (h16n_t)1.0f;

  const bool stamp = p.stamps != nullptr && bid_x == 0 && bid_y == 0 && bid_z == 0 && tid == 0;
#ifdef CGEN_WG2_ABLATE
  const bool plain_rd = (p.dbg & 8) != 0, no_mfma = (p.dbg & 16) != 0;
#else
  constexpr bool plain_rd = false;
#endif
  int nst = 0;
#define WG2_STAMP() do { if (stamp && nst < 60) p.stamps[nst++] = __builtin_readcyclecounter(); } while (0)
  if (stamp) p.stamps[nst++] = t_entry;
  WG2_STAMP();
  // One tile: activation pass + MFMAs on the buffer `cur` while (double-buffered problems) the DMA of tile t + 1 fills `nxt`.
  // Two things keep that DMA in flight: (1) `cur` / `nxt` are __restrict__ -- hipcc tracks an LDS-DMA as a pending LDS write
  // and, without alias information, puts s_waitcnt vmcnt(0) in front of the next LDS read; (2) the barrier between the
  // activation pass and the MFMAs is a bare lgkmcnt(0) + s_barrier: __syncthreads() drains vmcnt as well.
  auto tile_step = [&](char* __restrict__ cur, char* __restrict__ nxt, const int t) {
    if (dbuf && t + 1 < t_end && !(p.dbg & 1)) issue_tile(t + 1, nxt, nxt + p.xt.bytes);
    WG2_STAMP();
    char* Xb = cur;
    char* Gb = cur + p.xt.bytes;
    if (p.act != CGEN_ACT_NONE && !(p.dbg & 2)) {
      act_pass(Xb);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    WG2_STAMP();
    if (!(p.dbg & 4)) {
      // fragments are processed in chunks of CH: all transpose reads of a chunk are issued before its MFMAs (one LDS
      // latency per chunk instead of per fragment); the chunk test is wave-uniform and identical for the four waves
      constexpr int CH = (NJW % 4 == 0) ? 4 : 3;
#pragma unroll
      for (int ks = 0; ks < TILE_H / 2; ++ks) {
        h16x8 af[NCF];
        const char* gk = Gb + ks * 2 * p.gt.rowbytes;
#pragma unroll
        for (int a = 0; a < NCF; ++a) af[a] = tr_pair(gk + go0 + a * 32, gk + go1 + a * 32, plain_rd);
        if (do_bias) {
#pragma unroll
          for (int a = 0; a < NCF; ++a) accb[a] = mfma_h16(af[a], ones, accb[a], 0, 0, 0);
        }
        const char* xk = Xb + ks * 2 * p.xt.rowbytes;
#pragma unroll
        for (int j0 = 0; j0 < NJW; j0 += CH) {
          if (j0 < nj_eff) {
            h16x8 bfv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) bfv[u] = tr_pair(xk + xo0[j0 + u], xk + xo1[j0 + u], plain_rd);
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
              for (int a = 0; a < NCF; ++a) {
#ifdef CGEN_WG2_ABLATE  // (tools/coexec_probe.py, CGEN_WG2_DBG & 16): keep the LDS reads alive without the matrix instruction
                if (no_mfma) {
                  union { h16x8 v; float f[4]; } ua, ub;
                  ua.v = af[a]; ub.v = bfv[u];
                  acc[a][j0 + u][0] += ua.f[0] + ub.f[0]; acc[a][j0 + u][1] += ua.f[1] + ub.f[1];
                  acc[a][j0 + u][2] += ua.f[2] + ub.f[2]; acc[a][j0 + u][3] += ua.f[3] + ub.f[3];
                  continue;
                }
#endif
                acc[a][j0 + u] = mfma_h16(af[a], bfv[u], acc[a][j0 + u], 0, 0, 0);
              }
          }
        }
      }
    }
    WG2_STAMP();
  };
  if (!(p.dbg & 1) && t_begin < t_end) issue_tile(t_begin, smem, smem + p.xt.bytes);
  int cur = 0;
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();  // hipcc waits vmcnt(0) here: tile t has landed; and every wave is done reading the other buffer
    WG2_STAMP();
    tile_step(smem + cur * bufbytes, smem + (cur ^ 1) * bufbytes, t);
    if (dbuf) {
      cur ^= 1;
    } else if (t + 1 < t_end) {  // single buffer (the tile pair is too big for two): fetch the next tile once this one is consumed
      __syncthreads();
      if (!(p.dbg & 1)) issue_tile(t + 1, smem, smem + p.xt.bytes);
    }
  }
  __syncthreads();  // (the batched kernels start the next problem's DMA right away)

  // ---- write the partial slab straight from the accumulators.  Lane (g, t16) of fragment (a, j) holds
  // D[co = a*16 + 4g + e][ci = cb_j + t16]: 16 consecutive input channels = one 64-byte run per (co, tap).  Everything
  // but a per-lane constant is wave-uniform, so a store costs no address arithmetic: what made this tail expensive
  // before was 64-bit index math per store, not the stores.
  {
    // this lane's real input-channel index for fragment j (8-granular concat index -> segment -> OIHW channel), -1: padding
    const int cl = cA + t16;
    float* pw_lane = p.pw + ((size_t)sp * p.Co + co_base + g * 4) * TAPS * p.ci_total;
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      const int jf = wave + 4 * j;
      if (jf < njf) {
        const int c = cl + jcb[j];
        int sidx = 0;
#pragma unroll
        for (int k = 1; k < CGEN_MAX_SEG; ++k) sidx += (k < p.nseg && c >= p.seg_koff[k]) ? 1 : 0;
        int koff = p.seg_koff[0], segc = p.seg[0].c, soff = p.seg_off[0];
#pragma unroll
        for (int k = 1; k < CGEN_MAX_SEG; ++k)
          if (sidx == k) { koff = p.seg_koff[k]; segc = p.seg[k].c; soff = p.seg_off[k]; }
        const int cs = c - koff;
        const bool cv = jcb[j] + t16 < cw && cs < segc;
        float* col = pw_lane + jtap[j] * p.ci_total + soff + cs;
#pragma unroll
        for (int a = 0; a < NCF; ++a)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cv && co_base + a * 16 + g * 4 + e < p.Co) col[(size_t)(a * 16 + e) * TAPS * p.ci_total] = acc[a][j][e];
      }
    }
  }
  if (do_bias && (lane & 15) == 0) {
#pragma unroll
    for (int a = 0; a < NCF; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co_base + a * 16 + (lane >> 4) * 4 + e;
        if (co < p.Co) p.pb[(size_t)sp * p.Co + co] = accb[a][e];
      }
  }
  WG2_STAMP();
  if (stamp) p.stamps[63] = nst;
#undef WG2_STAMP
}

template <int NCF, int NJW, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_tile_kernel(Wg2P p) {
  wgrad_tile_body<NCF, NJW, KS>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Horizontally batched form: ONE launch runs the workgroups of many independent weight-gradient problems of the same
// kernel variant.  The engine defers all weight gradients of a step to the end of the backward pass, where they are
// independent; most of them (everything below 48x48) fill a fraction of the CUs with one or two tiles per workgroup,
// i.e. they are latency chains.  Packed into one grid, every CU always holds workgroups of *some* problem.
// blocks[b] = {problem, split, channel window, co range}; problems are read through a uniform pointer (scalar loads).
template <int NCF, int NJW, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_tile_batched_kernel(const Wg2P* __restrict__ probs, const int4* __restrict__ blocks, const int nblocks) {
  // gridDim.x < nblocks: a resident set of workgroups walks the block list (the engine caps the grid when the launch
  // runs in the background of the backward chain, so that the chain's kernels always find free CUs)
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int4 bi = blocks[b];
    // BY VALUE: through a reference into global memory hipcc must re-read every field after each barrier / store (it cannot
    // prove the table is not written), and every such s_load waits on lgkmcnt, i.e. on all LDS traffic in flight
    const Wg2P p = probs[__builtin_amdgcn_readfirstlane(bi.x)];
    wgrad_tile_body<NCF, NJW, KS>(p, __builtin_amdgcn_readfirstlane(bi.y), __builtin_amdgcn_readfirstlane(bi.z),
                                  __builtin_amdgcn_readfirstlane(bi.w));
    __syncthreads();
  }
}

// All variants in ONE launch: a flush of the deferred weight gradients used to be up to 14 dependent launches (one per
// variant and LDS class), each with its own tail and, in the background of the backward chain, each a barrier packet on
// the side queue -- the rocprofv3 timeline showed the chain being dispatched only every ~60 us while those packets waited
// on each other.  Here every workgroup walks the common block list and runs the body its problem asks for.
__global__ __launch_bounds__(256, 2) void wgrad_tile_mega_kernel(const Wg2P* __restrict__ probs, const int4* __restrict__ blocks, const int nblocks) {
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int4 bi = blocks[b];
    const Wg2P p = probs[__builtin_amdgcn_readfirstlane(bi.x)];  // by value (see the batched kernel)
    const int bx = __builtin_amdgcn_readfirstlane(bi.y), by = __builtin_amdgcn_readfirstlane(bi.z), bz = __builtin_amdgcn_readfirstlane(bi.w);
    switch (p.variant) {
      case 2: wgrad_tile_body<1, 16, 1>(p, bx, by, bz); break;
      case 3: wgrad_tile_body<1, 16, 3>(p, bx, by, bz); break;
      case 4: wgrad_tile_body<2, 12, 1>(p, bx, by, bz); break;
      case 5: wgrad_tile_body<2, 12, 3>(p, bx, by, bz); break;
      case 8: wgrad_tile_body<4, 6, 1>(p, bx, by, bz); break;
      case 9: wgrad_tile_body<4, 6, 3>(p, bx, by, bz); break;
      case 12: wgrad_tile_body<6, 4, 1>(p, bx, by, bz); break;
      case 13: wgrad_tile_body<6, 4, 3>(p, bx, by, bz); break;
      case 16: wgrad_tile_body<8, 3, 1>(p, bx, by, bz); break;
      case 33: wgrad_tile_body<1, 16, 7>(p, bx, by, bz); break;
      case 34: wgrad_tile_body<2, 16, 7>(p, bx, by, bz); break;
      case 36: wgrad_tile_body<2, 16, 1>(p, bx, by, bz); break;
      case 37: wgrad_tile_body<2, 16, 3>(p, bx, by, bz); break;
      default: wgrad_tile_body<8, 3, 3>(p, bx, by, bz); break;
    }
    __syncthreads();
  }
}

struct Wg2Geom { int ncf, njw, cwin, nsplit, tps, ntiles, tiles_x, tiles_y, n_cwin, n_co, dbuf; PixTile xt, gt; size_t lds; };

// returns false when the shape is not served by the tiled kernel
static bool wgrad2_geometry(int N, int H, int W, int co, int ctot8, int ks, Wg2Geom& g) {
  // tiny images waste most of a tile, but the packed launch still beats the generic kernel -- down to 1x1 images (round 2): the
  // 21 generic launches of ukbb192's 1x1-resolution layers sat in the exposed tail of the step (0.21 ms); packed: 16.89 vs 17.1 ms
  static const int min_hw = [] { const char* e = getenv("CGEN_WG2_MINHW"); return e ? atoi(e) : 1; }();
  if (!(ks == 1 || ks == 3 || ks == 7) || H < min_hw || W < min_hw) return false;
  const int taps = ks * ks;
  const int halo = ks / 2;
  g.ncf = co <= 16 ? 1 : (co <= 32 ? 2 : (co <= 64 ? 4 : (co <= 96 ? 6 : 8)));
  g.njw = g.ncf == 1 ? 16 : WG2_MAXACC / g.ncf;
  // The widest co block is not always the cheapest slab: the accumulators cap (co block) x (input-channel window), and every
  // window re-reads the gradient tile while every co block re-reads the input tile.  Pick the block width that moves the
  // fewest bytes per tile position (32 -> 160 @ 3x3: 128 + 32 co x two 16-channel windows = 151 KB; three 64-co blocks with
  // one 32-channel window = 82 KB).  A narrower block must save >= 10 %: it also means more LDS fragment reads per MFMA.
  static const int ncf_search = [] { const char* e = getenv("CGEN_WG2_NCF_SEARCH"); return e ? atoi(e) : 1; }();
  if (ncf_search && ks != 7) {
    // (co-block width in 16-column units, accumulator fragments per wave): the five standard slabs, plus a taller two-block
    // slab (32 fragments) that holds 112 input channels of a 3x3 conv in one window -- 96 -> 24 needs no second pass
    static const int cand[6][2] = {{1, 16}, {2, 12}, {2, 16}, {4, 6}, {6, 4}, {8, 3}};
    static const int tall = [] { const char* e = getenv("CGEN_WG2_TALL"); return e ? atoi(e) : 1; }();
    const int halo_ = ks / 2, c16_ = pad_to(ctot8, 16), co8 = pad_to(co, 8);
    const long xpix = (long)(TILE_H + 2 * halo_) * (TILE_W + 2 * halo_), gpix = (long)TILE_H * TILE_W;
    long best = -1, base = -1;
    int best_ncf = g.ncf, best_njw = g.njw;
    for (int k = 0; k < 6; ++k) {
      const int ncf = cand[k][0], njw = cand[k][1];
      if (ncf > g.ncf || (njw == 16 && ncf == 2 && !tall)) continue;
      const int mg = (4 * njw) / taps;
      if (mg < 1) continue;
      int cw = std::min(mg * 16, c16_);
      const PixTile gt = mk_pixtile(ncf * 16, 2, TILE_H, TILE_W);
      for (; cw >= 16; cw -= 16) {
        const PixTile xt = mk_pixtile(cw, 2, TILE_H + 2 * halo_, TILE_W + 2 * halo_);
        if (gt.ppp >= 1 && xt.ppp >= 1 && xt.bytes + gt.bytes <= 78 * 1024) break;
      }
      if (cw < 16) continue;
      const long n_cw = ceil_div(ctot8, cw), n_co = ceil_div(co, ncf * 16);
      const long bytes = n_co * n_cw * (xpix * std::min(cw, ctot8) * 2 + gpix * std::min(ncf * 16, co8) * 2);
      if (ncf == g.ncf && njw == g.njw) base = bytes;
      if (best < 0 || bytes < best) { best = bytes; best_ncf = ncf; best_njw = njw; }
    }
    static const int ncf_pct = [] { const char* e = getenv("CGEN_WG2_NCF_PCT"); return e ? atoi(e) : 90; }();
    if (base > 0 && (best_ncf != g.ncf || best_njw != g.njw) && best * 100 <= base * ncf_pct) {
      g.ncf = best_ncf;
      g.njw = best_njw;
    }
  }
  if (ks == 7) {  // the 7x7 stem (Cin <= 8: ONE 16-channel group, 49 taps -> 49 fragments = 13 per wave): at most 32 co columns per workgroup
    if (ctot8 > 16) return false;
    g.ncf = co <= 16 ? 1 : 2;
    g.njw = 16;
  }
  g.n_co = ceil_div(co, g.ncf * 16);
  int maxgrp = (4 * g.njw) / taps;  // 16-channel groups per window that fit the accumulators
  if (maxgrp < 1) return false;
  int cwin = maxgrp * 16;
  const int c16 = pad_to(ctot8, 16);
  if (cwin > c16) cwin = c16;
  g.gt = mk_pixtile(g.ncf * 16, 2, TILE_H, TILE_W);
  for (;; cwin -= 16) {  // LDS budget: two workgroups per CU
    if (cwin < 16) return false;
    g.xt = mk_pixtile(cwin, 2, TILE_H + 2 * halo, TILE_W + 2 * halo);
    if (g.gt.ppp >= 1 && g.xt.ppp >= 1 && g.xt.bytes + g.gt.bytes <= 78 * 1024) break;
  }
  if (g.gt.ppp < 1) return false;
  {  // equal windows: 96 channels as 48 + 48 rather than 80 + 16 -- same passes over the gradient tile, a smaller input tile
     // (which may make room for the second tile buffer) and workgroups of equal length.  Measured: the packed launches alone
     // 4.00 -> 3.88 ms, the STEP 1940 -> 1928 img/s (three interleaved pairs) -- off.
    static const int balance = [] { const char* e = getenv("CGEN_WG2_BALANCE"); return e ? atoi(e) : 0; }();
    const int nwin = ceil_div(ctot8, cwin);
    const int bal = pad_to(ceil_div(ctot8, nwin), 16);
    if (balance && nwin > 1 && bal < cwin) {
      cwin = bal;
      g.xt = mk_pixtile(cwin, 2, TILE_H + 2 * halo, TILE_W + 2 * halo);
    }
  }
  g.lds = (size_t)g.xt.bytes + g.gt.bytes;
  // two tile buffers where they fit next to a second workgroup on the CU: the next tile's DMA runs under this tile's MFMAs
  // (ablation on ukbb192: DMA-only 2.1 ms, MFMA-only ~1.6 ms, both 4.6 ms with one buffer -- the two phases did not overlap)
  static const int dbuf_on = [] { const char* e = getenv("CGEN_WG2_DBUF"); return e ? atoi(e) : 1; }();
  g.dbuf = (dbuf_on && 2 * g.lds <= 78 * 1024) ? 1 : 0;
  if (g.dbuf) g.lds *= 2;
  g.cwin = cwin;
  g.n_cwin = ceil_div(ctot8, cwin);
  g.tiles_x = ceil_div(W, TILE_W); g.tiles_y = ceil_div(H, TILE_H);
  g.ntiles = N * g.tiles_x * g.tiles_y;
  static const int want_total = [] { const char* e = getenv("CGEN_WG2_WANT"); return e ? atoi(e) : 32; }();  // (round 2: 160 -> 64 -> 32 workgroups per problem; with 16 tiles per workgroup: split-K partials 945 -> 429 MB per ukbb192 step, 1779 -> 1930 img/s overall)
  // workgroups per problem: the batched launch packs all problems of a step into one grid, so a problem need not fill
  // the chip by itself -- fewer, longer workgroups mean fewer split-K partials (measured optimum on MI355X: ~160 / >= 4 tiles)
  int want = ceil_div(want_total, g.n_cwin * g.n_co);
  {  // bound the split-K partials of one conv (they are written and re-read by cgen_wgrad_reduce)
    const char* e = getenv("CGEN_WG2_PARTIAL_MB");
    const long cap = (e ? atol(e) : 4096) << 20;  // off by default: capping costs more wgrad time than it saves in the reduce
    const long wbytes = 4L * co * taps * ctot8;
    const int maxs = (int)(cap / (wbytes > 0 ? wbytes : 1));
    if (want > maxs) want = maxs;
  }
  if (want < 1) want = 1;
  g.tps = ceil_div(g.ntiles, want);
  // long workgroups: the packed launch fills the chip whatever a single problem does, and every split costs a partial slab
  // (12 tiles per workgroup halves the split-K partials of ukbb192, 1.8 -> 0.94 GB per step, +1.7 %); the batch-256 32x32
  // models measured 1-2 % better with 4
  static const int min_tps_env = [] { const char* e = getenv("CGEN_WG2_MINTPS"); return e ? atoi(e) : 0; }();
  const int min_tps = min_tps_env > 0 ? min_tps_env : (N <= 64 ? 16 : 4);
  if (g.tps < min_tps && g.ntiles >= min_tps) g.tps = min_tps;
  g.nsplit = ceil_div(g.ntiles, g.tps);
  return true;
}

template <int NCF, int NJW>
static void launch_wgrad2_ks(const Wg2P& p, const Wg2Geom& g, hipStream_t st) {
  dim3 grid(g.nsplit, g.n_cwin, g.n_co), block(256);
  if (p.KS == 7) {
    if constexpr (NJW == 16 && NCF <= 2) {
      static bool once7 = false;
      if (!once7) { (void)hipFuncSetAttribute((const void*)wgrad_tile_kernel<NCF, 16, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once7 = true; }
      hipLaunchKernelGGL((wgrad_tile_kernel<NCF, 16, 7>), grid, block, g.lds, st, p);
    }
    return;
  }
  if (p.KS == 3) {
    static bool once3 = false;
    if (!once3) { (void)hipFuncSetAttribute((const void*)wgrad_tile_kernel<NCF, NJW, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once3 = true; }
    hipLaunchKernelGGL((wgrad_tile_kernel<NCF, NJW, 3>), grid, block, g.lds, st, p);
  } else {
    static bool once1 = false;
    if (!once1) { (void)hipFuncSetAttribute((const void*)wgrad_tile_kernel<NCF, NJW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once1 = true; }
    hipLaunchKernelGGL((wgrad_tile_kernel<NCF, NJW, 1>), grid, block, g.lds, st, p);
  }
}

// ============================================================================= lean persistent conv for short-K convs (bf16)
// PMC counters on MI355X showed what actually bounds the short-K / wide-output convs (the C/4 -> C half of every Block,
// forward and data-gradient, and the 1x1 projections): the one-tile-per-workgroup kernel issues ~1150 VALU and ~700 SALU
// instructions per wave for 40 MFMAs -- 64-bit address arithmetic per epilogue chunk, software bf16 rounding, per-piece
// DMA address decoding, the weight-slab copy repeated for every tile.  A wave64 VALU instruction occupies the SIMD for
// 4 cycles (quarter-rate integer multiplies 16), an MFMA 16x16x32 for 16: the kernel was VALU-issue bound.
// This kernel does the same math with ~5x fewer instructions per tile:
//   * persistent workgroups: the weight slab of the workgroup's output channels is copied to LDS ONCE;
//   * everything that depends only on the lane (DMA lane -> channel/segment mapping, K-step -> LDS offset table,
//     epilogue element offsets, bias) is computed once per launch; per tile the tile origin is SCALAR arithmetic and
//     every global address is (uniform 64-bit base) + (32-bit lane offset);
//   * weight rows are permuted inside each 32-row block so that a lane ends up holding 8 CONSECUTIVE output channels of
//     its pixel (two MFMA results): the epilogue is a 16-byte load / store per lane straight from the accumulators --
//     no LDS staging, no extra barrier; bias is the accumulators' initial value; bf16 rounding is v_cvt_pk_bf16_f32;
//   * epilogue operands (aux, residual) are requested before the halo DMA wait and consumed after the MFMA loop.
struct PxP {
  PixTile xt;
  int tiles_x, tiles_y, ntiles, nks;
  int pad_x0, pad_x;
  int ldw, wpieces, rows_pad, co8;  // co8: output channels incl. zero padding to 8 (== Co unless the output view carries cpad)
  FastDiv d_gprw, d_ctot8, d_tx, d_ty;
  int ktab[PX_MAXKS * 4];  // [K-step][lane group fg]: ((tap row * rowbytes + channel * 2) << 3) | (tap column == 2) << 2 | tap column -- read with VECTOR loads from the kernarg segment
  unsigned long long* stamps;  // optional (CGEN_PX_STAMPS): 100 MHz wall-clock stamps {entry, tile landed, MFMAs done, exit} of every workgroup
};

// bf16 pair -> two floats

template <int NP, int KS, bool ONESEG, bool REM>
__global__ __launch_bounds__(256, (NP <= 2 && !REM) ? 3 : 2) void conv_px_kernel(ConvP p, PxP q) {
  CGEN_SETPRIO();  // the chain's waves win issue arbitration over background weight-gradient waves on the same SIMD
  typedef h16_t T;
  constexpr int G = 8, HALO = KS / 2, HW = TILE_W + 2 * HALO, TAPS = KS * KS;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wsb = smem;
  char* Xb = smem + (size_t)q.wpieces * 1024;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int co_base = blockIdx.y * (NP * 32);
  unsigned long long* stamp = (q.stamps != nullptr && tid == 0) ? q.stamps + 8 * (blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
  if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();

  // ---- weight slab, once.  LDS row (pr*32 + h*16 + j) holds output channel pr*32 + (j>>2)*8 + h*4 + (j&3): MFMA "h" of
  // pair pr then leaves channels 8*fg + 4*h + {0..3} of the pair in lane group fg
  {
    // branch-free: a lane past the slab's last group re-copies the last group into its own (padding) slot, rows past the
    // weight image are clamped to its last row (their outputs are never stored); columns never leave the row (ldw <= krow)
    const int gpr_w = q.ldw / G, last = NP * 32 * gpr_w - 1;
    for (int piece = wave; piece < q.wpieces; piece += 4) {
      const int pu = __builtin_amdgcn_readfirstlane(piece);
      const int gi = min(pu * 64 + lane, last);
      const int r = fdiv(gi, q.d_gprw), k = (gi - r * gpr_w) * G;
      const int j = r & 15, h = (r >> 4) & 1, pr = r >> 5;
      const int row = min(co_base + pr * 32 + (j >> 2) * 8 + h * 4 + (j & 3), q.rows_pad - 1);
      const T* src = (const T*)p.w + (__umul24(row, p.krow) + k);
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Wsb + (size_t)pu * 1024), 16, 0, 0);
    }
  }
  if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
  // ---- lane constants
  // K-step -> byte offset of this lane's 8 input channels inside a tile row (tap shift included).  The (K-step, lane group)
  // -> (tap, channel) decoding is done on the host: q.ktab sits in the kernarg segment, which is ordinary memory -- every
  // lane fetches its entries with vector loads that are all in flight together.  They are CONSUMED only after the first
  // tile's DMA has been issued (vmcnt is in order: consuming them here would expose the weight-slab latency).
  int koff[PX_MAXKS], kt[PX_MAXKS];
  {
    constexpr size_t q_off = (sizeof(ConvP) + alignof(PxP) - 1) / alignof(PxP) * alignof(PxP);
    const int* ktab = (const int*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + q_off + offsetof(PxP, ktab)) + fg;
#pragma unroll
    for (int i = 0; i < PX_MAXKS; ++i) { kt[i] = ktab[i * 4]; koff[i] = 0; }
  }
  bool koff_pending = true;
  // halo DMA: lane -> (pixel inside a piece, 16-byte channel group) -> segment, element offset (see wgrad_tile_kernel)
  const int xl = fdiv(lane, q.xt.d_gpr), xcg = lane - xl * q.xt.gpr;
  int x_si = 0, x_off = 0;
  const bool x_lane = xl < q.xt.ppp && xcg * G < p.ctot8;
  bool x_data = false;
  LaneTile LX;
  LX.xl = xl; LX.lane = x_lane;
  if (ONESEG) {
    const int c = xcg * G;
    x_data = x_lane && c < p.seg[0].c;
    x_off = (int)__umul24(xl, (int)p.seg[0].sw) + c;
    LX.sh = (int)p.seg[0].sh; LX.swp = (int)(p.seg[0].sw * q.xt.ppp);
  } else {
    const int c = xcg * G;
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k) x_si += (k < p.nseg && c >= p.seg_koff[k]) ? 1 : 0;
    View sv = p.seg[0];
    int ko = p.seg_koff[0];
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k)
      if (x_si == k) { sv = p.seg[k]; ko = p.seg_koff[k]; }
    const int cs = c - ko;
    x_data = x_lane && cs < sv.c;
    x_off = (int)(xl * sv.sw) + cs;
    LX.sh = (int)p.seg[0].sh; LX.swp = (int)(p.seg[0].sw * q.xt.ppp);
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k)
      if (x_si == k) { LX.sh = (int)p.seg[k].sh; LX.swp = (int)(p.seg[k].sw * q.xt.ppp); }
  }
  LX.data = x_data;
  const int xpieces = q.xt.rows * q.xt.ppr;
  // epilogue: this lane owns pixel (row wave*2 + f, column fr) and channels co_base + pr*32 + fg*8 .. +8
  const int ch0 = co_base + fg * 8;
  const bool has_aux = p.aux.p != nullptr, has_r1 = p.res1.p != nullptr, has_r2 = p.res2.p != nullptr;
  // remainder planes of the f16 residual trunk (cgen_conv_args): an instance of their own, the plain one keeps its registers
  const bool has_r1l = REM && p.r1_rem != 0, has_orem = REM && p.out_rem != 0;
  int eo_out[2], eo_aux[2], eo_r1[2], eo_r2[2];  // byte offsets from the tile origin of each tensor (strides < 2^24)
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int ry = wave * 2 + f;
    eo_out[f] = (int)(__umul24(ry, (int)p.out.sh) + __umul24(fr, (int)p.out.sw) + ch0) * 2;
    eo_aux[f] = (int)(__umul24(ry, (int)p.aux.sh) + __umul24(fr, (int)p.aux.sw) + ch0) * 2;  // (absent operands: strides are 0)
    eo_r1[f] = (int)(__umul24(ry, (int)p.res1.sh) + __umul24(fr, (int)p.res1.sw) + ch0) * 2;
    eo_r2[f] = (int)(__umul24(ry, (int)p.res2.sh) + __umul24(fr, (int)p.res2.sw) + ch0) * 2;
  }
  f32x4 binit[NP][2];
#pragma unroll
  for (int pr = 0; pr < NP; ++pr)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int co = ch0 + pr * 32 + h * 4;
      // (Co % 4 == 0: checked on the host); no bias / channels past Co: an unconditional load of zeros, no branch
      const float4 b = *(const float4*)((p.bias && co + 4 <= p.Co) ? (const void*)(p.bias + co) : (const void*)g_zero16);
      binit[pr][h] = (f32x4){b.x, b.y, b.z, b.w};
    }
  const char* a_base = Wsb + (size_t)fr * q.ldw * 2 + fg * 16;
  const char* x_rows = Xb + (size_t)(wave * 2) * q.xt.rowbytes;

  if (stamp) stamp[5] = __builtin_amdgcn_s_memrealtime();
  for (int t = (int)blockIdx.x; t < q.ntiles; t += gridDim.x) {
    // ---- tile origin: scalar
    const int b1 = fdiv(t, q.d_tx), tx = t - b1 * q.tiles_x;
    const int n = fdiv(b1, q.d_ty), ty = b1 - n * q.tiles_y;
    const int y0 = ty * TILE_H, x0 = tx * TILE_W;
    {  // halo tile DMA (row pieces)
      const T* my_org = vptr32<T>(p.seg[0], n, y0 - HALO, x0 - HALO);
      if (!ONESEG && p.nseg > 1) {
        if (x_si == 1) my_org = vptr32<T>(p.seg[1], n, y0 - HALO, x0 - HALO);
        if (x_si == 2) my_org = vptr32<T>(p.seg[2], n, y0 - HALO, x0 - HALO);
        if (x_si == 3) my_org = vptr32<T>(p.seg[3], n, y0 - HALO, x0 - HALO);
      }
      dma_tile<T>(q.xt, LX, my_org + x_off, Xb, wave, max(0, HALO - y0), min(TILE_H + 2 * HALO, p.H + HALO - y0), max(0, HALO - x0), min(HW, p.W + HALO - x0));
    }
    // ---- epilogue operands: requested now, consumed after the MFMA loop
    const bool colv = x0 + fr < p.W;
    bool pv[2];
    uint4 ea[NP][2], er[NP][2], el[REM ? NP : 1][2];
    {
      const char* aux_t = has_aux ? (const char*)vptr32<T>(p.aux, n, y0, x0) : nullptr;
      const char* r1_t = has_r1 ? (const char*)vptr32<T>(p.res1, n, y0, x0) : nullptr;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        pv[f] = colv && y0 + wave * 2 + f < p.H;
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
          const bool ok = pv[f] && ch0 + pr * 32 + 8 <= q.co8;
          ea[pr][f] = make_uint4(0, 0, 0, 0);
          er[pr][f] = make_uint4(0, 0, 0, 0);
          if (has_aux) ea[pr][f] = *(const uint4*)(ok ? aux_t + eo_aux[f] + pr * 64 : (const char*)g_zero16);
          if (has_r1) er[pr][f] = *(const uint4*)(ok ? r1_t + eo_r1[f] + pr * 64 : (const char*)g_zero16);
          if constexpr (REM) {
            el[pr][f] = make_uint4(0, 0, 0, 0);
            if (has_r1l) el[pr][f] = *(const uint4*)(ok ? r1_t + p.r1_rem + eo_r1[f] + pr * 64 : (const char*)g_zero16);
          }
        }
      }
    }
    if (koff_pending) {
      koff_pending = false;
#pragma unroll
      for (int i = 0; i < PX_MAXKS; ++i) asm volatile("" : "+v"(kt[i]));  // pins the consumption (and its vmcnt wait) below the DMA issue
      const int px0 = pix_off(q.xt, 0, fr);
      const int d01 = KS > 1 ? pix_off(q.xt, 0, fr + 1) - px0 : 0, d02 = KS > 1 ? pix_off(q.xt, 0, fr + 2) - px0 : 0;
      const int e02 = d02 - 2 * d01;  // offset(dx) = dx * d01 + (dx == 2) * e02: two multiply-adds, no compare/select chain
#pragma unroll
      for (int i = 0; i < PX_MAXKS; ++i) {
        if (KS > 1) koff[i] = (int)__umul24(kt[i] & 3, d01) + __mul24((kt[i] >> 2) & 1, e02) + ((kt[i] >> 3) + px0);  // e02 < 0 when only x+1 crosses a piece
        else koff[i] = (kt[i] >> 3) + px0;
      }
    }
    if (stamp) stamp[6] = __builtin_amdgcn_s_memrealtime();
    // ---- activation in place on the pieces this wave fetched itself (hipcc waits for the DMAs first), then hand over
    if (p.act != CGEN_ACT_NONE) act_pieces<T>(Xb, wave, lane, xpieces, p.act);
    if (stamp) stamp[7] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();

    // ---- MFMAs: 2 tile rows x NP*32 channels per wave, bias as the initial value
    f32x4 acc[NP][2][2];
    {  // K-step 0 always exists: the bias registers are its C operand (no accumulator initialisation moves)
      const h16x8 b0 = *(const h16x8*)(x_rows + koff[0]);
      const h16x8 b1 = *(const h16x8*)(x_rows + q.xt.rowbytes + koff[0]);
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const h16x8 aq = *(const h16x8*)(a_base + (size_t)(pr * 32 + h * 16) * q.ldw * 2);
          acc[pr][h][0] = mfma_h16(aq, b0, binit[pr][h], 0, 0, 0);
          acc[pr][h][1] = mfma_h16(aq, b1, binit[pr][h], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 1; i < PX_MAXKS; ++i) {
      if (i >= q.nks) break;  // (a break, not a guard: one scalar compare-and-branch per K-step instead of 15 precomputed masks)
      const h16x8 b0 = *(const h16x8*)(x_rows + koff[i]);
      const h16x8 b1 = *(const h16x8*)(x_rows + q.xt.rowbytes + koff[i]);
#pragma unroll
      for (int pr = 0; pr < NP; ++pr)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const h16x8 aq = *(const h16x8*)(a_base + (size_t)(pr * 32 + h * 16) * q.ldw * 2 + i * 64);
          acc[pr][h][0] = mfma_h16(aq, b0, acc[pr][h][0], 0, 0, 0);
          acc[pr][h][1] = mfma_h16(aq, b1, acc[pr][h][1], 0, 0, 0);
        }
    }

    if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue straight from the accumulators: v = acc * act'(aux) + res1 + res2 -> bf16 -> one 16-byte store
    {
      char* out_t = (char*)vptr32<T>(p.out, n, y0, x0);
      const char* r2_t = has_r2 ? (const char*)vptr32<T>(p.res2, n, y0, x0) : nullptr;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
          if (!(pv[f] && ch0 + pr * 32 + 8 <= q.co8)) continue;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = acc[pr][0][f][e]; v[4 + e] = acc[pr][1][f][e]; }
          if (has_aux) {
            const uint32_t w[4] = {ea[pr][f].x, ea[pr][f].y, ea[pr][f].z, ea[pr][f].w};
            if (p.dact == CGEN_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] = h_lo(w[e]) > 0.f ? v[2 * e] : 0.f;
                v[2 * e + 1] = h_hi(w[e]) > 0.f ? v[2 * e + 1] : 0.f;
              }
            } else if (p.dact == CGEN_ACT_GELU) {
              const F8 gp = gelu8_bwd_h16(make_uint4(w[0], w[1], w[2], w[3]));
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= gp.v[e];
            }
          }
          if (has_r1) {
            const uint32_t w[4] = {er[pr][f].x, er[pr][f].y, er[pr][f].z, er[pr][f].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += h_lo(w[e]); v[2 * e + 1] += h_hi(w[e]); }
          }
          if (has_r2) {
            const uint4 r = *(const uint4*)(r2_t + eo_r2[f] + pr * 64);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += h_lo(w[e]); v[2 * e + 1] += h_hi(w[e]); }
          }
          if (REM && has_r1l) {
            const uint32_t w[4] = {el[REM ? pr : 0][f].x, el[REM ? pr : 0][f].y, el[REM ? pr : 0][f].z, el[REM ? pr : 0][f].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += h_lo(w[e]); v[2 * e + 1] += h_hi(w[e]); }
          }
          uint4 o;
          o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]); o.z = f2h_pk(v[4], v[5]); o.w = f2h_pk(v[6], v[7]);
          if (has_orem) {  // what the rounding just dropped goes to the remainder plane
            uint4 ol;
            ol.x = f2h_pk(v[0] - h_lo(o.x), v[1] - h_hi(o.x)); ol.y = f2h_pk(v[2] - h_lo(o.y), v[3] - h_hi(o.y));
            ol.z = f2h_pk(v[4] - h_lo(o.z), v[5] - h_hi(o.z)); ol.w = f2h_pk(v[6] - h_lo(o.w), v[7] - h_hi(o.w));
            *(uint4*)(out_t + p.out_rem + eo_out[f] + pr * 64) = ol;
          }
          *(uint4*)(out_t + eo_out[f] + pr * 64) = o;
        }
    }
    __syncthreads();  // every wave is done reading the tile before the next DMA overwrites it
    if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int NP, int KS, bool ONESEG, bool REM>
static void launch_px_inst3(const ConvP& p, const PxP& q, dim3 grid, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)conv_px_kernel<NP, KS, ONESEG, REM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL((conv_px_kernel<NP, KS, ONESEG, REM>), grid, dim3(256), lds, st, p, q);
}
template <int NP, int KS, bool ONESEG>
static void launch_px_inst2(const ConvP& p, const PxP& q, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (NP <= 3) {  // (the remainder-plane instance of four channel pairs would spill: launch_conv_px caps np at 3 for it)
    if (p.out_rem || p.r1_rem) { launch_px_inst3<NP, KS, ONESEG, true>(p, q, grid, lds, st); return; }
  }
  launch_px_inst3<NP, KS, ONESEG, false>(p, q, grid, lds, st);
}
template <int NP>
static void launch_px_inst(const ConvP& p, const PxP& q, dim3 grid, size_t lds, hipStream_t st) {
  if (p.KS == 3) {
    if (p.nseg == 1) launch_px_inst2<NP, 3, true>(p, q, grid, lds, st);
    else launch_px_inst2<NP, 3, false>(p, q, grid, lds, st);
  } else {
    if (p.nseg == 1) launch_px_inst2<NP, 1, true>(p, q, grid, lds, st);
    else launch_px_inst2<NP, 1, false>(p, q, grid, lds, st);
  }
}


static bool launch_conv_px(const ConvP& p, hipStream_t st) {
  const int G = 8, halo = p.KS / 2;
  PxP q;
  memset(&q, 0, sizeof(q));
  q.nks = ceil_div(p.taps * p.ctot8, 32);
  if (q.nks > PX_MAXKS || !p.epi_vec16) return false;
  q.co8 = p.Co;
  if (p.Co % 8 != 0) {
    // ragged output width: the rows past Co of the weight image are zero, so the kernel may write whole 8-channel chunks
    // if the output asks for zero padding (out.cpad) and every epilogue operand is zero-padded as well
    const int c8 = pad_to(p.Co, 8);
    auto padded = [&](const View& v) { return !v.p || v.cpad >= c8; };
    if (p.Co % 4 != 0 || p.out.cpad < c8 || !padded(p.aux) || !padded(p.res1) || !padded(p.res2)) return false;
    q.co8 = c8;
  }
  // the kernel does all its address arithmetic in 32 bits: every view must span less than 2^31 bytes (tile overhang included)
  for (int sg = 0; sg < p.nseg; ++sg)
    if (!fits_i32(p.seg[sg], p.N, p.H + TILE_H + 2, p.W + TILE_W + 2)) return false;
  if (!fits_i32(p.out, p.N, p.H + TILE_H, p.W + TILE_W) || !fits_i32(p.aux, p.N, p.H + TILE_H, p.W + TILE_W) ||
      !fits_i32(p.res1, p.N, p.H + TILE_H, p.W + TILE_W) || !fits_i32(p.res2, p.N, p.H + TILE_H, p.W + TILE_W)) return false;
  q.xt = mk_pixtile(p.ctot8, 2, TILE_H + 2 * halo, TILE_W + 2 * halo);
  if (q.xt.ppp < 1) return false;
  q.ldw = lds_stride(q.nks * 32, 2);  // <= ceil32(K) + 16 <= krow
  q.rows_pad = pad_to(p.Co, 16);
  q.tiles_x = ceil_div(p.W, TILE_W); q.tiles_y = ceil_div(p.H, TILE_H);
  q.ntiles = p.N * q.tiles_x * q.tiles_y;
  // channel pairs per workgroup: as many as keep three workgroups per CU (LDS) -- fewer re-reads of the input tile --
  // but small launches are split further so every CU has work
  const int np_all = ceil_div(p.Co, 32);
  int np = np_all < 4 ? np_all : 4;
  auto lds_of = [&](int n_) { return (size_t)ceil_div(n_ * 32 * (q.ldw / G), 64) * 1024 + (size_t)q.xt.bytes; };
  static const int lds_budget = [] { const char* e = getenv("CGEN_PX_LDS"); return (e ? atoi(e) : 52) * 1024; }();
  while (np > 1 && lds_of(np) > (size_t)lds_budget) --np;
  if (np == 3 && np_all == 4) np = 2;
  if ((p.out_rem || p.r1_rem) && np > 3) np = np_all % 3 == 0 ? 3 : 2;  // remainder planes: one more operand set in registers
  // Two pairs at most by default: the one- and two-pair instances fit 168 VGPRs (three workgroups per CU, which is what the grid
  // below assumes), the three- and four-pair ones need 208 / 248 and leave a third of the launched workgroups waiting for a slot.
  // Measured on ukbb192 B = 32 (interleaved A/B, two rounds): 1954 / 1957 img/s before, 1969 / 1964 with the cap and
  // __launch_bounds__(256, 3) on the small instances (alone: 1955 / 1959).
  static const int max_np = [] { const char* e = getenv("CGEN_PX_MAXNP"); return e ? atoi(e) : 2; }();
  if (np > max_np) np = max_np;
  static const int min_wgs = [] { const char* e = getenv("CGEN_PX_MINWG"); return e ? atoi(e) : 512; }();
  while (np > 1 && (int64_t)q.ntiles * ceil_div(np_all, np) < min_wgs) --np;
  const size_t lds = lds_of(np);
  if (lds > 150 * 1024) return false;
  q.wpieces = ceil_div(np * 32 * (q.ldw / G), 64);
  q.d_gprw = mk_fastdiv(q.ldw / G); q.d_ctot8 = mk_fastdiv(p.ctot8);
  q.d_tx = mk_fastdiv(q.tiles_x); q.d_ty = mk_fastdiv(q.tiles_y);
  for (int i = 0; i < q.nks; ++i)
    for (int g4 = 0; g4 < 4; ++g4) {
      const int kidx = i * 32 + g4 * 8;
      int tap = kidx / p.ctot8;
      const int c = kidx - tap * p.ctot8;
      if (tap >= p.taps) tap = p.taps - 1;  // columns past the last tap carry zero weights; keep the address legal
      const int dy = tap / p.KS, dx = tap - dy * p.KS;
      q.ktab[i * 4 + g4] = ((dy * q.xt.rowbytes + c * 2) << 3) | ((dx == 2) << 2) | dx;
    }
  { const char* e = getenv("CGEN_PX_STAMPS"); q.stamps = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
  const int parts = ceil_div(np_all, np);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 3) per_cu = 3;
  if (per_cu < 1) per_cu = 1;
  int gx = 256 * per_cu / parts;
  if (gx < 1) gx = 1;
  if (gx > q.ntiles) gx = q.ntiles;
  q.pad_x = 0;
  dim3 grid(gx, parts);
  switch (np) {
    case 1: launch_px_inst<1>(p, q, grid, lds, st); break;
    case 2: launch_px_inst<2>(p, q, grid, lds, st); break;
    case 3: launch_px_inst<3>(p, q, grid, lds, st); break;
    case 4: launch_px_inst<4>(p, q, grid, lds, st); break;
    default: return false;
  }
  return true;
}

// ============================================================================= weight-stationary persistent conv (bf16)
// fwd / dgrad for KS in {1,3} on >= 5x5 images.  The forward kernel above re-stages the weight slab in LDS for every
// 128-pixel tile (for a 96->24 3x3 conv that is 55 KB of weights per 25 KB of activations).  Here the K axis
// (tap, channel) is split across the four waves and every wave keeps ITS K-steps of the weight image in registers for
// the whole launch; a persistent workgroup then streams 8x16 pixel tiles through LDS (row-piece DMA as in the wgrad
// kernel), each wave accumulates all 128 pixels x (16*NTC) channels over its K-steps, and the four partial sums are
// combined through LDS in a fixed order (deterministic) before the fused epilogue.  LDS holds only the halo tile.
struct WsP {
  PixTile xt;
  int tiles_x, tiles_y, ntiles, nk;  // nk = K-steps that carry weights
  int rows_pad, red_bytes, dbg, pad0;
  FastDiv d_ctot8, d_tx, d_ty;
  unsigned long long* stamps;  // optional (CGEN_WS_STAMPS): per-phase cycle stamps of workgroup 0
};

template <int NTC, int NKW, bool ONESEG>
__global__ __launch_bounds__(256, 2) void conv_ws_kernel(ConvP p, WsP q) {
  CGEN_SETPRIO();  // the chain's waves win issue arbitration over background weight-gradient waves on the same SIMD
  typedef h16_t T;
  constexpr int G = 8, NF = NTC * TILE_H;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  const int KS = p.KS, HALO = KS / 2, TAPS = p.taps;
  const int HW = TILE_W + 2 * HALO;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int co_base = blockIdx.y * (NTC * 16);

  // ---- this wave's K-steps of the weight image, resident in registers
  h16x8 aw[NTC][NKW];
  int koff[NKW];
  {
    int pxo[3];  // LDS byte offset of pixel x = fr + dx inside a tile row
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) pxo[dx] = pix_off(q.xt, 0, fr + (dx < KS ? dx : 0));
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
      const int ks = wave + 4 * i;
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
        // unconditional: rows past the image are clamped to its last row (never stored), K-steps past nk read the 32 zero
        // columns that close every weight row (krow = ceil32(K) + 32)
        const int row = min(co_base + t * 16 + fr, q.rows_pad - 1);
        aw[t][i] = *(const h16x8*)((const T*)p.w + (__umul24(row, p.krow) + min(ks, q.nk) * 32 + fg * 8));
      }
      const int kidx = min(ks, q.nk - 1) * 32 + fg * 8;
      int tap = fdiv(kidx, q.d_ctot8);
      const int c = kidx - tap * p.ctot8;
      tap = tap < TAPS ? tap : TAPS - 1;  // columns past the last tap carry zero weights; keep the address legal
      const int dy = KS == 3 ? (tap * 11) >> 5 : 0, dx = tap - dy * KS;  // (KS in {1, 3}: tap < 9)
      koff[i] = dy * q.xt.rowbytes + (dx == 0 ? pxo[0] : (dx == 1 ? pxo[1] : pxo[2])) + c * 2;
    }
  }

  // ---- per-lane DMA constants (see wgrad_tile_kernel)
  const int xl = fdiv(lane, q.xt.d_gpr), xcg = lane - xl * q.xt.gpr;
  int x_si = 0, x_off = 0;
  const bool x_lane = xl < q.xt.ppp && xcg * G < p.ctot8;
  bool x_data = false;
  LaneTile LX;
  LX.xl = xl; LX.lane = x_lane;
  if (ONESEG) {
    const int c = xcg * G;
    x_data = x_lane && c < p.seg[0].c;
    x_off = (int)__umul24(xl, (int)p.seg[0].sw) + c;
    LX.sh = (int)p.seg[0].sh; LX.swp = (int)(p.seg[0].sw * q.xt.ppp);
  } else {
    const int c = xcg * G;
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k) x_si += (k < p.nseg && c >= p.seg_koff[k]) ? 1 : 0;
    View sv = p.seg[0];
    int ko = p.seg_koff[0];
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k)
      if (x_si == k) { sv = p.seg[k]; ko = p.seg_koff[k]; }
    const int cs = c - ko;
    x_data = x_lane && cs < sv.c;
    x_off = (int)(xl * sv.sw) + cs;
    LX.sh = (int)p.seg[0].sh; LX.swp = (int)(p.seg[0].sw * q.xt.ppp);
#pragma unroll
    for (int k = 1; k < CGEN_MAX_SEG; ++k)
      if (x_si == k) { LX.sh = (int)p.seg[k].sh; LX.swp = (int)(p.seg[k].sw * q.xt.ppp); }
  }
  LX.data = x_data;
  const int xpieces = q.xt.rows * q.xt.ppr;
  // ---- epilogue lane constants (the host only selects this kernel when every epilogue access is a whole aligned chunk):
  // chunk k of this lane = pixel (row ef[k], column ex[k]) of the tile, channels co_base + ech*8 .. +8
  constexpr int CPP = NTC * 2;  // 16-byte chunks per pixel; divides 64, so a lane's chunk column is the same for every k
  const int ech = lane % CPP;
  const bool chv = co_base + ech * 8 + 8 <= p.Co;
  Bias8 ebias;
  ebias.b0 = ebias.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (chv) bias8_load(p, co_base + ech * 8, ebias);
  int ef[NTC], ex[NTC], erd[NTC], eo_out[NTC], eo_aux[NTC], eo_r1[NTC], eo_r2[NTC];
#pragma unroll
  for (int k = 0; k < NTC; ++k) {
    const int pl = (lane + 64 * k) / CPP;
    ef[k] = wave * 2 + (pl >> 4); ex[k] = pl & 15;
    erd[k] = ((ech >> 1) * TILE_H + ef[k]) * 64 + (ech & 1) * 32 + ex[k];  // f32x4 index of the "lo" half in the partial-sum area
    const int c0 = co_base + ech * 8;
    eo_out[k] = (int)(__umul24(ef[k], (int)p.out.sh) + __umul24(ex[k], (int)p.out.sw) + c0) * 2;  // strides < 2^24
    eo_aux[k] = (int)(__umul24(ef[k], (int)p.aux.sh) + __umul24(ex[k], (int)p.aux.sw) + c0) * 2;
    eo_r1[k] = (int)(__umul24(ef[k], (int)p.res1.sh) + __umul24(ex[k], (int)p.res1.sw) + c0) * 2;
    eo_r2[k] = (int)(__umul24(ef[k], (int)p.res2.sh) + __umul24(ex[k], (int)p.res2.sw) + c0) * 2;
  }
  const bool has_aux = p.aux.p != nullptr, has_r1 = p.res1.p != nullptr, has_r2 = p.res2.p != nullptr;
  const bool stamp = q.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  int nst = 0;
#define WS_STAMP() do { if (stamp && nst < 60) q.stamps[nst++] = __builtin_readcyclecounter(); } while (0)
  WS_STAMP();

  for (int t = (int)blockIdx.x; t < q.ntiles; t += gridDim.x) {
    const int b1 = fdiv(t, q.d_tx), tx = t - b1 * q.tiles_x;
    const int n = fdiv(b1, q.d_ty), ty = b1 - n * q.tiles_y;
    const int y0 = ty * TILE_H, x0 = tx * TILE_W;
    // ---- halo tile DMA (row pieces)
    if (!(q.dbg & 1)) {
      const T* my_org = vptr32<T>(p.seg[0], n, y0 - HALO, x0 - HALO);
      if (!ONESEG && p.nseg > 1) {
        if (x_si == 1) my_org = vptr32<T>(p.seg[1], n, y0 - HALO, x0 - HALO);
        if (x_si == 2) my_org = vptr32<T>(p.seg[2], n, y0 - HALO, x0 - HALO);
        if (x_si == 3) my_org = vptr32<T>(p.seg[3], n, y0 - HALO, x0 - HALO);
      }
      dma_tile<T>(q.xt, LX, my_org + x_off, Xb, wave, max(0, HALO - y0), min(TILE_H + 2 * HALO, p.H + HALO - y0), max(0, HALO - x0), min(HW, p.W + HALO - x0));
    }
    // epilogue operands of this tile: requested now (in flight with the halo DMA), consumed after the reduction
    uint4 ea[NTC], er1[NTC], er2[NTC];
    bool ev[NTC];
    {
      const char* aux_t = (const char*)vptr32<T>(p.aux, n, y0, x0);
      const char* r1_t = (const char*)vptr32<T>(p.res1, n, y0, x0);
      const char* r2_t = (const char*)vptr32<T>(p.res2, n, y0, x0);
#pragma unroll
      for (int k = 0; k < NTC; ++k) {
        ev[k] = chv && y0 + ef[k] < p.H && x0 + ex[k] < p.W && !(q.dbg & 8);
        ea[k] = er1[k] = er2[k] = make_uint4(0, 0, 0, 0);
        if (has_aux) ea[k] = *(const uint4*)(ev[k] ? aux_t + eo_aux[k] : (const char*)g_zero16);
        if (has_r1) er1[k] = *(const uint4*)(ev[k] ? r1_t + eo_r1[k] : (const char*)g_zero16);
        if (has_r2) er2[k] = *(const uint4*)(ev[k] ? r2_t + eo_r2[k] : (const char*)g_zero16);
      }
    }
    WS_STAMP();
    __syncthreads();  // hipcc waits vmcnt(0) here: the tile has landed
    WS_STAMP();
    if (p.act != CGEN_ACT_NONE && !(q.dbg & 2)) {
      act_pieces<T>(Xb, wave, lane, xpieces, p.act);
      __syncthreads();
    }
    WS_STAMP();
    // ---- MFMAs: this wave's K-steps x all 8 tile rows
    f32x4 acc[NTC][TILE_H];
#pragma unroll
    for (int tt = 0; tt < NTC; ++tt)
#pragma unroll
      for (int f = 0; f < TILE_H; ++f) acc[tt][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(q.dbg & 4)) {
#pragma unroll
      for (int i = 0; i < NKW; ++i) {
        h16x8 bq[TILE_H];
#pragma unroll
        for (int f = 0; f < TILE_H; ++f) bq[f] = *(const h16x8*)(Xb + f * q.xt.rowbytes + koff[i]);
#pragma unroll
        for (int f = 0; f < TILE_H; ++f)
#pragma unroll
          for (int tt = 0; tt < NTC; ++tt) acc[tt][f] = mfma_h16(aw[tt][i], bq[f], acc[tt][f], 0, 0, 0);
      }
    }
    __syncthreads();  // everyone is done reading the tile: its LDS is reused for the partial sums
    WS_STAMP();
    {
      f32x4* red = (f32x4*)smem + (size_t)wave * NF * 64 + lane;
#pragma unroll
      for (int tt = 0; tt < NTC; ++tt)
#pragma unroll
        for (int f = 0; f < TILE_H; ++f) red[(tt * TILE_H + f) * 64] = acc[tt][f];
    }
    __syncthreads();
    {
      // each wave finalises 2 tile rows; a lane owns an 8-channel (16-byte) chunk of one pixel so that consecutive lanes
      // cover consecutive chunks of a pixel row (coalesced epilogue I/O).  Fixed summation order: deterministic.
      char* out_t = (char*)vptr32<T>(p.out, n, y0, x0);
#pragma unroll
      for (int k = 0; k < NTC; ++k) {
        const f32x4* rd = (const f32x4*)smem + erd[k];
        f32x4 lo = rd[0], hi = rd[16];
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) {
          const f32x4 a = rd[(size_t)wv * NF * 64], bq2 = rd[(size_t)wv * NF * 64 + 16];
          lo[0] += a[0]; lo[1] += a[1]; lo[2] += a[2]; lo[3] += a[3];
          hi[0] += bq2[0]; hi[1] += bq2[1]; hi[2] += bq2[2]; hi[3] += bq2[3];
        }
        if (ev[k]) {
          float v[8] = {lo[0] + ebias.b0.x, lo[1] + ebias.b0.y, lo[2] + ebias.b0.z, lo[3] + ebias.b0.w,
                        hi[0] + ebias.b1.x, hi[1] + ebias.b1.y, hi[2] + ebias.b1.z, hi[3] + ebias.b1.w};
          if (has_aux) {
            const uint32_t w[4] = {ea[k].x, ea[k].y, ea[k].z, ea[k].w};
            if (p.dact == CGEN_ACT_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] = h_lo(w[e]) > 0.f ? v[2 * e] : 0.f;
                v[2 * e + 1] = h_hi(w[e]) > 0.f ? v[2 * e + 1] : 0.f;
              }
            } else if (p.dact == CGEN_ACT_GELU) {
              const F8 gp = gelu8_bwd_h16(make_uint4(w[0], w[1], w[2], w[3]));
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= gp.v[e];
            }
          }
          if (has_r1) {
            const uint32_t w[4] = {er1[k].x, er1[k].y, er1[k].z, er1[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += h_lo(w[e]); v[2 * e + 1] += h_hi(w[e]); }
          }
          if (has_r2) {
            const uint32_t w[4] = {er2[k].x, er2[k].y, er2[k].z, er2[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += h_lo(w[e]); v[2 * e + 1] += h_hi(w[e]); }
          }
          uint4 o;
          o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]); o.z = f2h_pk(v[4], v[5]); o.w = f2h_pk(v[6], v[7]);
          *(uint4*)(out_t + eo_out[k]) = o;
        }
      }
    }
    __syncthreads();  // partial sums consumed before the next tile's DMA overwrites them
    WS_STAMP();
  }
  if (stamp) q.stamps[63] = nst;
#undef WS_STAMP
}

template <int NTC, int NKW, bool ONESEG>
static void launch_ws_inst2(const ConvP& p, const WsP& q, int grid_x, int grid_y, size_t lds, hipStream_t st) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)conv_ws_kernel<NTC, NKW, ONESEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL((conv_ws_kernel<NTC, NKW, ONESEG>), dim3(grid_x, grid_y), dim3(256), lds, st, p, q);
}
template <int NTC, int NKW>
static void launch_ws_inst(const ConvP& p, const WsP& q, int grid_x, int grid_y, size_t lds, hipStream_t st) {
  if (p.nseg == 1) launch_ws_inst2<NTC, NKW, true>(p, q, grid_x, grid_y, lds, st);
  else launch_ws_inst2<NTC, NKW, false>(p, q, grid_x, grid_y, lds, st);
}

static bool launch_conv_ws(const ConvP& p, hipStream_t st) {
  WsP q;
  memset(&q, 0, sizeof(q));
  const int halo = p.KS / 2;
  q.nk = ceil_div(p.taps * p.ctot8, 32);
  const int nkw = ceil_div(q.nk, 4);
  const int ntc = p.Co <= 16 ? 1 : 2;
  static const int buckets[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
  int bk = -1;
  for (int b : buckets) if (b >= nkw) { bk = b; break; }
  if (bk < 0) return false;                      // K too long for the register-resident weights: multi-pass kernel
  q.xt = mk_pixtile(p.ctot8, 2, TILE_H + 2 * halo, TILE_W + 2 * halo);
  if (q.xt.ppp < 1) return false;
  if (!p.epi_vec16 || p.Co % 8 != 0) return false;  // the kernel only has the whole-chunk epilogue
  // 32-bit address arithmetic in the kernel: every view must span less than 2^31 bytes (tile overhang included)
  for (int sg = 0; sg < p.nseg; ++sg)
    if (!fits_i32(p.seg[sg], p.N, p.H + TILE_H + 2, p.W + TILE_W + 2)) return false;
  if (!fits_i32(p.out, p.N, p.H + TILE_H, p.W + TILE_W) || !fits_i32(p.aux, p.N, p.H + TILE_H, p.W + TILE_W) ||
      !fits_i32(p.res1, p.N, p.H + TILE_H, p.W + TILE_W) || !fits_i32(p.res2, p.N, p.H + TILE_H, p.W + TILE_W)) return false;
  q.red_bytes = 4 * ntc * TILE_H * 1024;
  const size_t lds = (size_t)(q.xt.bytes > q.red_bytes ? q.xt.bytes : q.red_bytes);
  static const int ws_maxlds = [] { const char* e = getenv("CGEN_WS_MAXLDS"); return (e ? atoi(e) : 100) * 1024; }();
  if (lds > (size_t)ws_maxlds) return false;     // (<= 78 KB keeps two workgroups per CU; the 196->24 posterior-in conv at 48x48 needs 90 KB and still beats the tile kernel 47 vs 70 us)
  q.tiles_x = ceil_div(p.W, TILE_W); q.tiles_y = ceil_div(p.H, TILE_H);
  q.ntiles = p.N * q.tiles_x * q.tiles_y;
  q.rows_pad = pad_to(p.Co, 16);
  q.d_ctot8 = mk_fastdiv(p.ctot8); q.d_tx = mk_fastdiv(q.tiles_x); q.d_ty = mk_fastdiv(q.tiles_y);
  { const char* e = getenv("CGEN_WS_DBG"); q.dbg = e ? atoi(e) : 0; }
  { const char* e = getenv("CGEN_WS_STAMPS"); q.stamps = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
  const int grid_y = ceil_div(p.Co, ntc * 16);
  // measured on MI355X: the register-resident weights pay off when every wave owns >= 3 K-steps and the halo tile is
  // re-staged for at most nkw/2 output-channel tiles; short-K / wide-output (expanding) convs stay on the tile kernel
  if (nkw < 3 || (!getenv("CGEN_CONV_FORCE_WS") && nkw < 2 * grid_y)) return false;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int grid_x = 256 * per_cu / grid_y;
  if (grid_x < 1) grid_x = 1;
  if (grid_x > q.ntiles) grid_x = q.ntiles;
#define WS_CASE(NKW) case NKW: if (ntc == 1) launch_ws_inst<1, NKW>(p, q, grid_x, grid_y, lds, st); else launch_ws_inst<2, NKW>(p, q, grid_x, grid_y, lds, st); break;
  switch (bk) {
    WS_CASE(3) WS_CASE(4) WS_CASE(5) WS_CASE(6) WS_CASE(8) WS_CASE(10) WS_CASE(12) WS_CASE(16)
    default: return false;
  }
#undef WS_CASE
  return true;
}

static inline int wgrad_ntc(int co) { return co <= 16 ? 1 : (co <= 32 ? 2 : 4); }

static void wgrad_geometry(int P, int co, int ci_chunks, int ks, int& nsplit, int& pps) {
  const int ntc = wgrad_ntc(co);
  const int tiles = ci_chunks * ceil_div(co, ntc * 16) * ceil_div(ks * ks, WG_MAXT);
  int want = ceil_div(2048, tiles);
  int maxs = ceil_div(P, 4 * WG_PK);
  nsplit = want < 1 ? 1 : (want > maxs ? maxs : want);
  if (nsplit < 1) nsplit = 1;
  pps = pad_to(ceil_div(P, nsplit), WG_PK);
  nsplit = ceil_div(P, pps);
}

template <typename T>
static int launch_wgrad(const WgP& p, hipStream_t st) {
  const int ntc = wgrad_ntc(p.Co);
  dim3 grid(p.nsplit, p.n_cichunks, ceil_div(p.Co, ntc * 16) * p.n_tapgroups), block(256);
  if (ntc == 1) hipLaunchKernelGGL((wgrad_kernel<T, 1>), grid, block, 0, st, p);
  else if (ntc == 2) hipLaunchKernelGGL((wgrad_kernel<T, 2>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((wgrad_kernel<T, 4>), grid, block, 0, st, p);
  return check_launch("cgen_conv2d_wgrad");
}

// ============================================================================= multi-tensor weight prep / reduce
#define MT_CHUNK 1024  // elements per block

__global__ __launch_bounds__(256) void wprep_kernel(const cgen_wprep_desc* descs, const int* csite, const int* cidx) {
  // image row r, column k:  k = tap * ctot8 + c  with c running over the segments at 8-channel granularity
  const cgen_wprep_desc d = descs[csite[blockIdx.x]];
  const int taps = d.ks * d.ks;
  const int64_t base = (int64_t)cidx[blockIdx.x] * MT_CHUNK;
  int ctot8 = 0;
  if (d.mode == 0) { for (int s = 0; s < d.nseg; ++s) ctot8 += (d.seg_c[s] + 7) & ~7; }
  else ctot8 = (d.co + 7) & ~7;
  if (d.mode >= 8) {  // fragment-ordered images of the fused default Block (csrc/block4.hip; layout in include/cgen_hip.h)
    for (int i = threadIdx.x; i < MT_CHUNK; i += 256) {
      const int64_t o = base + i;
      if (o >= d.numel) break;
      const int frag = (int)(o >> 9), w = (int)(o & 511), ln = w >> 3, e = w & 7;
      const int r32 = ln & 31, kg = ln >> 5;
      const int q = (int)((uint32_t)frag / (uint32_t)d.k_pad), r = frag - q * d.k_pad;
      // channel (inside its 32-row block) that fragment row r32 carries.  Phase 3 (modes 12 / 13): the lane (pixel, kg) of the accumulator
      // tile ends up with 16 consecutive channels, 16 kg ..; phases 0-2 (modes 8-11): with HW = half the block's real channels (the
      // bottleneck rounded up to 8: 4, 8, 12 or 16), channels HW kg .. + HW in its first HW accumulators -- no lane idles on padding
      const int kgr = (r32 >> 2) & 1, jr = (r32 & 3) + 4 * (r32 >> 3);
      int perm = 16 * kgr + jr;
      if (d.mode <= 11) {
        const int bw = (d.mode == 8 || d.mode == 10) ? d.co : d.ci_total;  // bottleneck width
        const int mbq = d.mode <= 9 ? q : q / 9;
        const int hw = min(32, ((bw + 7) & ~7) - 32 * mbq) / 2;
        perm = jr < hw ? hw * kgr + jr : 1 << 20;
      }
      float v = 0.f;
      if (d.mode <= 9) {  // phase 0: frag = (32-row block * chunks + chunk) * 2 + step; a lane's K = 32 chunk + 16 step + 8 kg + e
        const int row = 32 * q + perm, kc = 32 * (r >> 1) + 16 * (r & 1) + 8 * kg + e;
        if (d.mode == 8) {
          if (row < d.co) {
            int cc = kc, off = 0, ci = -1;
            for (int s = 0; s < d.nseg; ++s) {  // (every segment padded to whole 32-channel chunks)
              const int c32 = (d.seg_c[s] + 31) & ~31;
              if (cc < c32) { if (cc < d.seg_c[s]) ci = off + cc; break; }
              cc -= c32; off += d.seg_c[s];
            }
            if (ci >= 0) v = d.src[(int64_t)row * d.ci_total + ci];
          }
        } else if (row < d.ci_total && kc < d.co) {
          v = d.src[(int64_t)kc * d.ci_total + row];
        }
      } else if (d.mode <= 11) {  // 3x3: frag = ((32-row block * 9 + tap) * groups + 16-channel group); k_pad = groups
        const int tap = q % 9, row = 32 * (q / 9) + perm, k = 16 * r + 8 * kg + e;
        if (d.mode == 10) { if (row < d.co && k < d.ci_total) v = d.src[((int64_t)row * d.ci_total + k) * 9 + tap]; }
        else if (row < d.ci_total && k < d.co) v = d.src[((int64_t)k * d.ci_total + row) * 9 + (8 - tap)];
      } else {  // phase 3: frag = 32-row block * groups + 16-channel group of the bottleneck; k_pad = groups
        const int row = 32 * q + perm, k = 16 * r + 8 * kg + e;
        if (d.mode == 12) { if (row < d.co && k < d.ci_total) v = d.src[(int64_t)row * d.ci_total + k]; }
        else if (row < d.seg_c[0] && k < d.co) v = d.src[(int64_t)k * d.ci_total + d.seg_off + row];
      }
      ((h16_t*)d.dst)[o] = f2h(v);
    }
    return;
  }
  if (d.mode >= 6) {  // 16-row images of the row-streaming Block instance: frag = tap * chunks + q (k_pad = chunks)
    for (int i = threadIdx.x; i < MT_CHUNK; i += 256) {
      const int64_t o = base + i;
      if (o >= d.numel) break;
      const int frag = (int)(o >> 9), w = (int)(o & 511), ln = w >> 3, e = w & 7;
      const int row = ln & 15, kc = 8 * (ln >> 4) + e;
      const int tap = (int)((uint32_t)frag / (uint32_t)d.k_pad), q = frag - tap * d.k_pad, c = 32 * q + kc;
      float v = 0.f;
      if (d.mode == 6) { if (row < d.co && c < d.ci_total) v = d.src[((int64_t)row * d.ci_total + c) * 9 + tap]; }
      else if (row < d.ci_total && c < d.co) v = d.src[((int64_t)c * d.ci_total + row) * 9 + (8 - tap)];
      ((h16_t*)d.dst)[o] = f2h(v);
    }
    return;
  }
  if (d.mode >= 2) {  // fragment-ordered images of the fused Block kernel (csrc/block.hip; layout in include/cgen_hip.h)
    for (int i = threadIdx.x; i < MT_CHUNK; i += 256) {
      const int64_t o = base + i;
      if (o >= d.numel) break;
      int frag = (int)(o >> 9);
      const int w = (int)(o & 511), ln = w >> 3, e = w & 7;
      const int r32 = ln & 31, kg = ln >> 5;
      int row = 16 * ((r32 >> 2) & 1) + (r32 & 3) + 4 * (r32 >> 3);  // channel (inside its 32-block) that fragment row r32 carries
      float v = 0.f;
      if (d.mode <= 3) {  // phase A: frag = (32-row block * chunks + chunk) * 18 + kk; k_pad = fragments per 32-row block (0: one block)
        if (d.k_pad > 0) { const int mb = (int)((uint32_t)frag / (uint32_t)d.k_pad); frag -= mb * d.k_pad; row += 32 * mb; }
        const int j = (int)((uint32_t)frag / 18u), kk = frag - j * 18, half = kk >= 9 ? 1 : 0, tap = kk - 9 * half;  // (a wave's nine fragments are contiguous)
        const int kc = 32 * j + 16 * half + 8 * kg + e;  // position on the concatenated input axis (segments in whole chunks)
        if (d.mode == 2) {
          if (row < d.co) {
            int cc = kc, off = 0, ci = -1;
            for (int s = 0; s < d.nseg; ++s) {  // (every segment padded to whole 32-channel chunks)
              const int c32 = (d.seg_c[s] + 31) & ~31;
              if (cc < c32) { if (cc < d.seg_c[s]) ci = off + cc; break; }
              cc -= c32; off += d.seg_c[s];
            }
            if (ci >= 0) v = d.src[((int64_t)row * d.ci_total + ci) * 9 + tap];
          }
        } else if (row < d.ci_total && kc < d.co) {
          v = d.src[((int64_t)kc * d.ci_total + row) * 9 + (8 - tap)];
        }
      } else {  // phase B: frag = pair * k_pad + K16-step
        const int pair = (int)((uint32_t)frag / (uint32_t)d.k_pad), ks16 = frag - pair * d.k_pad;
        const int k = 16 * ks16 + 8 * kg + e;
        const int bw = d.mode == 4 ? d.ci_total : d.co;  // bottleneck width (a multiple of 8)
        const int tap = (int)((uint32_t)k / (uint32_t)bw), c = k - tap * bw;
        const int och = pair * 32 + row;
        if (tap < 9) {
          if (d.mode == 4) { if (och < d.co) v = d.src[((int64_t)och * d.ci_total + c) * 9 + tap]; }
          else if (och < d.seg_c[0]) v = d.src[((int64_t)c * d.ci_total + d.seg_off + och) * 9 + (8 - tap)];
        }
      }
      ((h16_t*)d.dst)[o] = f2h(v);
    }
    return;
  }
  for (int i = threadIdx.x; i < MT_CHUNK; i += 256) {
    const int64_t o = base + i;
    if (o >= d.numel) break;
    // (32-bit unsigned index arithmetic: an image has < 2^31 elements -- the 64-bit division cost ~100 instructions per element
    //  and the launch was 0.2 ms at the head of every step's critical path)
    const uint32_t o32 = (uint32_t)o;
    const int r = (int)(o32 / (uint32_t)d.k_pad);
    const int k = (int)(o32 - (uint32_t)r * (uint32_t)d.k_pad);
    const int tap = (int)((uint32_t)k / (uint32_t)ctot8), c = k - tap * ctot8;
    float v = 0.f;
    if (tap < taps) {
      if (d.mode == 0) {  // forward image: r = co
        if (r < d.co) {
          int cc = c, off = 0, ci = -1;
          for (int s = 0; s < d.nseg; ++s) {
            const int c8 = (d.seg_c[s] + 7) & ~7;
            if (cc < c8) { if (cc < d.seg_c[s]) ci = off + cc; break; }
            cc -= c8; off += d.seg_c[s];
          }
          if (ci >= 0) v = d.src[((int64_t)r * d.ci_total + ci) * taps + tap];
        }
      } else {  // dgrad image of one segment: r = local ci, c = co, taps flipped
        if (r < d.seg_c[0] && c < d.co) v = d.src[((int64_t)c * d.ci_total + d.seg_off + r) * taps + (taps - 1 - tap)];
      }
    }
    if (d.dtype == CGEN_F32) ((float*)d.dst)[o] = v; else ((h16_t*)d.dst)[o] = f2h(v);
  }
}

__global__ __launch_bounds__(256) void wred_kernel(const cgen_wred_desc* descs, const int* csite, const int* cidx) {
  // Threads walk the PARTIAL layout [co][tap][ci] (what the wgrad kernels wrote), 4 consecutive ci per thread, so the
  // nsplit-deep reads are coalesced 16-byte streams; the sum is scattered into the OIHW gradient (1/nsplit of the bytes).
  const cgen_wred_desc d = descs[csite[blockIdx.x]];
  const int taps = d.ks * d.ks;
  const float us = d.unscale == 0.f ? 1.f : d.unscale;  // 1 / loss scale of the f16 engine (a power of two: exact)
  const int nw = d.co * d.ci_total * taps;  // < 2^31 (checked on the host)
  const int s0 = cidx[blockIdx.x] * MT_CHUNK + threadIdx.x * 4;
  if (s0 >= d.numel) return;
  if (s0 < nw) {
    const bool vec = (nw & 3) == 0 && (((uintptr_t)d.partial_w) & 15) == 0;  // (16-byte loads of 4 consecutive partial elements: any layout)
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const int cnt = min(4, nw - s0);
    if (vec) {
      const float4* src = (const float4*)(d.partial_w + s0);
      const size_t stride = (size_t)nw / 4;
      // four independent chains of streaming (non-temporal) 16-byte loads: the partials are read exactly once; the sum order
      // ((c0 + c1) + (c2 + c3)) over splits dealt round-robin is fixed => deterministic
      float4 b[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      int sp = 0;
      // sixteen loads in flight per thread first: a thread walks its element's splits alone, and the streaming weight-gradient kernel
      // leaves up to 144 of them (192^2 layers) -- at four per round trip the final reduce was 36 dependent HBM latencies long
      // (0.15 ms at the end of every step for 108 MB).  The order of the additions is the one below: bit-identical sums.
      for (; sp + 16 <= d.nsplit; sp += 16) {
        f32x4 t[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) t[c] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(sp + c) * stride));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) { b[c].x += t[4 * u + c][0]; b[c].y += t[4 * u + c][1]; b[c].z += t[4 * u + c][2]; b[c].w += t[4 * u + c][3]; }
      }
      for (; sp + 4 <= d.nsplit; sp += 4) {
        float4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 t = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(sp + c) * stride));  // one global_load_dwordx4 nt
          v[c] = make_float4(t[0], t[1], t[2], t[3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { b[c].x += v[c].x; b[c].y += v[c].y; b[c].z += v[c].z; b[c].w += v[c].w; }
      }
      for (int c = 0; sp < d.nsplit; ++sp, ++c) {
        const float4 v0 = src[(size_t)sp * stride];
        b[c].x += v0.x; b[c].y += v0.y; b[c].z += v0.z; b[c].w += v0.w;
      }
      a[0] = (b[0].x + b[1].x) + (b[2].x + b[3].x); a[1] = (b[0].y + b[1].y) + (b[2].y + b[3].y);
      a[2] = (b[0].z + b[1].z) + (b[2].z + b[3].z); a[3] = (b[0].w + b[1].w) + (b[2].w + b[3].w);
    } else {
      for (int e = 0; e < cnt; ++e) {
        float acc = 0.f;
        for (int sp = 0; sp < d.nsplit; ++sp) acc += d.partial_w[(size_t)sp * nw + s0 + e];
        a[e] = acc;
      }
    }
    for (int e = 0; e < cnt; ++e) {
      const unsigned sidx = (unsigned)(s0 + e);
      size_t o;
      if (d.layout == 1) {  // [ci][flipped tap][co] (the streaming kernel with grad_out as its shifted operand, csrc/wgrad3.hip)
        const unsigned r = sidx / (unsigned)d.co, co = sidx - r * d.co;
        const unsigned ci = r / (unsigned)taps, ft = r - ci * taps;
        o = ((size_t)co * d.ci_total + ci) * taps + (taps - 1 - ft);
      } else {
        const unsigned r = sidx / (unsigned)d.ci_total, ci = sidx - r * d.ci_total;
        const unsigned co = r / (unsigned)taps, tap = r - co * taps;
        o = ((size_t)co * d.ci_total + ci) * taps + tap;
      }
#ifdef WRED_ABL_NOSCATTER  // (ablation build, wrong results: the sums go out in partial order, coalesced)
      o = sidx;
#endif
      d.grad_w[o] = d.accumulate ? d.grad_w[o] + a[e] * us : a[e] * us;
    }
  }
  if (d.grad_b && s0 + 3 >= nw) {  // elements nw .. numel-1 are the bias gradient: this thread's (up to four) sums run side by side --
    // one after the other they were 4 x nsplit / 16 dependent round trips on the one thread that owns them, the end of the launch
    int co4[4];
    bool ok4[4];
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int o = s0 + e; ok4[e] = o >= nw && o < d.numel; co4[e] = ok4[e] ? o - nw : 0; }
    int sp = 0;
    for (; sp + 16 <= d.nsplit; sp += 16) {  // (loads first, additions in split order: the same sums as a plain loop)
      float t[4][16];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 16; ++c) t[e][c] = d.partial_b[(size_t)(sp + c) * d.co + co4[e]];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc4[e] += t[e][c];
    }
    for (; sp < d.nsplit; ++sp)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc4[e] += d.partial_b[(size_t)sp * d.co + co4[e]];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ok4[e]) d.grad_b[co4[e]] = d.accumulate ? d.grad_b[co4[e]] + acc4[e] * us : acc4[e] * us;
  }
}

}  // namespace cgen

using namespace cgen;

static int conv_fill(const cgen_conv_args* a, ConvP& p) {
  CGEN_REQUIRE(a, "cgen_conv2d: null args");
  CGEN_REQUIRE(a->dtype == CGEN_F32 || a->dtype == CGEN_F16 || a->dtype == CGEN_F32S, "cgen_conv2d: bad dtype %d", a->dtype);
  CGEN_REQUIRE(a->ks == 1 || a->ks == 3 || a->ks == 5 || a->ks == 7, "cgen_conv2d: kernel size %d unsupported", a->ks);
  CGEN_REQUIRE(a->nseg >= 1 && a->nseg <= CGEN_MAX_SEG, "cgen_conv2d: nseg %d", a->nseg);
  CGEN_REQUIRE(a->n > 0 && a->h > 0 && a->w > 0 && a->out.c > 0 && a->out.p && a->weight, "cgen_conv2d: bad shape/pointers");
  CGEN_REQUIRE((int64_t)a->n * a->h * a->w < (1ll << 31), "cgen_conv2d: too many pixels");
  const int esz = a->dtype == CGEN_F16 ? 2 : 4;
  memset(&p, 0, sizeof(p));
  p.split = a->dtype == CGEN_F32S ? 1 : 0;
  p.N = a->n; p.H = a->h; p.W = a->w; p.KS = a->ks; p.pad = a->ks / 2; p.nseg = a->nseg; p.act = a->act; p.dact = a->dact;
  p.Co = a->out.c; p.P = a->n * a->h * a->w; p.taps = a->ks * a->ks;
  int koff = 0;
  for (int s = 0; s < a->nseg; ++s) {
    CGEN_REQUIRE(a->seg[s].p && a->seg[s].c > 0, "cgen_conv2d: segment %d empty", s);
    p.seg[s] = mk(a->seg[s]);
    p.seg_koff[s] = koff;
    p.seg_vec[s] = vec16_ok(a->seg[s], esz);
    koff += pad_to(a->seg[s].c, 8);
  }
  for (int s = a->nseg; s < CGEN_MAX_SEG; ++s) p.seg_koff[s] = 1 << 30;
  p.tap0 = 0; p.tap1 = p.taps;
  if (a->h == 1 && a->w == 1 && a->ks > 1) { p.tap0 = p.taps / 2; p.tap1 = p.tap0 + 1; }
  p.dma_ok = 1;
  for (int s = 0; s < a->nseg; ++s) p.dma_ok = p.dma_ok && dma_clean(a->seg[s], esz);
  p.ctot8 = koff;
  p.krow = pad_to(p.taps * koff, 32) + 32;
  p.w = a->weight; p.bias = a->bias;
  CGEN_REQUIRE(((uintptr_t)a->weight) % 16 == 0, "cgen_conv2d: weight image must be 16-byte aligned");
  p.out = mk(a->out); p.aux = mk(a->aux); p.res1 = mk(a->res1); p.res2 = mk(a->res2);
  CGEN_REQUIRE((!a->out_rem && !a->res1_rem) || a->dtype == CGEN_F16, "cgen_conv2d: remainder planes are an f16 feature");
  CGEN_REQUIRE(a->out_rem % 16 == 0 && a->res1_rem % 16 == 0, "cgen_conv2d: remainder-plane offsets must be multiples of 16 bytes");
  p.out_rem = a->out_rem; p.r1_rem = a->res1.p ? a->res1_rem : 0;
  if (!a->dact) p.aux.p = nullptr;
  // 4-channel vector epilogue: needs 4*esz-byte alignment on every view it touches
  auto epi_ok = [&](const cgen_view& v) {
    if (!v.p) return true;
    const int q = 4 * esz;
    return ((uintptr_t)v.p % q == 0) && ((v.sn * esz) % q == 0) && ((v.sh * esz) % q == 0) && ((v.sw * esz) % q == 0);
  };
  p.force_generic = getenv("CGEN_CONV_GENERIC") != nullptr;
  p.epi_vec = epi_ok(a->out) && epi_ok(a->aux) && epi_ok(a->res1) && epi_ok(a->res2) && (!a->bias || ((uintptr_t)a->bias % 16 == 0));
  p.epi_vec16 = vec16_ok(a->out, esz) && vec16_ok(a->aux, esz) && vec16_ok(a->res1, esz) && vec16_ok(a->res2, esz) && (!a->bias || ((uintptr_t)a->bias % 16 == 0));
  return CGEN_OK;
}

extern "C" int cgen_conv2d(const cgen_conv_args* a, cgen_stream_t stream) {
  ConvP p;
  const int rc = conv_fill(a, p);
  if (rc != CGEN_OK) return rc;
  return a->dtype != CGEN_F16 ? launch_conv<float>(p, (hipStream_t)stream) : launch_conv<h16_t>(p, (hipStream_t)stream);
}

// ---- pair launch of two small-image convs (conv_smallp_pair_kernel)
static bool smallp_pair_fill(const cgen_conv_args* a, SpArgs& s) {
  if (!a || a->dtype != CGEN_F16 || conv_fill(a, s.p) != CGEN_OK) return false;
  const ConvP& p = s.p;
  if (!smallp_takes(p)) return false;
  s.nks = ceil_div((p.tap1 - p.tap0) * p.ctot8, 32);
  s.gx = ceil_div(p.P, 32); s.gy = ceil_div(p.Co, 2 * 16); s.pad = 0;
  s.d_ctot8 = mk_fastdiv(p.ctot8); s.d_hw = mk_fastdiv(p.H * p.W); s.d_w = mk_fastdiv(p.W);
  return true;
}

extern "C" int cgen_conv2d_pair_supported(const cgen_conv_args* a, const cgen_conv_args* b) {
  SpArgs sa, sb;
  if (!smallp_pair_fill(a, sa) || !smallp_pair_fill(b, sb)) return 0;
  return sa.p.KS == sb.p.KS && (sa.p.nseg == 1) == (sb.p.nseg == 1) ? 1 : 0;
}

extern "C" int cgen_conv2d_pair(const cgen_conv_args* a, const cgen_conv_args* b, cgen_stream_t stream) {
  SpArgs sa, sb;
  CGEN_REQUIRE(smallp_pair_fill(a, sa) && smallp_pair_fill(b, sb) && sa.p.KS == sb.p.KS && (sa.p.nseg == 1) == (sb.p.nseg == 1),
               "cgen_conv2d_pair: the two convs are not served by one small-image instance (ask cgen_conv2d_pair_supported first)");
  const dim3 grid(sa.gx + sb.gx, std::max(sa.gy, sb.gy));
  hipStream_t st = (hipStream_t)stream;
#define SPP_LAUNCH(KS_) do { if (sa.p.nseg == 1) hipLaunchKernelGGL((conv_smallp_pair_kernel<KS_, 2, 4, true>), grid, dim3(256), 0, st, sa, sb); \
    else hipLaunchKernelGGL((conv_smallp_pair_kernel<KS_, 2, 4, false>), grid, dim3(256), 0, st, sa, sb); } while (0)
  if (sa.p.KS == 1) SPP_LAUNCH(1); else SPP_LAUNCH(3);
#undef SPP_LAUNCH
  conv_trace(sa.p, "smlp2");
  return check_launch("cgen_conv2d_pair");
}

static int count_chunks(const cgen_view* seg, int nseg) {
  int c = 0;
  for (int s = 0; s < nseg; ++s) c += (seg[s].c + 31) / 32;
  return c;
}

static int ctot8_of(const int32_t* seg_c, int nseg) {
  int c = 0;
  for (int s = 0; s < nseg; ++s) c += pad_to(seg_c[s], 8);
  return c;
}

// can the streaming tiled kernel serve this call?
static bool wgrad_tiled_ok(const cgen_wgrad_args* a, Wg2Geom& g) {
  if (a->dtype != CGEN_F16 || getenv("CGEN_WGRAD_GENERIC")) return false;
  if (!dma_clean(a->gout, 2)) return false;
  int segc[CGEN_MAX_SEG];
  for (int s = 0; s < a->nseg; ++s) {
    if (!dma_clean(a->seg[s], 2)) return false;
    segc[s] = a->seg[s].c;
  }
  // the kernel does its address arithmetic in 32 bits: every view must span less than 2^31 bytes (tile overhang included)
  auto fits = [&](const cgen_view& v) {
    const int64_t ext = (int64_t)a->n * v.sn + (int64_t)(a->h + TILE_H + a->ks) * v.sh + (int64_t)(a->w + TILE_W + a->ks) * v.sw + v.c;
    return ext * 2 < ((int64_t)1 << 31);
  };
  if (!fits(a->gout)) return false;
  for (int s = 0; s < a->nseg; ++s)
    if (!fits(a->seg[s])) return false;
  return wgrad2_geometry(a->n, a->h, a->w, a->gout.c, ctot8_of(segc, a->nseg), a->ks, g);
}

extern "C" int cgen_conv2d_wgrad_plan(const cgen_wgrad_args* a, int32_t* tiled_out) {
  if (!a || a->nseg < 1 || a->nseg > CGEN_MAX_SEG) return 0;
  {
    Wg3Plan g3;
    if (wg3_plan(a, g3)) {
      if (tiled_out) *tiled_out = 2 + g3.q.layout;
      return g3.nsplit_total;
    }
  }
  Wg2Geom g;
  if (wgrad_tiled_ok(a, g)) {
    if (tiled_out) *tiled_out = 1;
    return g.nsplit;
  }
  if (tiled_out) *tiled_out = 0;
  int ci_total = 0;
  for (int s = 0; s < a->nseg; ++s) ci_total += a->seg[s].c;
  int nsplit, pps;
  wgrad_geometry(a->n * a->h * a->w, a->gout.c, (ci_total + 31) / 32, a->ks, nsplit, pps);
  return nsplit;
}

// Wg2P + geometry of one problem for the tiled bf16 kernel (false: not eligible, the caller uses the generic kernel)
static bool build_wg2(const cgen_wgrad_args* a, Wg2P& q, Wg2Geom& g) {
  if (!wgrad_tiled_ok(a, g)) return false;
  memset(&q, 0, sizeof(q));
  q.N = a->n; q.H = a->h; q.W = a->w; q.KS = a->ks; q.nseg = a->nseg; q.act = a->act; q.Co = a->gout.c; q.taps = a->ks * a->ks;
  int k8 = 0, o2 = 0;
  for (int s = 0; s < a->nseg; ++s) {
    q.seg[s] = mk(a->seg[s]); q.seg_koff[s] = k8; q.seg_off[s] = o2;
    k8 += pad_to(a->seg[s].c, 8); o2 += a->seg[s].c;
  }
  for (int s = a->nseg; s < CGEN_MAX_SEG; ++s) q.seg_koff[s] = 1 << 30;
  q.ci_total = o2;
  q.ctot8 = k8;
  q.gout = mk(a->gout);
  q.pw = a->partial_w; q.pb = a->partial_b;
  q.tiles_x = g.tiles_x; q.tiles_y = g.tiles_y; q.ntiles = g.ntiles; q.nsplit = g.nsplit; q.tiles_per_split = g.tps;
  q.cwin = g.cwin; q.cog = g.ncf * 16; q.xt = g.xt; q.gt = g.gt;
  q.d_tx = mk_fastdiv(g.tiles_x); q.d_ty = mk_fastdiv(g.tiles_y);
  q.variant = a->ks == 7 ? 32 + g.ncf : (g.ncf == 2 && g.njw == 16 ? 36 + (a->ks == 3 ? 1 : 0) : g.ncf * 2 + (a->ks == 3 ? 1 : 0));
  q.dbuf = g.dbuf;
  { const char* e = getenv("CGEN_WG2_DBG"); q.dbg = e ? atoi(e) : 0; }
  return true;
}

template <int NCF, int NJW>
static void launch_wgrad2_batched(int ks, const Wg2P* probs, const int4* blocks, int nblocks, int grid, size_t lds, hipStream_t st) {
  if (ks == 3) {
    static bool once3 = false;
    if (!once3) { (void)hipFuncSetAttribute((const void*)wgrad_tile_batched_kernel<NCF, NJW, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once3 = true; }
    hipLaunchKernelGGL((wgrad_tile_batched_kernel<NCF, NJW, 3>), dim3(grid), dim3(256), lds, st, probs, blocks, nblocks);
  } else {
    static bool once1 = false;
    if (!once1) { (void)hipFuncSetAttribute((const void*)wgrad_tile_batched_kernel<NCF, NJW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once1 = true; }
    hipLaunchKernelGGL((wgrad_tile_batched_kernel<NCF, NJW, 1>), dim3(grid), dim3(256), lds, st, probs, blocks, nblocks);
  }
}

extern "C" int cgen_conv2d_wgrad_batch_plan(const cgen_wgrad_args* args, int32_t count, void* blob_host, int64_t capacity, int64_t* blob_bytes,
                                            cgen_wgrad_batch_launch* launches, int32_t max_launches, int32_t* n_launches, int32_t* eligible) {
  CGEN_REQUIRE(args && count >= 0 && blob_bytes && n_launches && eligible, "cgen_conv2d_wgrad_batch_plan: null args");
  struct Item { int idx; Wg2P q; Wg2Geom g; int key; long cost; };
  std::vector<Item> items;
  struct Item3 { int idx; Wg3Plan g; };
  std::vector<Item3> items3;  // problems of the streaming kernel (csrc/wgrad3.hip): their own launch, table and block list
  for (int i = 0; i < count; ++i) {
    Item it;
    it.idx = i;
    eligible[i] = 0;
    if (args[i].dtype != CGEN_F16 || !args[i].partial_w) continue;
    {
      Item3 i3;
      i3.idx = i;
      if (wg3_plan(&args[i], i3.g)) {
        if (i3.g.nsplit_total != args[i].nsplit) continue;
        eligible[i] = 1;
        items3.push_back(i3);
        continue;
      }
    }
    if (!build_wg2(&args[i], it.q, it.g)) continue;
    if (it.g.nsplit != args[i].nsplit) continue;
    eligible[i] = 1;
    // one launch per (variant, kernel size, LDS class): a big-LDS problem must not lower everyone's occupancy
    it.key = (it.g.ncf * 4 + (args[i].ks == 3 ? 1 : 0)) * 2 + (it.g.lds > 40 * 1024 ? 1 : 0);
    it.cost = (long)it.g.nsplit * it.g.n_cwin * it.g.n_co * it.g.tps;
    if (blob_host && getenv("CGEN_WG2_PLAN_DEBUG"))
      fprintf(stderr, "wg2 plan: %dx%dx%d ks%d ctot8 %3d co %3d | ncf %d cwin %3d n_cwin %d n_co %d | tiles %5d tps %3d nsplit %3d | lds %6zu dbuf %d | tile-visits %ld\n",
              args[i].n, args[i].h, args[i].w, args[i].ks, it.q.ctot8, it.q.Co, it.g.ncf, it.g.cwin, it.g.n_cwin, it.g.n_co, it.g.ntiles, it.g.tps,
              it.g.nsplit, it.g.lds, it.g.dbuf, (long)it.g.ntiles * it.g.n_cwin * it.g.n_co);
    items.push_back(it);
  }
  static const bool mega = [] { const char* e = getenv("CGEN_WGRAD_MEGA"); return !e || atoi(e) != 0; }();
  if (!mega) {  // the per-variant launches have no 7x7 instance: the stem goes to the caller's single launch
    std::vector<Item> keep;
    for (auto& it : items) { if (args[it.idx].ks == 7 || (it.g.ncf == 2 && it.g.njw == 16)) eligible[it.idx] = 0; else keep.push_back(it); }
    items.swap(keep);
  }
  if (mega) {  // one launch for everything: longest blocks first (the resident workgroups take them round robin)
    for (auto& it : items) {
      it.key = 0;
      it.cost = (long)it.g.tps * it.g.ncf * std::min(it.g.cwin, it.q.ctot8) * it.q.taps;
      // (a block's length is its DMA bytes, not its FLOPs: 1933 -> 1940 img/s in three interleaved pairs)
      static const int by_bytes = [] { const char* e = getenv("CGEN_WG2_SORT_BYTES"); return e ? atoi(e) : 1; }();
      if (by_bytes) it.cost = (long)it.g.tps * (long)(it.g.xt.bytes + it.g.gt.bytes);
    }
  }
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key != b.key ? a.key < b.key : a.cost > b.cost; });
  const int64_t probs_bytes = pad_to((int)(items.size() * sizeof(Wg2P)), 256);
  int64_t nblocks_total = 0;
  for (auto& it : items) nblocks_total += (int64_t)it.g.nsplit * it.g.n_cwin * it.g.n_co;
  const int64_t old_bytes = pad_to((int)(probs_bytes + nblocks_total * (int64_t)sizeof(int4)), 256);
  // streaming-kernel section: [Wg3P table][one int4 per workgroup], longest blocks first
  std::stable_sort(items3.begin(), items3.end(), [](const Item3& a, const Item3& b) { return a.g.block_bytes > b.g.block_bytes; });
  const int64_t probs3_bytes = pad_to((int)(items3.size() * sizeof(Wg3P)), 256);
  int64_t nblocks3 = 0;
  for (auto& it : items3) nblocks3 += it.g.nblocks;
  *blob_bytes = old_bytes + (items3.empty() ? 0 : probs3_bytes + nblocks3 * (int64_t)sizeof(int4));
  int nl = 0;
  for (size_t i = 0; i < items.size(); ++i)
    if (i == 0 || items[i].key != items[i - 1].key) ++nl;
  if (!items3.empty()) ++nl;
  *n_launches = nl;
  if (!blob_host) return CGEN_OK;  // size query
  CGEN_REQUIRE(capacity >= *blob_bytes && launches && max_launches >= nl, "cgen_conv2d_wgrad_batch_plan: buffers too small");
  memset(blob_host, 0, (size_t)*blob_bytes);
  Wg2P* probs = (Wg2P*)blob_host;
  int4* blocks = (int4*)((char*)blob_host + probs_bytes);
  int64_t b = 0;
  int li = -1;
  for (size_t i = 0; i < items.size(); ++i) {
    const Item& it = items[i];
    probs[i] = it.q;
    if (i == 0 || it.key != items[i - 1].key) {
      ++li;
      launches[li].ncf = mega ? 0 : it.g.ncf; launches[li].ks = args[it.idx].ks; launches[li].lds_bytes = 0; launches[li].nblocks = 0;
      launches[li].blocks_offset = probs_bytes + b * (int64_t)sizeof(int4);
    }
    if ((int32_t)it.g.lds > launches[li].lds_bytes) launches[li].lds_bytes = (int32_t)it.g.lds;
    for (int z = 0; z < it.g.n_co; ++z)
      for (int y = 0; y < it.g.n_cwin; ++y)
        for (int x = 0; x < it.g.nsplit; ++x) { blocks[b] = make_int4((int)i, x, y, z); ++b; }
    launches[li].nblocks += it.g.nsplit * it.g.n_cwin * it.g.n_co;
  }
  if (!items3.empty()) {
    // The block list interleaves the problems: block k of every problem before block k + 1 of any (each problem's blocks are
    // equally long; problems sorted longest first), so the resident workgroups start on the long problems and the tail is short.
    Wg3P* probs3 = (Wg3P*)((char*)blob_host + old_bytes);
    int4* blocks3 = (int4*)((char*)blob_host + old_bytes + probs3_bytes);
    ++li;
    launches[li].ncf = -3; launches[li].ks = (int32_t)items3.size(); launches[li].lds_bytes = 0; launches[li].nblocks = (int32_t)nblocks3;
    launches[li].blocks_offset = old_bytes + probs3_bytes;
    int64_t b3 = 0;
    int maxb = 0;
    for (size_t i = 0; i < items3.size(); ++i) {
      probs3[i] = items3[i].g.q;
      probs3[i].pw = args[items3[i].idx].partial_w; probs3[i].pb = args[items3[i].idx].partial_b;
      if ((int32_t)items3[i].g.lds > launches[li].lds_bytes) launches[li].lds_bytes = (int32_t)items3[i].g.lds;
      maxb = std::max(maxb, items3[i].g.nblocks);
    }
    // Block order = start order (the final flush launches one workgroup per block, the background flush walks the list with a
    // resident set): LONGEST FIRST over all blocks, by estimated duration -- bytes over the rate the problem's image side streams at
    // (tools/bench_wgrad3.py batch: >= 96^2 ~4 TB/s, 48^2 ~3, 24^2 ~1.4, 12^2 and below ~0.9).  The first version dealt the
    // problems round-robin (block k of every problem, k = 0, 1, ...): the list then ENDED with blocks 30..63 of the dozen 192^2
    // problems alone, ~300 of the longest blocks on a half-empty chip, and the launch took 2.5 ms whatever the kernel's speed.
    if (getenv("CGEN_WG3_ORDER") && atoi(getenv("CGEN_WG3_ORDER")) == 0) {
      for (int k = 0; k < maxb; ++k)
        for (size_t i = 0; i < items3.size(); ++i) {
          const Wg3P& q3 = items3[i].g.q;
          if (k < items3[i].g.nblocks) { const int w = k / q3.nsplit; blocks3[b3] = make_int4((int)i, k % q3.nsplit, w % q3.n_pwin, w / q3.n_pwin); ++b3; }
        }
    } else {
      struct Ord { double cost; int i, k; };
      std::vector<Ord> ord;
      ord.reserve((size_t)nblocks3);
      for (size_t i = 0; i < items3.size(); ++i) {
        const Wg3P& q3 = items3[i].g.q;
        const int side = std::min(q3.H, q3.W);
        const double slow = side >= 96 ? 1.0 : side >= 48 ? 1.35 : side >= 24 ? 2.9 : 4.5;
        // the last split of a problem may be short: tiles of split s
        for (int k = 0; k < items3[i].g.nblocks; ++k) {
          const int sp = k % q3.nsplit;
          const int nt = std::min(q3.tps, q3.ntiles - sp * q3.tps);
          ord.push_back({(double)items3[i].g.block_bytes * slow * (double)nt / (double)q3.tps, (int)i, k});
        }
      }
      std::stable_sort(ord.begin(), ord.end(), [](const Ord& a, const Ord& b) { return a.cost > b.cost; });
      for (const Ord& o : ord) {
        const Wg3P& q3 = items3[o.i].g.q;
        const int w = o.k / q3.nsplit;
        blocks3[b3] = make_int4(o.i, o.k % q3.nsplit, w % q3.n_pwin, w / q3.n_pwin); ++b3;
      }
    }
  }
  return CGEN_OK;
}

extern "C" int cgen_conv2d_wgrad_batch_run(const void* blob_dev, const cgen_wgrad_batch_launch* launches, int32_t n_launches, int32_t max_workgroups,
                                           cgen_stream_t stream) {
  CGEN_REQUIRE(blob_dev && (launches || n_launches == 0) && n_launches >= 0, "cgen_conv2d_wgrad_batch_run: null args");
  hipStream_t st = (hipStream_t)stream;
  const Wg2P* probs = (const Wg2P*)blob_dev;
  for (int i = 0; i < n_launches; ++i) {
    const cgen_wgrad_batch_launch& l = launches[i];
    if (l.nblocks <= 0) continue;
    const int4* blocks = (const int4*)((const char*)blob_dev + l.blocks_offset);
    const int grid = max_workgroups > 0 ? std::min(l.nblocks, max_workgroups) : l.nblocks;
    switch (l.ncf) {
      case -3: {  // streaming kernel: l.ks problems, their table right in front of the block list
        const int64_t probs3_bytes = pad_to((int)(l.ks * sizeof(Wg3P)), 256);
        wg3_launch_mega((const Wg3P*)((const char*)blob_dev + l.blocks_offset - probs3_bytes), blocks, l.nblocks, grid, (size_t)l.lds_bytes, st);
        break;
      }
      case 0: {
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute((const void*)wgrad_tile_mega_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
        hipLaunchKernelGGL(wgrad_tile_mega_kernel, dim3(grid), dim3(256), (size_t)l.lds_bytes, st, probs, blocks, l.nblocks);
        break;
      }
      case 1: launch_wgrad2_batched<1, 16>(l.ks, probs, blocks, l.nblocks, grid, (size_t)l.lds_bytes, st); break;
      case 2: launch_wgrad2_batched<2, 12>(l.ks, probs, blocks, l.nblocks, grid, (size_t)l.lds_bytes, st); break;
      case 4: launch_wgrad2_batched<4, 6>(l.ks, probs, blocks, l.nblocks, grid, (size_t)l.lds_bytes, st); break;
      case 6: launch_wgrad2_batched<6, 4>(l.ks, probs, blocks, l.nblocks, grid, (size_t)l.lds_bytes, st); break;
      case 8: launch_wgrad2_batched<8, 3>(l.ks, probs, blocks, l.nblocks, grid, (size_t)l.lds_bytes, st); break;
      default: return cgen::fail(CGEN_EINVAL, "cgen_conv2d_wgrad_batch_run: bad variant %d", l.ncf);
    }
  }
  return check_launch("cgen_conv2d_wgrad_batch_run");
}

extern "C" int cgen_conv2d_wgrad(const cgen_wgrad_args* a, cgen_stream_t stream) {
  CGEN_REQUIRE(a && a->partial_w, "cgen_conv2d_wgrad: null args");
  CGEN_REQUIRE(a->dtype == CGEN_F32 || a->dtype == CGEN_F16, "cgen_conv2d_wgrad: bad dtype");
  CGEN_REQUIRE(a->nseg >= 1 && a->nseg <= CGEN_MAX_SEG && a->gout.p && a->gout.c > 0, "cgen_conv2d_wgrad: bad args");
  const int esz = a->dtype == CGEN_F32 ? 4 : 2;
  WgP p;
  memset(&p, 0, sizeof(p));
  p.N = a->n; p.H = a->h; p.W = a->w; p.KS = a->ks; p.pad = a->ks / 2; p.nseg = a->nseg; p.act = a->act;
  p.Co = a->gout.c; p.P = a->n * a->h * a->w; p.taps = a->ks * a->ks;
  int off = 0, ch = 0;
  for (int s = 0; s < a->nseg; ++s) {
    p.seg[s] = mk(a->seg[s]);
    p.seg_off[s] = off;
    p.seg_vec[s] = vec16_ok(a->seg[s], esz);
    p.seg_chunk0[s] = ch;
    off += a->seg[s].c;
    ch += (a->seg[s].c + 31) / 32;
  }
  p.seg_chunk0[a->nseg] = ch;
  p.ci_total = off;
  p.n_cichunks = ch;
  p.n_tapgroups = ceil_div(p.taps, WG_MAXT);
  {  // streaming kernel of round 5 (csrc/wgrad3.hip) where it serves the shape
    Wg3Plan g3;
    if (wg3_plan(a, g3)) {
      CGEN_REQUIRE(g3.nsplit_total == a->nsplit, "cgen_conv2d_wgrad: nsplit %d != expected %d", a->nsplit, g3.nsplit_total);
      wg3_launch_single(g3, (hipStream_t)stream);
      return check_launch("cgen_conv2d_wgrad(stream)");
    }
  }
  {  // tiled bf16 kernel when the shape allows it
    int segc[CGEN_MAX_SEG];
    for (int s = 0; s < a->nseg; ++s) segc[s] = a->seg[s].c;
    Wg2Geom g;
    Wg2P q;
    if (build_wg2(a, q, g)) {
      CGEN_REQUIRE(g.nsplit == a->nsplit, "cgen_conv2d_wgrad: nsplit %d != expected %d", a->nsplit, g.nsplit);
      { const char* e = getenv("CGEN_WG2_DBG"); q.dbg = e ? atoi(e) : 0; }
      { const char* e = getenv("CGEN_WG2_STAMPS"); q.stamps = e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }
      hipStream_t st = (hipStream_t)stream;
      switch (g.ncf) {
        case 1: launch_wgrad2_ks<1, 16>(q, g, st); break;
        case 2: if (a->ks == 7 || g.njw == 16) launch_wgrad2_ks<2, 16>(q, g, st); else launch_wgrad2_ks<2, 12>(q, g, st); break;
        case 4: launch_wgrad2_ks<4, 6>(q, g, st); break;
        case 6: launch_wgrad2_ks<6, 4>(q, g, st); break;
        default: launch_wgrad2_ks<8, 3>(q, g, st); break;
      }
      return check_launch("cgen_conv2d_wgrad(tile)");
    }
  }
  // geometry must match what the caller sized the partial buffer with
  int ns, pps;
  wgrad_geometry(p.P, p.Co, (p.ci_total + 31) / 32, p.KS, ns, pps);
  CGEN_REQUIRE(ns == a->nsplit, "cgen_conv2d_wgrad: nsplit %d != expected %d", a->nsplit, ns);
  p.nsplit = ns; p.pix_per_split = pps;
  p.gout = mk(a->gout);
  {
    const int q = 4 * esz;
    p.gout_vec = ((uintptr_t)a->gout.p % q == 0) && ((a->gout.sn * esz) % q == 0) && ((a->gout.sh * esz) % q == 0) && ((a->gout.sw * esz) % q == 0);
  }
  p.pw = a->partial_w; p.pb = a->partial_b;
  return a->dtype == CGEN_F32 ? launch_wgrad<float>(p, (hipStream_t)stream) : launch_wgrad<h16_t>(p, (hipStream_t)stream);
}

extern "C" int cgen_weight_prep(const cgen_wprep_desc* descs, const int32_t* csite, const int32_t* cidx, int32_t nchunks,
                                cgen_stream_t stream) {
  if (nchunks <= 0) return CGEN_OK;
  CGEN_REQUIRE(descs && csite && cidx, "cgen_weight_prep: null table");
  hipLaunchKernelGGL(wprep_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, descs, csite, cidx);
  return check_launch("cgen_weight_prep");
}

extern "C" int cgen_wgrad_reduce(const cgen_wred_desc* descs, const int32_t* csite, const int32_t* cidx, int32_t nchunks,
                                 cgen_stream_t stream) {
  if (nchunks <= 0) return CGEN_OK;
  CGEN_REQUIRE(descs && csite && cidx, "cgen_wgrad_reduce: null table");
  hipLaunchKernelGGL(wred_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, descs, csite, cidx);
  return check_launch("cgen_wgrad_reduce");
}
