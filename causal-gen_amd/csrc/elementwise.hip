// Data-movement and element-wise kernels on NHWC views (HBM-bound; one pass, 4-channel vector access when the
// views allow it).  See include/cgen_hip.h for the reference lines each entry point replaces.
#include "common.h"

namespace cgen {

}  // namespace cgen
#include "elementwise_bodies.inc"
namespace cgen {

template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(Shape4 so, int d, View in, View out) {
  avgpool_fwd_body<T, V>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, so, d, in, out);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(Shape4 si, int d, View gout, View gin, int accumulate) {
  avgpool_bwd_body<T, V>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, si, d, gout, gin, accumulate);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(Shape4 so, int hi, int wi, float ish, float isw, View in,
                                                           const float* bias, View out) {
  upsample_fwd_body<T, V>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, so, hi, wi, ish, isw, in, bias, out);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(Shape4 si, int ho, int wo, float ish, float isw, View gout,
                                                           View gin, int accumulate) {
  upsample_bwd_body<T, V>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, si, ho, wo, ish, isw, gout, gin, accumulate);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void batch_broadcast_kernel(Shape4 s, const float* src, View out) {
  batch_broadcast_body<T, V>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, s, src, out);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void axpby_kernel(Shape4 s, View in, View out, float alpha, float beta, int c_from, int accumulate) {
  axpby_body<T, V>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, s, in, out, alpha, beta, c_from, accumulate);
}

// aten::adaptive_avg_pool2d (vae.py:79-81, Block with a float down-rate): window of output cell o along an axis of length
// `in` -> `out` is [floor(o * in / out), ceil((o + 1) * in / out))
__device__ __forceinline__ int ap_start(int o, int in, int out) { return (int)(((int64_t)o * in) / out); }
__device__ __forceinline__ int ap_end(int o, int in, int out) { return (int)((((int64_t)(o + 1)) * in + out - 1) / out); }

template <typename T, int V>
__global__ __launch_bounds__(256) void adaptive_avgpool_fwd_kernel(Shape4 so, int hi, int wi, View in, View out) {
  GRID_STRIDE_GLOBAL(g) {
    int n, y, x, c;
    if (!decode<V>(g, so, n, y, x, c)) return;
    const int y0 = ap_start(y, hi, so.h), y1 = ap_end(y, hi, so.h), x0 = ap_start(x, wi, so.w), x1 = ap_end(x, wi, so.w);
    float a[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        float v[V];
        VecIO<T, V>::ld(vptr<T>(in, n, yy, xx) + c, v);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += v[e];
      }
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] *= inv;
    VecIO<T, V>::st(vptr<T>(out, n, y, x) + c, a);
  }
}

// gather form of the backward pass: input pixel (y, x) collects gout / area from every output cell whose window holds it
// (windows overlap by at most one pixel per side, so the candidates are the cells around floor(y * out / in))
template <typename T, int V>
__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_kernel(Shape4 si, int ho, int wo, View gout, View gin, int accumulate) {
  GRID_STRIDE_GLOBAL(g) {
    int n, y, x, c;
    if (!decode<V>(g, si, n, y, x, c)) return;
    float a[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = 0.f;
    const int oy_c = (int)(((int64_t)y * ho) / si.h), ox_c = (int)(((int64_t)x * wo) / si.w);
    for (int oy = max(0, oy_c - 1); oy <= min(ho - 1, oy_c + 1); ++oy) {
      const int y0 = ap_start(oy, si.h, ho), y1 = ap_end(oy, si.h, ho);
      if (y < y0 || y >= y1) continue;
      for (int ox = max(0, ox_c - 1); ox <= min(wo - 1, ox_c + 1); ++ox) {
        const int x0 = ap_start(ox, si.w, wo), x1 = ap_end(ox, si.w, wo);
        if (x < x0 || x >= x1) continue;
        float v[V];
        VecIO<T, V>::ld(vptr<T>(gout, n, oy, ox) + c, v);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += v[e] * inv;
      }
    }
    T* dst = vptr<T>(gin, n, y, x) + c;
    if (accumulate) {
      float o[V];
      VecIO<T, V>::ld(dst, o);
#pragma unroll
      for (int e = 0; e < V; ++e) a[e] += o[e];
    }
    VecIO<T, V>::st(dst, a);
  }
}

// src = min(floorf(dst * (float)(1/scale_factor)), in-1)
template <typename T>
__global__ __launch_bounds__(256) void batch_reduce_kernel(Shape4 s, View in, float* out, int accumulate, float unscale) {
  // a workgroup owns 64 consecutive (y, x, c) elements; its 4 waves split the batch and are combined in a fixed order
  __shared__ float part[4][64];
  const int64_t per = (int64_t)s.h * s.w * s.c;
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  for (int64_t g0 = (int64_t)blockIdx.x * 64; g0 < per; g0 += (int64_t)gridDim.x * 64) {
    const int64_t g = g0 + lane;
    float a = 0.f;
    if (g < per) {
      const int c = (int)(g % s.c);
      const int x = (int)((g / s.c) % s.w);
      const int y = (int)(g / ((int64_t)s.c * s.w));
      const T* p0 = vptr<T>(in, 0, y, x) + c;
      for (int n = phase; n < s.n; n += 4) a += Elem<T>::ld(p0 + (int64_t)n * in.sn);
    }
    part[phase][lane] = a;
    __syncthreads();
    if (phase == 0 && g < per) {
      const float t = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * unscale;
      out[g] = accumulate ? out[g] + t : t;
    }
    __syncthreads();
  }
}

// whole contiguous tensors (the common case: gradient accumulation, fills): flat 16-byte streaming, no index decoding
template <typename T>
__global__ __launch_bounds__(256) void axpby_flat_kernel(int64_t nvec, const uint4* in, uint4* out, float alpha, int accumulate) {
  constexpr int E = 16 / sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    Pack16<T> a, o;
    float v[E];
    if (in) {
      a.v = in[i];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = alpha * Elem<T>::ld(&a.e[e]);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = alpha;
    }
    if (accumulate) {
      o.v = out[i];
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] += Elem<T>::ld(&o.e[e]);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) Elem<T>::st(&o.e[e], v[e]);
    out[i] = o.v;
  }
}

template <typename T, typename S>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(Shape4 s, const S* src, View out, float sub, float mul) {
  GRID_STRIDE_GLOBAL(g) {
    int n, y, x, c;
    if (!decode<1>(g, s, n, y, x, c)) return;
    const float v = ((float)src[(((int64_t)n * s.c + c) * s.h + y) * s.w + x] - sub) * mul;
    Elem<T>::st(vptr<T>(out, n, y, x) + c, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(Shape4 s, View in, float* dst) {
  const int64_t total = (int64_t)s.n * s.c * s.h * s.w;
  GRID_STRIDE_GLOBAL(g) {
    if (g >= total) return;
    const int x = (int)(g % s.w);
    const int y = (int)((g / s.w) % s.h);
    const int c = (int)((g / ((int64_t)s.w * s.h)) % s.c);
    const int n = (int)(g / ((int64_t)s.w * s.h * s.c));
    dst[g] = Elem<T>::ld(vptr<T>(in, n, y, x) + c);
  }
}

// im2col for the KSxKS stem: out[n,y,x, c*KS*KS + tap] = in[n, y+dy, x+dx, c] (zero outside the image and in the padding
// channels [C*KS*KS, out.cpad)).  Turns the thin-K 7x7 stem (Ci = 1 or 3) into a 1x1 conv over 49*Ci channels that the
// tiled MFMA kernels serve; the OIHW weight [Co][Ci][7][7] is already the [Co][49*Ci] matrix this needs.
template <typename T, int KS>  // KS > 0: compile-time kernel size (the divisions below become multiplies); 0: run-time `ks`
__global__ __launch_bounds__(256) void im2col_kernel(Shape4 s, int ks_rt, int cin, View in, View out, int cphys) {
  const int ks = KS > 0 ? KS : ks_rt;
  const int taps = ks * ks, pad = ks / 2;
  const int groups = cphys / 8;
  const int64_t total = (int64_t)s.n * s.h * s.w * groups;
  const bool vec = ((uintptr_t)out.p % 16 == 0) && ((out.sn * sizeof(T)) % 16 == 0) && ((out.sh * sizeof(T)) % 16 == 0) && ((out.sw * sizeof(T)) % 16 == 0);
  GRID_STRIDE_GLOBAL(g) {
    if (g >= total) return;
    const int cg = (int)(g % groups) * 8;
    int64_t r = g / groups;
    const int x = (int)(r % s.w); r /= s.w;
    const int y = (int)(r % s.h);
    const int n = (int)(r / s.h);
    const T* src0 = vptr<T>(in, n, y, x);
    T vals[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int oc = cg + e;
      const int c = oc / taps, tap = oc - c * taps;
      const int dy = tap / ks - pad, dx = tap - (tap / ks) * ks - pad;
      const int yy = y + dy, xx = x + dx;
      const bool ok = oc < cin * taps && yy >= 0 && yy < s.h && xx >= 0 && xx < s.w;
      // (never a load under a condition: hipcc would branch around it and wait on the spot; an in-range dummy address instead)
      const T v = *(ok ? src0 + ((int64_t)dy * in.sh + (int64_t)dx * in.sw + c) : src0);
      vals[e] = ok ? v : (T)0;
    }
    T* dst = vptr<T>(out, n, y, x) + cg;
    if (vec && sizeof(T) == 2) {
      union { T e[8]; uint4 v; } u;
#pragma unroll
      for (int e = 0; e < 8; ++e) u.e[e] = vals[e];
      *(uint4*)dst = u.v;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = vals[e];
    }
  }
}


// ============================================================================= direct 7x7 stem conv (vae.py:104-110, 126)
// The encoder's first layer: Cin = 1 or 3, 7x7, Cout = widths[0].  K = 49 * Cin is too thin for the tiled MFMA convs, and
// the im2col + 1x1 route writes and re-reads a [N,H,W,56] patch tensor (132 MB at 32 x 192^2) on the forward critical path.
// Here a workgroup stages the (8+6) x (32+6) input halo tile and the whole [49 * Cin][Cout] weight matrix in LDS as f32;
// a thread keeps the 49-value window of its pixel in registers and accumulates all Cout outputs (weights are LDS broadcast
// reads: every lane of a wave asks for the same address).  Output traffic only: 2 B in, 2 * Cout B out per pixel.
#define STEM_TH 8
#define STEM_TW 64
template <typename T, int NCO>  // NCO: Cout / 16 (1, 2 or 4)
__global__ __launch_bounds__(256) void stem7_kernel(int N, int H, int W, int cin, View in, const float* __restrict__ wgt,
                                                    const float* __restrict__ bias, View out, int round_bf16) {
  constexpr int KS = 7, TAPS = 49, PAD = 3, XH = STEM_TH + KS - 1, XW = STEM_TW + KS - 1, CO = NCO * 16;
  constexpr int NPX = NCO == 4 ? 1 : 2;  // pixels per thread (columns tx and tx + 32): every weight read from LDS feeds NPX FMAs
  extern __shared__ __attribute__((aligned(16))) float stem_smem[];
  float* xs = stem_smem;                   // [cin][XH][XW]
  float* ws = stem_smem + cin * XH * XW;   // [cin * 49][CO]   (offset is a multiple of 4 floats: XH * XW = 14 * 70)
  const int tid = threadIdx.x;
  const int tiles_x = (W + STEM_TW - 1) / STEM_TW, tiles_y = (H + STEM_TH - 1) / STEM_TH;
  int b = blockIdx.x;
  const int txi = b % tiles_x; b /= tiles_x;
  const int tyi = b % tiles_y;
  const int n = b / tiles_y;
  const int y0 = tyi * STEM_TH, x0 = txi * STEM_TW;
  for (int i = tid; i < cin * TAPS * CO; i += 256) {  // OIHW [co][c][7][7] -> [c * 49 + tap][co]
    const int co = i % CO, k = i / CO;
    float v = wgt[(size_t)co * cin * TAPS + k];
    if (round_bf16) v = h2f(f2h(v));  // the bf16 engine multiplies bf16-rounded weights everywhere else too
    ws[i] = v;
  }
  for (int i = tid; i < cin * XH * XW; i += 256) {
    const int xx = i % XW, r = i / XW, yy = r % XH, c = r / XH;
    const int gy = y0 + yy - PAD, gx = x0 + xx - PAD;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const T* src = vptr<T>(in, n, ok ? gy : 0, ok ? gx : 0) + c;
    const float v = Elem<T>::ld(src);
    xs[i] = ok ? v : 0.f;
  }
  __syncthreads();
  const int ty = tid >> 5, tx = tid & 31;
#pragma unroll 1
  for (int half = 0; half < 2 / NPX; ++half) {  // (NCO == 4: one pixel at a time, twice)
    float acc[NPX][CO];
#pragma unroll
    for (int q = 0; q < NPX; ++q)
#pragma unroll
      for (int j = 0; j < CO; ++j) acc[q][j] = bias ? bias[j] : 0.f;
    for (int c = 0; c < cin; ++c) {
      float win[NPX][TAPS];
      const float* xp = xs + (c * XH + ty) * XW + tx + half * 32;
#pragma unroll
      for (int q = 0; q < NPX; ++q)
#pragma unroll
        for (int dy = 0; dy < KS; ++dy)
#pragma unroll
          for (int dx = 0; dx < KS; ++dx) win[q][dy * KS + dx] = xp[dy * XW + dx + q * 32];
      const float* wp = ws + (size_t)c * TAPS * CO;
#pragma unroll  // (fully: a run-time index would put the window in scratch)
      for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int j = 0; j < CO; j += 4) {
          const float4 w4 = *(const float4*)(wp + t * CO + j);
#pragma unroll
          for (int q = 0; q < NPX; ++q) {
            const float xv = win[q][t];
            acc[q][j] = fmaf(xv, w4.x, acc[q][j]); acc[q][j + 1] = fmaf(xv, w4.y, acc[q][j + 1]);
            acc[q][j + 2] = fmaf(xv, w4.z, acc[q][j + 2]); acc[q][j + 3] = fmaf(xv, w4.w, acc[q][j + 3]);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
      const int gy = y0 + ty, gx = x0 + tx + (half + q) * 32;
      if (gy < H && gx < W) {
        T* dst = vptr<T>(out, n, gy, gx);
#pragma unroll
        for (int j = 0; j < CO; j += 4) {
          float v[4] = {acc[q][j], acc[q][j + 1], acc[q][j + 2], acc[q][j + 3]};
          VecIO<T, 4>::st(dst + j, v);
        }
      }
    }
  }
}

static inline int grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;  // grid-stride beyond ~16 blocks per CU
  return (int)b;
}

static inline bool vec4_ok(int esz, int c, std::initializer_list<const cgen_view*> vs) {
  if (c % 4) return false;
  const int q = 4 * esz;
  for (const cgen_view* v : vs) {
    if (!v || !v->p) continue;
    if (((uintptr_t)v->p % q) || ((v->sn * esz) % q) || ((v->sh * esz) % q) || ((v->sw * esz) % q)) return false;
  }
  return true;
}

#define DISPATCH_TV(dtype, vec, KERNEL, items, stream, ...)                                                   \
  do {                                                                                                        \
    const int grid__ = grid_for(items);                                                                       \
    if ((dtype) == CGEN_F32) {                                                                                \
      if (vec) hipLaunchKernelGGL((KERNEL<float, 4>), dim3(grid__), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<float, 1>), dim3(grid__), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);     \
    } else {                                                                                                  \
      if (vec) hipLaunchKernelGGL((KERNEL<h16_t, 4>), dim3(grid__), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<h16_t, 1>), dim3(grid__), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);     \
    }                                                                                                         \
  } while (0)


// ----------------------------------------------------------------------------- strided im2col / col2im, unary ops
// (config 1, simple_vae.py: 5x5/s2/p1 and 3x3/s2/p1 convolutions run as im2col + a 1x1 conv; LeakyReLU; clamp(min))
template <typename T>
__global__ __launch_bounds__(256) void im2col_strided_kernel(Shape4 so, int hi, int wi, int ks, int stride, int pad, int cin, View in, View out,
                                                             int cphys) {
  const int taps = ks * ks;
  const int64_t total = (int64_t)so.n * so.h * so.w * cphys;
  GRID_STRIDE_GLOBAL(g) {
    if (g >= total) return;
    const int oc = (int)(g % cphys);
    int64_t r = g / cphys;
    const int ox = (int)(r % so.w); r /= so.w;
    const int oy = (int)(r % so.h);
    const int n = (int)(r / so.h);
    T v = (T)0;
    if (oc < cin * taps) {
      const int c = oc / taps, tap = oc - c * taps;
      const int yy = oy * stride - pad + tap / ks, xx = ox * stride - pad + tap % ks;
      if (yy >= 0 && yy < hi && xx >= 0 && xx < wi) v = *(vptr<T>(in, n, yy, xx) + c);
    }
    *(vptr<T>(out, n, oy, ox) + oc) = v;
  }
}

// gin[n,y,x,c] (+)= sum over (oy, ox, tap) that read this input pixel of gcol[n,oy,ox, c*taps + tap]   (gather form)
template <typename T>
__global__ __launch_bounds__(256) void col2im_strided_kernel(Shape4 si, int ho, int wo, int ks, int stride, int pad, View gcol, View gin,
                                                             int accumulate) {
  const int taps = ks * ks;
  const int64_t total = (int64_t)si.n * si.h * si.w * si.c;
  GRID_STRIDE_GLOBAL(g) {
    if (g >= total) return;
    const int c = (int)(g % si.c);
    int64_t r = g / si.c;
    const int x = (int)(r % si.w); r /= si.w;
    const int y = (int)(r % si.h);
    const int n = (int)(r / si.h);
    float a = 0.f;
    for (int tap = 0; tap < taps; ++tap) {
      const int ny = y + pad - tap / ks, nx = x + pad - tap % ks;
      if (ny < 0 || nx < 0 || ny % stride || nx % stride) continue;
      const int oy = ny / stride, ox = nx / stride;
      if (oy < ho && ox < wo) a += Elem<T>::ld(vptr<T>(gcol, n, oy, ox) + c * taps + tap);
    }
    T* o = vptr<T>(gin, n, y, x) + c;
    Elem<T>::st(o, accumulate ? Elem<T>::ld(o) + a : a);
  }
}

__device__ __forceinline__ float unary_f(int op, float p, float x) {
  if (op == CGEN_UNARY_LEAKY_RELU) return x > 0.f ? x : p * x;
  if (op == CGEN_UNARY_CLAMP_MIN) return x < p ? p : x;  // NaN stays NaN, as torch.clamp
  if (op == CGEN_UNARY_ADD) return x + p;
  return act_fwd(op, x);
}
__device__ __forceinline__ float unary_df(int op, float p, float x) {
  if (op == CGEN_UNARY_LEAKY_RELU) return x > 0.f ? 1.f : p;
  if (op == CGEN_UNARY_CLAMP_MIN) return x < p ? 0.f : 1.f;  // torch: gradient passes where x >= min
  if (op == CGEN_UNARY_ADD) return 1.f;
  return act_bwd(op, x);
}
template <typename T>
__global__ __launch_bounds__(256) void unary_fwd_kernel(Shape4 s, int op, float p, View in, View out) {
  const int64_t total = (int64_t)s.n * s.h * s.w * s.c;
  GRID_STRIDE_GLOBAL(g) {
    if (g >= total) return;
    const int c = (int)(g % s.c);
    int64_t r = g / s.c;
    const int x = (int)(r % s.w); r /= s.w;
    const int y = (int)(r % s.h);
    const int n = (int)(r / s.h);
    Elem<T>::st(vptr<T>(out, n, y, x) + c, unary_f(op, p, Elem<T>::ld(vptr<T>(in, n, y, x) + c)));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void unary_bwd_kernel(Shape4 s, int op, float p, View xin, View gout, View gin, int accumulate) {
  const int64_t total = (int64_t)s.n * s.h * s.w * s.c;
  GRID_STRIDE_GLOBAL(g) {
    if (g >= total) return;
    const int c = (int)(g % s.c);
    int64_t r = g / s.c;
    const int x = (int)(r % s.w); r /= s.w;
    const int y = (int)(r % s.h);
    const int n = (int)(r / s.h);
    const float v = Elem<T>::ld(vptr<T>(gout, n, y, x) + c) * unary_df(op, p, Elem<T>::ld(vptr<T>(xin, n, y, x) + c));
    T* o = vptr<T>(gin, n, y, x) + c;
    Elem<T>::st(o, accumulate ? Elem<T>::ld(o) + v : v);
  }
}

}  // namespace cgen

using namespace cgen;

#define CHECK_DTYPE(name) CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, name ": bad dtype %d", dtype)
static inline int esz_of(int dtype) { return dtype == CGEN_F32 ? 4 : 2; }

extern "C" int cgen_avgpool_fwd(int32_t dtype, int32_t n, int32_t ho, int32_t wo, int32_t d, cgen_view in, cgen_view out,
                                cgen_stream_t stream) {
  CHECK_DTYPE("cgen_avgpool_fwd");
  CGEN_REQUIRE(in.p && out.p && d >= 1 && in.c == out.c && n > 0 && ho > 0 && wo > 0, "cgen_avgpool_fwd: bad args");
  Shape4 so{n, ho, wo, out.c};
  const bool v = vec4_ok(esz_of(dtype), out.c, {&in, &out});
  const int64_t items = (int64_t)n * ho * wo * (v ? out.c / 4 : out.c);
  DISPATCH_TV(dtype, v, avgpool_fwd_kernel, items, stream, so, d, mk(in), mk(out));
  return check_launch("cgen_avgpool_fwd");
}

extern "C" int cgen_avgpool_bwd(int32_t dtype, int32_t n, int32_t ho, int32_t wo, int32_t d, cgen_view gout, cgen_view gin,
                                int32_t accumulate, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_avgpool_bwd");
  CGEN_REQUIRE(gout.p && gin.p && d >= 1 && gin.c == gout.c, "cgen_avgpool_bwd: bad args");
  Shape4 si{n, ho * d, wo * d, gin.c};
  const bool v = vec4_ok(esz_of(dtype), gin.c, {&gout, &gin});
  const int64_t items = (int64_t)n * si.h * si.w * (v ? gin.c / 4 : gin.c);
  DISPATCH_TV(dtype, v, avgpool_bwd_kernel, items, stream, si, d, mk(gout), mk(gin), accumulate);
  return check_launch("cgen_avgpool_bwd");
}

extern "C" int cgen_adaptive_avgpool_fwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view in, cgen_view out,
                                         cgen_stream_t stream) {
  CHECK_DTYPE("cgen_adaptive_avgpool_fwd");
  CGEN_REQUIRE(in.p && out.p && in.c == out.c && n > 0 && ho > 0 && wo > 0 && hi >= ho && wi >= wo, "cgen_adaptive_avgpool_fwd: bad args");
  Shape4 so{n, ho, wo, out.c};
  const bool v = vec4_ok(esz_of(dtype), out.c, {&in, &out});
  const int64_t items = (int64_t)n * ho * wo * (v ? out.c / 4 : out.c);
  DISPATCH_TV(dtype, v, adaptive_avgpool_fwd_kernel, items, stream, so, hi, wi, mk(in), mk(out));
  return check_launch("cgen_adaptive_avgpool_fwd");
}

extern "C" int cgen_adaptive_avgpool_bwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view gout, cgen_view gin,
                                         int32_t accumulate, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_adaptive_avgpool_bwd");
  CGEN_REQUIRE(gout.p && gin.p && gin.c == gout.c && n > 0 && ho > 0 && wo > 0 && hi >= ho && wi >= wo, "cgen_adaptive_avgpool_bwd: bad args");
  Shape4 si{n, hi, wi, gin.c};
  const bool v = vec4_ok(esz_of(dtype), gin.c, {&gout, &gin});
  const int64_t items = (int64_t)n * hi * wi * (v ? gin.c / 4 : gin.c);
  DISPATCH_TV(dtype, v, adaptive_avgpool_bwd_kernel, items, stream, si, ho, wo, mk(gout), mk(gin), accumulate);
  return check_launch("cgen_adaptive_avgpool_bwd");
}

static inline float inv_scale(int out, int in) { return (float)(1.0 / ((double)out / (double)in)); }

extern "C" int cgen_upsample_fwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view in,
                                 const float* bias, cgen_view out, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_upsample_fwd");
  CGEN_REQUIRE(in.p && out.p && in.c == out.c && hi > 0 && wi > 0 && ho >= hi && wo >= wi, "cgen_upsample_fwd: bad args");
  Shape4 so{n, ho, wo, out.c};
  const bool v = vec4_ok(esz_of(dtype), out.c, {&in, &out});
  const int64_t items = (int64_t)n * ho * wo * (v ? out.c / 4 : out.c);
  DISPATCH_TV(dtype, v, upsample_fwd_kernel, items, stream, so, hi, wi, inv_scale(ho, hi), inv_scale(wo, wi), mk(in), bias, mk(out));
  return check_launch("cgen_upsample_fwd");
}

extern "C" int cgen_upsample_bwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view gout,
                                 cgen_view gin, int32_t accumulate, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_upsample_bwd");
  CGEN_REQUIRE(gout.p && gin.p && gin.c == gout.c, "cgen_upsample_bwd: bad args");
  Shape4 si{n, hi, wi, gin.c};
  const bool v = vec4_ok(esz_of(dtype), gin.c, {&gout, &gin});
  const int64_t items = (int64_t)n * hi * wi * (v ? gin.c / 4 : gin.c);
  DISPATCH_TV(dtype, v, upsample_bwd_kernel, items, stream, si, ho, wo, inv_scale(ho, hi), inv_scale(wo, wi), mk(gout), mk(gin), accumulate);
  return check_launch("cgen_upsample_bwd");
}

extern "C" int cgen_batch_reduce(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view in, float* out, int32_t accumulate,
                                 float unscale, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_batch_reduce");
  CGEN_REQUIRE(in.p && out, "cgen_batch_reduce: bad args");
  Shape4 s{n, h, w, in.c};
  const int64_t items = (int64_t)h * w * in.c;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(batch_reduce_kernel<float>, dim3(grid_for(items * 4)), dim3(256), 0, (hipStream_t)stream, s, mk(in), out, accumulate, unscale);
  else hipLaunchKernelGGL(batch_reduce_kernel<h16_t>, dim3(grid_for(items * 4)), dim3(256), 0, (hipStream_t)stream, s, mk(in), out, accumulate, unscale);
  return check_launch("cgen_batch_reduce");
}

extern "C" int cgen_batch_broadcast(int32_t dtype, int32_t n, int32_t h, int32_t w, const float* src, cgen_view out,
                                    cgen_stream_t stream) {
  CHECK_DTYPE("cgen_batch_broadcast");
  CGEN_REQUIRE(src && out.p, "cgen_batch_broadcast: bad args");
  Shape4 s{n, h, w, out.c};
  const bool v = vec4_ok(esz_of(dtype), out.c, {&out});
  const int64_t items = (int64_t)n * h * w * (v ? out.c / 4 : out.c);
  DISPATCH_TV(dtype, v, batch_broadcast_kernel, items, stream, s, src, mk(out));
  return check_launch("cgen_batch_broadcast");
}

extern "C" int cgen_axpby(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view in, cgen_view out, float alpha, float beta,
                          int32_t c_from, int32_t accumulate, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_axpby");
  CGEN_REQUIRE(out.p && (!in.p || in.c == out.c), "cgen_axpby: bad args");
  Shape4 s{n, h, w, out.c};
  {  // flat fast path: both tensors whole and contiguous, no per-channel scaling
    const int esz = esz_of(dtype), c = out.c;
    auto flat = [&](const cgen_view& v) {
      return v.sw == c && v.sh == (int64_t)w * c && v.sn == (int64_t)h * w * c && ((uintptr_t)v.p % 16) == 0;
    };
    if (c % (16 / esz) == 0 && c_from >= c && flat(out) && (!in.p || (flat(in) && in.c == c))) {
      const int64_t nvec = (int64_t)n * h * w * c * esz / 16;
      const dim3 g(grid_for(nvec)), b(256);
      if (dtype == CGEN_F32) hipLaunchKernelGGL(axpby_flat_kernel<float>, g, b, 0, (hipStream_t)stream, nvec, (const uint4*)in.p, (uint4*)out.p, alpha, accumulate);
      else hipLaunchKernelGGL(axpby_flat_kernel<h16_t>, g, b, 0, (hipStream_t)stream, nvec, (const uint4*)in.p, (uint4*)out.p, alpha, accumulate);
      return check_launch("cgen_axpby");
    }
  }
  const bool v = vec4_ok(esz_of(dtype), out.c, {&in, &out});
  const int64_t items = (int64_t)n * h * w * (v ? out.c / 4 : out.c);
  DISPATCH_TV(dtype, v, axpby_kernel, items, stream, s, mk(in), mk(out), alpha, beta, c_from, accumulate);
  return check_launch("cgen_axpby");
}

extern "C" int cgen_nchw_to_nhwc(int32_t src_is_u8, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, const void* src,
                                 cgen_view out, float sub, float mul, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_nchw_to_nhwc");
  CGEN_REQUIRE(src && out.p && out.c == c, "cgen_nchw_to_nhwc: bad args");
  Shape4 s{n, h, w, c};
  const int64_t items = (int64_t)n * h * w * c;
  const dim3 g(grid_for(items)), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CGEN_F32) {
    if (src_is_u8) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, uint8_t>), g, b, 0, st, s, (const uint8_t*)src, mk(out), sub, mul);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), g, b, 0, st, s, (const float*)src, mk(out), sub, mul);
  } else {
    if (src_is_u8) hipLaunchKernelGGL((nchw_to_nhwc_kernel<h16_t, uint8_t>), g, b, 0, st, s, (const uint8_t*)src, mk(out), sub, mul);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<h16_t, float>), g, b, 0, st, s, (const float*)src, mk(out), sub, mul);
  }
  return check_launch("cgen_nchw_to_nhwc");
}

extern "C" int cgen_im2col(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t ks, cgen_view in, cgen_view out,
                           cgen_stream_t stream) {
  CHECK_DTYPE("cgen_im2col");
  CGEN_REQUIRE(in.p && out.p && (ks == 3 || ks == 5 || ks == 7) && out.c == in.c * ks * ks, "cgen_im2col: bad args");
  const int cphys = (out.c + 7) / 8 * 8;
  CGEN_REQUIRE(out.cpad >= cphys, "cgen_im2col: out.cpad must cover the 8-channel padding (%d < %d)", out.cpad, cphys);
  Shape4 s{n, h, w, out.c};
  const int64_t items = (int64_t)n * h * w * (cphys / 8);
  if (dtype == CGEN_F32) {
    if (ks == 7) hipLaunchKernelGGL((im2col_kernel<float, 7>), dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, ks, in.c, mk(in), mk(out), cphys);
    else hipLaunchKernelGGL((im2col_kernel<float, 0>), dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, ks, in.c, mk(in), mk(out), cphys);
  } else {
    if (ks == 7) hipLaunchKernelGGL((im2col_kernel<h16_t, 7>), dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, ks, in.c, mk(in), mk(out), cphys);
    else hipLaunchKernelGGL((im2col_kernel<h16_t, 0>), dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, ks, in.c, mk(in), mk(out), cphys);
  }
  return check_launch("cgen_im2col");
}


extern "C" int cgen_stem_conv_supported(int32_t dtype, int32_t cin, int32_t ks, int32_t co) {
  // the kernel stages the halo tile and the whole [49 Cin][Cout] weight matrix in LDS as f32 and launches without the
  // >64 KB opt-in: a shape whose footprint exceeds 64 KB (Cin = 4 with Cout = 64: 65 856 B) is NOT served -- the caller
  // (vae._stem_site) then takes the im2col + 1x1 route instead of failing at launch
  const size_t lds = (size_t)(cin * (STEM_TH + 6) * (STEM_TW + 6) + cin * 49 * co) * sizeof(float);
  return (dtype == CGEN_F32 || dtype == CGEN_F16) && ks == 7 && cin >= 1 && cin <= 4 && (co == 16 || co == 32 || co == 64) &&
         lds <= 64 * 1024;
}

extern "C" int cgen_stem_conv_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t ks, int32_t co, cgen_view in,
                                  const float* weight_oihw, const float* bias, cgen_view out, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_stem_conv_fwd");
  CGEN_REQUIRE(cgen_stem_conv_supported(dtype, cin, ks, co), "cgen_stem_conv_fwd: shape not served (7x7, Cin <= 4, Cout 16 / 32 / 64)");
  CGEN_REQUIRE(in.p && out.p && weight_oihw && in.c == cin && out.c == co && n > 0 && h > 0 && w > 0, "cgen_stem_conv_fwd: bad args");
  const int esz = esz_of(dtype);
  CGEN_REQUIRE(vec4_ok(esz, co, {&out}), "cgen_stem_conv_fwd: the output view must allow 4-channel vector stores");
  const int tiles = n * ((h + STEM_TH - 1) / STEM_TH) * ((w + STEM_TW - 1) / STEM_TW);
  const size_t lds = (size_t)(cin * (STEM_TH + 6) * (STEM_TW + 6) + cin * 49 * co) * sizeof(float);
  const int rb = dtype == CGEN_F16 ? 1 : 0;
#define STEM_LAUNCH(T_, NCO_) hipLaunchKernelGGL((stem7_kernel<T_, NCO_>), dim3(tiles), dim3(256), lds, (hipStream_t)stream, n, h, w, cin, mk(in), weight_oihw, bias, mk(out), rb)
  if (dtype == CGEN_F32) {
    if (co == 16) STEM_LAUNCH(float, 1); else if (co == 32) STEM_LAUNCH(float, 2); else STEM_LAUNCH(float, 4);
  } else {
    if (co == 16) STEM_LAUNCH(h16_t, 1); else if (co == 32) STEM_LAUNCH(h16_t, 2); else STEM_LAUNCH(h16_t, 4);
  }
#undef STEM_LAUNCH
  return check_launch("cgen_stem_conv_fwd");
}

extern "C" int cgen_nhwc_to_nchw(int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, cgen_view in, float* dst,
                                 cgen_stream_t stream) {
  CHECK_DTYPE("cgen_nhwc_to_nchw");
  CGEN_REQUIRE(in.p && dst && in.c == c, "cgen_nhwc_to_nchw: bad args");
  Shape4 s{n, h, w, c};
  const int64_t items = (int64_t)n * h * w * c;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, mk(in), dst);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<h16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, mk(in), dst);
  return check_launch("cgen_nhwc_to_nchw");
}

extern "C" int cgen_im2col_strided(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t ks, int32_t stride, int32_t pad, int32_t ho,
                                   int32_t wo, cgen_view in, cgen_view out, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_im2col_strided");
  CGEN_REQUIRE(in.p && out.p && ks >= 1 && stride >= 1 && pad >= 0 && out.c == in.c * ks * ks, "cgen_im2col_strided: bad args");
  CGEN_REQUIRE(ho == (h + 2 * pad - ks) / stride + 1 && wo == (w + 2 * pad - ks) / stride + 1, "cgen_im2col_strided: bad output size");
  const int cphys = out.cpad > out.c ? out.cpad : out.c;
  Shape4 so{n, ho, wo, out.c};
  const int64_t items = (int64_t)n * ho * wo * cphys;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(im2col_strided_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, so, h, w, ks, stride, pad, in.c, mk(in), mk(out), cphys);
  else hipLaunchKernelGGL(im2col_strided_kernel<h16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, so, h, w, ks, stride, pad, in.c, mk(in), mk(out), cphys);
  return check_launch("cgen_im2col_strided");
}

extern "C" int cgen_col2im_strided(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t ks, int32_t stride, int32_t pad, int32_t ho,
                                   int32_t wo, cgen_view gcol, cgen_view gin, int32_t accumulate, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_col2im_strided");
  CGEN_REQUIRE(gcol.p && gin.p && ks >= 1 && stride >= 1 && pad >= 0 && gcol.c == gin.c * ks * ks, "cgen_col2im_strided: bad args");
  Shape4 si{n, h, w, gin.c};
  const int64_t items = (int64_t)n * h * w * gin.c;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(col2im_strided_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, si, ho, wo, ks, stride, pad, mk(gcol), mk(gin), accumulate);
  else hipLaunchKernelGGL(col2im_strided_kernel<h16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, si, ho, wo, ks, stride, pad, mk(gcol), mk(gin), accumulate);
  return check_launch("cgen_col2im_strided");
}

extern "C" int cgen_unary_fwd(int32_t dtype, int32_t op, float param, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view in, cgen_view out,
                              cgen_stream_t stream) {
  CHECK_DTYPE("cgen_unary_fwd");
  CGEN_REQUIRE(in.p && out.p && op >= 0 && op <= CGEN_UNARY_ADD, "cgen_unary_fwd: bad args");
  Shape4 s{n, h, w, c};
  const int64_t items = (int64_t)n * h * w * c;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(unary_fwd_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, op, param, mk(in), mk(out));
  else hipLaunchKernelGGL(unary_fwd_kernel<h16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, op, param, mk(in), mk(out));
  return check_launch("cgen_unary_fwd");
}

extern "C" int cgen_unary_bwd(int32_t dtype, int32_t op, float param, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view x, cgen_view gout,
                              cgen_view gin, int32_t accumulate, cgen_stream_t stream) {
  CHECK_DTYPE("cgen_unary_bwd");
  CGEN_REQUIRE(x.p && gout.p && gin.p && op >= 0 && op <= CGEN_UNARY_ADD, "cgen_unary_bwd: bad args");
  Shape4 s{n, h, w, c};
  const int64_t items = (int64_t)n * h * w * c;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(unary_bwd_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, op, param, mk(x), mk(gout), mk(gin), accumulate);
  else hipLaunchKernelGGL(unary_bwd_kernel<h16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, s, op, param, mk(x), mk(gout), mk(gin), accumulate);
  return check_launch("cgen_unary_bwd");
}
