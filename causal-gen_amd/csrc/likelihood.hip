// Likelihood kernels: discretised Gaussian (vae.py:352-422), discretised mixture of logistics (dmol.py), ELBO
// assembly (vae.py:450-457) and the counterfactual pixel step (dscm.py:55-63).  All f32 math; HBM-bound.
#include "common.h"

namespace cgen {

#define LIKE_PIX 1024  // pixels per block (4 per thread)

// ----------------------------------------------------------------------------- discretised Gaussian
#define DG_MIN_LS (-9.0f)
#define DG_K0 0.79788456080286535588f  // sqrt(2/pi)
#define DG_K1 0.044715f

__device__ __forceinline__ float dg_cdf(float u) { return 0.5f * (1.0f + tanhf(DG_K0 * (u + DG_K1 * u * u * u))); }
__device__ __forceinline__ float dg_cdf_grad(float u) {
  const float t = tanhf(DG_K0 * (u + DG_K1 * u * u * u));
  return 0.5f * (1.f - t * t) * DG_K0 * (1.f + 3.f * DG_K1 * u * u);
}

struct DgP {
  int n, h, w, c;
  int ar, pad0;  // ar: three channels in vae.py's autoregressive form (params = [loc | logscale | coeffs], 9 channels); 0: independent channels
  View params, x, g;
  float* part;
  const float* coef;
  int coef_stride;
  float hb, logc;
};

// per-pixel parameter decode shared by fwd/bwd: loc[c], ls[c] (clamped), tanh coeffs
template <typename T>
__device__ __forceinline__ void dg_load(const DgP& p, int b, int y, int x, float (&loc)[3], float (&ls_raw)[3], float (&k)[3],
                                        float (&xv)[3]) {
  const T* pp = vptr<T>(p.params, b, y, x);
  const T* xp = vptr<T>(p.x, b, y, x);
  for (int c = 0; c < p.c; ++c) {
    loc[c] = Elem<T>::ld(pp + c);
    ls_raw[c] = Elem<T>::ld(pp + p.c + c);
    xv[c] = Elem<T>::ld(xp + c);
  }
  if (p.ar) {
    for (int j = 0; j < 3; ++j) k[j] = tanhf(Elem<T>::ld(pp + 6 + j));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dgauss_nll_fwd_kernel(DgP p) {
  __shared__ float sm[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int npix = p.h * p.w;
  float acc = 0.f;
  for (int i = 0; i < LIKE_PIX / 256; ++i) {
    const int px = chunk * LIKE_PIX + i * 256 + threadIdx.x;
    if (px < npix) {
      const int y = px / p.w, x = px % p.w;
      float loc[3], lsr[3], k[3], xv[3];
      dg_load<T>(p, b, y, x, loc, lsr, k, xv);
      if (p.ar) {  // training-time autoregressive means use the true x (vae.py:370-377)
        loc[2] = loc[2] + k[1] * xv[0] + k[2] * xv[1];
        loc[1] = loc[1] + k[0] * xv[0];
      }
      for (int c = 0; c < p.c; ++c) {
        const float ls = fmaxf(lsr[c], DG_MIN_LS);
        const float inv = expf(-ls), d = xv[c] - loc[c];
        const float cp = dg_cdf(inv * (d + 1.f / 255.f)), cm = dg_cdf(inv * (d - 1.f / 255.f));
        float lp;
        if (xv[c] < -0.999f) lp = logf(fmaxf(cp, 1e-12f));
        else if (xv[c] > 0.999f) lp = logf(fmaxf(1.f - cm, 1e-12f));
        else lp = logf(fmaxf(cp - cm, 1e-12f));
        acc -= lp;
      }
    }
  }
  const float tot = block_sum_256(acc, sm);
  if (threadIdx.x == 0) p.part[(int64_t)b * gridDim.x + chunk] = tot;
}

template <typename T>
__global__ __launch_bounds__(256) void dgauss_nll_bwd_kernel(DgP p) {
  const int npix = p.h * p.w;
  const int64_t total = (int64_t)p.n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    const int y = px / p.w, x = px % p.w;
    float loc[3], lsr[3], k[3], xv[3];
    dg_load<T>(p, b, y, x, loc, lsr, k, xv);
    if (p.ar) {
      loc[2] = loc[2] + k[1] * xv[0] + k[2] * xv[1];
      loc[1] = loc[1] + k[0] * xv[0];
    }
    const float coef = p.coef[(int64_t)b * p.coef_stride];
    float gloc[3], gls[3];
    for (int c = 0; c < p.c; ++c) {
      const float ls = fmaxf(lsr[c], DG_MIN_LS);
      const float inv = expf(-ls), d = xv[c] - loc[c];
      const float up = inv * (d + 1.f / 255.f), um = inv * (d - 1.f / 255.f);
      const float cp = dg_cdf(up), cm = dg_cdf(um);
      float dup = 0.f, dum = 0.f;  // d lp / d up, d lp / d um
      if (xv[c] < -0.999f) { if (cp >= 1e-12f) dup = dg_cdf_grad(up) / cp; }
      else if (xv[c] > 0.999f) { if (1.f - cm >= 1e-12f) dum = -dg_cdf_grad(um) / (1.f - cm); }
      else { const float de = cp - cm; if (de >= 1e-12f) { dup = dg_cdf_grad(up) / de; dum = -dg_cdf_grad(um) / de; } }
      // up = inv*(d+1/255): d up/d loc = -inv ; d up/d ls = -up
      const float dlp_dloc = -(dup + dum) * inv;
      const float dlp_dls = (lsr[c] >= DG_MIN_LS) ? -(dup * up + dum * um) : 0.f;
      gloc[c] = -coef * dlp_dloc;
      gls[c] = -coef * dlp_dls;
    }
    T* go = vptr<T>(p.g, b, y, x);
    for (int c = 0; c < p.c; ++c) {
      Elem<T>::st(go + c, gloc[c]);
      Elem<T>::st(go + p.c + c, gls[c]);
    }
    if (p.ar) {  // coeff_raw -> tanh -> k0 (g<-r), k1 (b<-r), k2 (b<-g)
      Elem<T>::st(go + 6, gloc[1] * xv[0] * (1.f - k[0] * k[0]));
      Elem<T>::st(go + 7, gloc[2] * xv[0] * (1.f - k[1] * k[1]));
      Elem<T>::st(go + 8, gloc[2] * xv[1] * (1.f - k[2] * k[2]));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dgauss_sample_kernel(int n, int h, int w, int c, int ar, View params, float logt, const uint64_t* rng, uint32_t stream_id, float* xo, float* so) {
  const int npix = h * w;
  const int64_t total = (int64_t)n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    const T* pp = vptr<T>(params, b, px / w, px % w);
    float loc[3];
    for (int ch = 0; ch < c; ++ch) loc[ch] = Elem<T>::ld(pp + ch);
    if (ar) {  // inference: autoregressive on the clamped predicted channels (vae.py:360-369)
      const float k0 = tanhf(Elem<T>::ld(pp + 6)), k1 = tanhf(Elem<T>::ld(pp + 7)), k2 = tanhf(Elem<T>::ld(pp + 8));
      const float r = fminf(fmaxf(loc[0], -1.f), 1.f);
      const float g = fminf(fmaxf(loc[1] + k0 * r, -1.f), 1.f);
      const float bl = fminf(fmaxf(loc[2] + k1 * r + k2 * g, -1.f), 1.f);
      loc[0] = r; loc[1] = g; loc[2] = bl;
    }
    for (int ch = 0; ch < c; ++ch) {
      const int64_t o = ((int64_t)b * c + ch) * npix + px;
      const float sc = expf(fmaxf(Elem<T>::ld(pp + c + ch), DG_MIN_LS) + logt);
      // return_loc=False (vae.py:418-420): x = loc + scale * N(0,1), then the same clamp
      const float noise = rng ? sc * Philox::normal1(rng[0], rng[1], stream_id, (uint64_t)o) : 0.f;
      xo[o] = fminf(fmaxf(loc[ch] + noise, -1.f), 1.f);
      so[o] = sc;
    }
  }
}

// DGaussNet.forward (vae.py:352-386): (loc, logscale) of the pixel distribution from the heads' outputs [loc(c) | logscale(c) | coeffs(3)].
// logscale = max(ls, -9) + log t; RGB: tanh coefficients, autoregressive means on the TRUE pixels when x is given (vae.py:370-377), on the
// clamped predicted channels otherwise (vae.py:360-369).  NCHW f32 out (what the reference returns).
template <typename T>
__global__ __launch_bounds__(256) void dgauss_params_kernel(int n, int h, int w, int c, View params, View x, float logt, float* loc_o, float* ls_o) {
  const int npix = h * w;
  const int64_t total = (int64_t)n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    const T* pp = vptr<T>(params, b, px / w, px % w);
    float loc[3];
    for (int ch = 0; ch < c; ++ch) loc[ch] = Elem<T>::ld(pp + ch);
    if (c == 3) {
      const float k0 = tanhf(Elem<T>::ld(pp + 6)), k1 = tanhf(Elem<T>::ld(pp + 7)), k2 = tanhf(Elem<T>::ld(pp + 8));
      if (x.p) {
        const T* xp = vptr<T>(x, b, px / w, px % w);
        const float xr = Elem<T>::ld(xp), xg = Elem<T>::ld(xp + 1);
        loc[1] = loc[1] + k0 * xr;
        loc[2] = loc[2] + k1 * xr + k2 * xg;
      } else {
        const float r = fminf(fmaxf(loc[0], -1.f), 1.f);
        const float g = fminf(fmaxf(loc[1] + k0 * r, -1.f), 1.f);
        loc[0] = r; loc[1] = g; loc[2] = fminf(fmaxf(loc[2] + k1 * r + k2 * g, -1.f), 1.f);
      }
    }
    for (int ch = 0; ch < c; ++ch) {
      const int64_t o = ((int64_t)b * c + ch) * npix + px;
      loc_o[o] = loc[ch];
      ls_o[o] = fmaxf(Elem<T>::ld(pp + c + ch), DG_MIN_LS) + logt;
    }
  }
}

// ----------------------------------------------------------------------------- differentiable counterfactual pixel step
// dscm.py:52-56 with both likelihood heads' (loc, scale) decoded in place (DGaussNet.sample(h) with return_loc=True,
// vae.py:352-385,413-422):  rec/cf (loc, scale) = decode(params);  u = (x - rec_loc) / max(rec_scale, 1e-12);
// cf_x = clamp(cf_loc + cf_scale * u, -1, 1).  This is the BACKWARD of that composition: given d(loss)/d(cf_x) (NCHW f32,
// times `gscale` = 1 / cf_particles) it writes d/d(params_rec) and d/d(params_cf) in the NHWC parameter layout
// [loc(c) | logscale(c) | coeffs(3 if c == 3)], through the clamps (torch.clamp passes the gradient on [min, max], ends
// included), exp(max(logscale, -9)) and, for RGB, the autoregressive tanh-coefficient chain on the clamped channels.
template <typename T>
__device__ __forceinline__ void dgauss_decode(const T* pp, int c, int ar, float (&pre)[3], float (&loc)[3], float (&k)[3], float (&sc)[3], float (&ls)[3]) {
  for (int ch = 0; ch < c; ++ch) { pre[ch] = Elem<T>::ld(pp + ch); ls[ch] = Elem<T>::ld(pp + c + ch); sc[ch] = expf(fmaxf(ls[ch], DG_MIN_LS)); }
  k[0] = k[1] = k[2] = 0.f;
  if (ar) {
    k[0] = tanhf(Elem<T>::ld(pp + 6)); k[1] = tanhf(Elem<T>::ld(pp + 7)); k[2] = tanhf(Elem<T>::ld(pp + 8));
    loc[0] = fminf(fmaxf(pre[0], -1.f), 1.f);
    pre[1] = pre[1] + k[0] * loc[0];
    loc[1] = fminf(fmaxf(pre[1], -1.f), 1.f);
    pre[2] = pre[2] + k[1] * loc[0] + k[2] * loc[1];
    loc[2] = fminf(fmaxf(pre[2], -1.f), 1.f);
  } else {
    for (int ch = 0; ch < c; ++ch) loc[ch] = fminf(fmaxf(pre[ch], -1.f), 1.f);
  }
}
// gradient of decode: in `gloc` / `gsc` (wrt the clamped locs and the scales), out: raw parameter gradients written to gp
template <typename T>
__device__ __forceinline__ void dgauss_decode_bwd(T* gp, int c, int ar, int gc, const float (&pre)[3], const float (&loc)[3], const float (&k)[3],
                                                  const float (&sc)[3], const float (&ls)[3], const float (&gloc)[3], const float (&gsc)[3]) {
  auto in = [](float v) { return (v >= -1.f && v <= 1.f) ? 1.f : 0.f; };
  float gl[3] = {0.f, 0.f, 0.f}, gk[3] = {0.f, 0.f, 0.f};
  if (ar) {
    const float gb = gloc[2] * in(pre[2]);
    gl[2] = gb; gk[1] = gb * loc[0]; gk[2] = gb * loc[1];
    const float gg = (gloc[1] + gb * k[2]) * in(pre[1]);
    gl[1] = gg; gk[0] = gg * loc[0];
    gl[0] = (gloc[0] + gb * k[1] + gg * k[0]) * in(pre[0]);
  } else {
    for (int ch = 0; ch < c; ++ch) gl[ch] = gloc[ch] * in(pre[ch]);
  }
  for (int ch = 0; ch < c; ++ch) {
    Elem<T>::st(gp + ch, gl[ch]);
    Elem<T>::st(gp + c + ch, ls[ch] >= DG_MIN_LS ? gsc[ch] * sc[ch] : 0.f);
  }
  if (ar) for (int j = 0; j < 3; ++j) Elem<T>::st(gp + 6 + j, gk[j] * (1.f - k[j] * k[j]));
  for (int ch = (ar ? 9 : 2 * c); ch < gc; ++ch) Elem<T>::st(gp + ch, 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void cf_dgauss_bwd_kernel(int n, int h, int w, int c, int ar, View rec, View cf, View x, const float* g_cfx, float gscale,
                                                            View g_rec, View g_cf) {
  const int npix = h * w;
  const int64_t total = (int64_t)n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix), py = px / w, pxx = px % w;
    float rpre[3], rloc[3], rk[3], rsc[3], rls[3], cpre[3], cloc[3], ck[3], csc[3], cls[3];
    dgauss_decode<T>(vptr<T>(rec, b, py, pxx), c, ar, rpre, rloc, rk, rsc, rls);
    dgauss_decode<T>(vptr<T>(cf, b, py, pxx), c, ar, cpre, cloc, ck, csc, cls);
    const T* xp = vptr<T>(x, b, py, pxx);
    float g_rloc[3] = {0.f, 0.f, 0.f}, g_rsc[3] = {0.f, 0.f, 0.f}, g_cloc[3] = {0.f, 0.f, 0.f}, g_csc[3] = {0.f, 0.f, 0.f};
    for (int ch = 0; ch < c; ++ch) {
      const float rs = fmaxf(rsc[ch], 1e-12f);
      const float u = (Elem<T>::ld(xp + ch) - rloc[ch]) / rs;
      const float y = cloc[ch] + csc[ch] * u;
      const float gy = (y >= -1.f && y <= 1.f) ? g_cfx[((int64_t)b * c + ch) * npix + px] * gscale : 0.f;
      g_cloc[ch] = gy;
      g_csc[ch] = gy * u;
      const float gu = gy * csc[ch];
      g_rloc[ch] = -gu / rs;
      g_rsc[ch] = rsc[ch] >= 1e-12f ? -gu * u / rs : 0.f;
    }
    dgauss_decode_bwd<T>(vptr<T>(g_rec, b, py, pxx), c, ar, g_rec.c, rpre, rloc, rk, rsc, rls, g_rloc, g_rsc);
    dgauss_decode_bwd<T>(vptr<T>(g_cf, b, py, pxx), c, ar, g_cf.c, cpre, cloc, ck, csc, cls, g_cloc, g_csc);
  }
}

// ----------------------------------------------------------------------------- logit-space Gaussian (simple_vae.py GaussNet)
// nll (simple_vae.py:215-229): x in [-1,1] -> [0,255] + u, u ~ U[0,1) (dequantisation) -> logit(x / 256) (torch's
// SigmoidTransform.inv: the argument clamped to [tiny, 1 - eps]) -> -log N(. ; loc, exp(max(logscale, -9))), summed; no
// log-determinant term (the reference has none).  u is an injected NHWC view, or Philox uniforms indexed by element.
struct GsP {
  int n, h, w, c;
  View params, x, u, g;
  const uint64_t* rng;
  uint32_t stream_id, pad0;
  float* part;
  const float* coef;
  int coef_stride, pad1;
};

template <typename T>
__device__ __forceinline__ float gs_target(const GsP& p, int b, int y, int x, int c, float xv) {
  float u;
  if (p.u.p != nullptr) u = Elem<T>::ld(vptr<T>(p.u, b, y, x) + c);
  else {
    uint32_t r[4];
    Philox::gen(p.rng[0], p.rng[1], p.stream_id, ((uint64_t)((int64_t)b * p.h + y) * p.w + x) * p.c + c, r);
    u = Philox::u01(r[0]);
  }
  float v = ((xv + 1.0f) * 127.5f + u) * (1.0f / 256.0f);
  v = fminf(fmaxf(v, 1.17549435e-38f), 1.0f - 1.1920929e-07f);
  return logf(v) - log1pf(-v);
}

template <typename T>
__global__ __launch_bounds__(256) void gauss_nll_fwd_kernel(GsP p) {
  __shared__ float sm[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int npix = p.h * p.w;
  float acc = 0.f;
  for (int i = 0; i < LIKE_PIX / 256; ++i) {
    const int px = chunk * LIKE_PIX + i * 256 + threadIdx.x;
    if (px < npix) {
      const int y = px / p.w, x = px % p.w;
      const T* pp = vptr<T>(p.params, b, y, x);
      const T* xp = vptr<T>(p.x, b, y, x);
      for (int c = 0; c < p.c; ++c) {
        const float loc = Elem<T>::ld(pp + c), ls = fmaxf(Elem<T>::ld(pp + p.c + c), DG_MIN_LS);
        const float d = (gs_target<T>(p, b, y, x, c, Elem<T>::ld(xp + c)) - loc) * expf(-ls);
        acc += 0.5f * d * d + ls + 0.9189385332046727f;  // -log N: 0.5 d^2 + log sigma + 0.5 log(2 pi)
      }
    }
  }
  const float tot = block_sum_256(acc, sm);
  if (threadIdx.x == 0) p.part[(int64_t)b * gridDim.x + chunk] = tot;
}

template <typename T>
__global__ __launch_bounds__(256) void gauss_nll_bwd_kernel(GsP p) {
  const int npix = p.h * p.w;
  const int64_t total = (int64_t)p.n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    const int y = px / p.w, x = px % p.w;
    const T* pp = vptr<T>(p.params, b, y, x);
    const T* xp = vptr<T>(p.x, b, y, x);
    T* go = vptr<T>(p.g, b, y, x);
    const float coef = p.coef[(int64_t)b * p.coef_stride];
    for (int c = 0; c < p.c; ++c) {
      const float loc = Elem<T>::ld(pp + c), lsr = Elem<T>::ld(pp + p.c + c), ls = fmaxf(lsr, DG_MIN_LS);
      const float inv = expf(-ls), d = (gs_target<T>(p, b, y, x, c, Elem<T>::ld(xp + c)) - loc) * inv;
      Elem<T>::st(go + c, -coef * d * inv);
      Elem<T>::st(go + p.c + c, lsr >= DG_MIN_LS ? coef * (1.f - d * d) : 0.f);
    }
  }
}

// GaussNet.sample (simple_vae.py:231-238): scale = exp(logscale + log t) in BOTH modes; x = loc (+ scale * N(0,1)) ->
// sigmoid * 256 -> clamp((. - 128) / 128, -1, 1)
template <typename T>
__global__ __launch_bounds__(256) void gauss_sample_kernel(int n, int h, int w, int c, View params, float logt, const uint64_t* rng, uint32_t stream_id, float* xo, float* so) {
  const int npix = h * w;
  const int64_t total = (int64_t)n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    const T* pp = vptr<T>(params, b, px / w, px % w);
    for (int ch = 0; ch < c; ++ch) {
      const int64_t o = ((int64_t)b * c + ch) * npix + px;
      const float sc = expf(fmaxf(Elem<T>::ld(pp + c + ch), DG_MIN_LS) + logt);
      const float noise = rng ? sc * Philox::normal1(rng[0], rng[1], stream_id, (uint64_t)o) : 0.f;
      const float v = 256.0f / (1.0f + expf(-(Elem<T>::ld(pp + ch) + noise)));
      xo[o] = fminf(fmaxf((v - 128.0f) * (1.0f / 128.0f), -1.f), 1.f);
      so[o] = sc;
    }
  }
}

// ----------------------------------------------------------------------------- discretised mixture of logistics
#define DM_NMIX 10
#define DM_MIN_LS (-7.0f)
#define DM_LOG_127_5 4.84810637233259f

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_t(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.f ? 1.f : sigmoid_t(x); }

// log-prob of one (channel, mixture) and its derivatives w.r.t. mean and RAW log-scale
// (hb: half a bin, 1/255 for 8-bit pixels, 1/31 for the reference's low_bit = 5-bit branch; logc = log(1 / (2 hb)), dmol.py:52-60, 88-116)
__device__ __forceinline__ float dm_logprob(float xv, float mean, float ls_raw, float* d_mean, float* d_ls, const float hb, const float logc) {
  const float ls = fmaxf(ls_raw, DM_MIN_LS);
  const float inv = expf(-ls), d = xv - mean;
  const float up = inv * (d + hb), um = inv * (d - hb), mid = inv * d;
  float lp, dup = 0.f, dum = 0.f, dmid = 0.f, dls_direct = 0.f;
  if (xv < -0.999f) {
    lp = up - softplus_t(up);
    dup = 1.f - softplus_grad(up);
  } else if (xv > 0.999f) {
    lp = -softplus_t(um);
    dum = -softplus_grad(um);
  } else {
    const float cp = sigmoid_t(up), cm = sigmoid_t(um), de = cp - cm;
    if (de > 1e-5f) {
      lp = logf(fmaxf(de, 1e-12f));
      dup = cp * (1.f - cp) / de;
      dum = -cm * (1.f - cm) / de;
    } else {
      lp = mid - ls - 2.f * softplus_t(mid) - logc;
      dmid = 1.f - 2.f * softplus_grad(mid);
      dls_direct = -1.f;
    }
  }
  if (d_mean) {
    *d_mean = -(dup + dum + dmid) * inv;
    *d_ls = (ls_raw > DM_MIN_LS) ? (-(dup * up + dum * um + dmid * mid) + dls_direct) : 0.f;
  }
  return lp;
}

template <typename T>
__device__ __forceinline__ void dm_load(const T* lp, float (&l)[100]) {
#pragma unroll
  for (int i = 0; i < 100; ++i) l[i] = Elem<T>::ld(lp + i);
}

struct DmP {
  int n, h, w;
  View logits, x, g;
  float* part;
  const float* coef;
  int coef_stride;
  float hb, logc;
};

// returns log p(x) for the pixel; if G != nullptr writes d(log p)/d logits into G[100]
__device__ __forceinline__ float dm_pixel(const float (&l)[100], const float (&xv)[3], float* G, const float hb = 1.f / 255.f, const float logc = DM_LOG_127_5) {
  float S[DM_NMIX];
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < DM_NMIX; ++m) mx = fmaxf(mx, l[m]);
  float se = 0.f;
#pragma unroll
  for (int m = 0; m < DM_NMIX; ++m) se += expf(l[m] - mx);
  const float lse_logits = mx + logf(se);
  float dmean[3][DM_NMIX], dls[3][DM_NMIX], kk[3][DM_NMIX];
#pragma unroll
  for (int m = 0; m < DM_NMIX; ++m) {
    const float k0 = tanhf(l[10 + 20 + m]), k1 = tanhf(l[10 + 30 + 20 + m]), k2 = tanhf(l[10 + 60 + 20 + m]);
    kk[0][m] = k0; kk[1][m] = k1; kk[2][m] = k2;
    const float mr = l[10 + m];
    const float mg = l[10 + 30 + m] + k0 * xv[0];
    const float mb = l[10 + 60 + m] + k1 * xv[0] + k2 * xv[1];
    float s = l[m] - lse_logits;
    s += dm_logprob(xv[0], mr, l[10 + 10 + m], G ? &dmean[0][m] : nullptr, G ? &dls[0][m] : nullptr, hb, logc);
    s += dm_logprob(xv[1], mg, l[10 + 30 + 10 + m], G ? &dmean[1][m] : nullptr, G ? &dls[1][m] : nullptr, hb, logc);
    s += dm_logprob(xv[2], mb, l[10 + 60 + 10 + m], G ? &dmean[2][m] : nullptr, G ? &dls[2][m] : nullptr, hb, logc);
    S[m] = s;
  }
  float smx = -INFINITY;
#pragma unroll
  for (int m = 0; m < DM_NMIX; ++m) smx = fmaxf(smx, S[m]);
  float sse = 0.f;
#pragma unroll
  for (int m = 0; m < DM_NMIX; ++m) sse += expf(S[m] - smx);
  const float out = smx + logf(sse);
  if (G) {
#pragma unroll
    for (int m = 0; m < DM_NMIX; ++m) {
      const float wm = expf(S[m] - out);          // posterior responsibility
      const float pim = expf(l[m] - lse_logits);  // prior mixture weight
      G[m] = wm - pim;
      G[10 + m] = wm * dmean[0][m];
      G[10 + 10 + m] = wm * dls[0][m];
      G[10 + 20 + m] = wm * dmean[1][m] * xv[0] * (1.f - kk[0][m] * kk[0][m]);
      G[10 + 30 + m] = wm * dmean[1][m];
      G[10 + 30 + 10 + m] = wm * dls[1][m];
      G[10 + 30 + 20 + m] = wm * dmean[2][m] * xv[0] * (1.f - kk[1][m] * kk[1][m]);
      G[10 + 60 + m] = wm * dmean[2][m];
      G[10 + 60 + 10 + m] = wm * dls[2][m];
      G[10 + 60 + 20 + m] = wm * dmean[2][m] * xv[1] * (1.f - kk[2][m] * kk[2][m]);
    }
  }
  return out;
}

template <typename T>
__global__ __launch_bounds__(256) void dmol_nll_fwd_kernel(DmP p) {
  __shared__ float sm[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int npix = p.h * p.w;
  float acc = 0.f;
  for (int i = 0; i < LIKE_PIX / 256; ++i) {
    const int px = chunk * LIKE_PIX + i * 256 + threadIdx.x;
    if (px < npix) {
      const int y = px / p.w, x = px % p.w;
      float l[100], xv[3];
      dm_load<T>(vptr<T>(p.logits, b, y, x), l);
      const T* xp = vptr<T>(p.x, b, y, x);
      xv[0] = Elem<T>::ld(xp); xv[1] = Elem<T>::ld(xp + 1); xv[2] = Elem<T>::ld(xp + 2);
      acc -= dm_pixel(l, xv, nullptr, p.hb, p.logc);
    }
  }
  const float tot = block_sum_256(acc, sm);
  if (threadIdx.x == 0) p.part[(int64_t)b * gridDim.x + chunk] = tot;
}

template <typename T>
__global__ __launch_bounds__(256) void dmol_nll_bwd_kernel(DmP p) {
  const int npix = p.h * p.w;
  const int64_t total = (int64_t)p.n * npix;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    const int y = px / p.w, x = px % p.w;
    float l[100], xv[3], G[100];
    dm_load<T>(vptr<T>(p.logits, b, y, x), l);
    const T* xp = vptr<T>(p.x, b, y, x);
    xv[0] = Elem<T>::ld(xp); xv[1] = Elem<T>::ld(xp + 1); xv[2] = Elem<T>::ld(xp + 2);
    dm_pixel(l, xv, G, p.hb, p.logc);
    const float coef = -p.coef[(int64_t)b * p.coef_stride];
    T* go = vptr<T>(p.g, b, y, x);
#pragma unroll
    for (int i = 0; i < 100; ++i) Elem<T>::st(go + i, coef * G[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dmol_decode_kernel(int n, int h, int w, View logits, int mode, const uint64_t* rng,
                                                          uint32_t stream_id, float logt, float* xo, float* so) {
  const int npix = h * w;
  const int64_t total = (int64_t)n * npix;
  uint64_t seed = 0, off = 0;
  if (mode == 2) { seed = rng[0]; off = rng[1]; }
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
    const int b = (int)(gi / npix), px = (int)(gi % npix);
    float l[100];
    dm_load<T>(vptr<T>(logits, b, px / w, px % w), l);
    float sel[DM_NMIX];
    uint32_t r[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0}, r3[4] = {0, 0, 0, 0};
    if (mode >= 10) {  // top-k (dmol.py:178-188): logits below the k-th largest are switched off, then renormalised
      const int k = mode - 10;
      float srt[DM_NMIX];
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) srt[m] = l[m];
#pragma unroll
      for (int a = 1; a < DM_NMIX; ++a)  // insertion sort, descending
#pragma unroll
        for (int b2 = a; b2 > 0; --b2)
          if (srt[b2] > srt[b2 - 1]) { const float tmp = srt[b2]; srt[b2] = srt[b2 - 1]; srt[b2 - 1] = tmp; }
      float thr = srt[0];
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) if (m == k - 1) thr = srt[m];
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) if (l[m] < thr) l[m] = -INFINITY;
    }
    if (mode == 0 || mode >= 10) {  // soft: softmax weights (dmol.py:170-172)
      float mx = -INFINITY, se = 0.f;
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) mx = fmaxf(mx, l[m]);
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) se += expf(l[m] - mx);
      const float lse = mx + logf(se);
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) sel[m] = expf(l[m] - lse);
    } else {
      if (mode == 2) {  // Gumbel-max with uniforms in [1e-5, 1-1e-5] (dmol.py:128-129)
        Philox::gen(seed, off, stream_id, (uint64_t)gi * 4 + 0, r);
        Philox::gen(seed, off, stream_id, (uint64_t)gi * 4 + 1, r2);
        Philox::gen(seed, off, stream_id, (uint64_t)gi * 4 + 2, r3);
      }
      int am = 0;
      float best = -INFINITY;
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) {
        float v = l[m];
        if (mode == 2) {
          const uint32_t rr = m < 4 ? r[m] : (m < 8 ? r2[m - 4] : r3[m - 8]);
          const float u = 1e-5f + (1.f - 2e-5f) * Philox::u01(rr);
          v -= logf(-logf(u));
        }
        if (v > best) { best = v; am = m; }
      }
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) sel[m] = (m == am) ? 1.f : 0.f;
    }
    float mu[3], ls[3], co[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float a = 0.f, s = 0.f, k = 0.f;
#pragma unroll
      for (int m = 0; m < DM_NMIX; ++m) {
        a += l[10 + 30 * c + m] * sel[m];
        s += l[10 + 30 * c + 10 + m] * sel[m];
        k += tanhf(l[10 + 30 * c + 20 + m]) * sel[m];
      }
      mu[c] = a; ls[c] = fmaxf(s, DM_MIN_LS); co[c] = k;
    }
    if (mode == 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        ls[c] += logt;
        const float u = 1e-5f + (1.f - 2e-5f) * Philox::u01(c == 0 ? r3[2] : (c == 1 ? r3[3] : r2[3] ^ 0x9E3779B9u));
        mu[c] += expf(ls[c]) * (logf(u) - logf(1.f - u));
      }
    }
    const float x0 = fminf(fmaxf(mu[0], -1.f), 1.f);
    const float x1 = fminf(fmaxf(mu[1] + co[0] * x0, -1.f), 1.f);
    const float x2 = fminf(fmaxf(mu[2] + co[1] * x0 + co[2] * x1, -1.f), 1.f);
    const float xs[3] = {x0, x1, x2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t o = ((int64_t)b * 3 + c) * npix + px;
      xo[o] = xs[c];
      so[o] = expf(ls[c]);
    }
  }
}

// ----------------------------------------------------------------------------- ELBO assembly + cf pixels
// One workgroup of 16 waves: wave w sums the partials of samples w, w+16, ... (lanes stride over the chunks, fixed shuffle
// tree), then thread 0 adds the per-sample values in sample order -- deterministic, and ~100x shorter than the one-thread-
// per-sample serial chains of round 1 (112 us on the critical path between forward and backward at ukbb192).
// `beta_dev` (optional) overrides `beta` with a value read from device memory, so a captured hipGraph follows the KL
// warm-up schedule without re-capture.
__global__ __launch_bounds__(1024) void elbo_finalize_kernel(int n, const float* nll_part, int nll_count, float nll_div,
                                                             const float* kl_part, int kl_count, float kl_div, float beta,
                                                             const float* beta_dev, float* out3) {
  extern __shared__ float per[];  // [2*n]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = wave; b < n; b += 16) {
    float a = 0.f, k = 0.f;
    for (int j = lane; j < nll_count; j += 64) a += nll_part[(int64_t)b * nll_count + j];
    for (int j = lane; j < kl_count; j += 64) k += kl_part[(int64_t)b * kl_count + j];
    a = wave_sum(a);
    k = wave_sum(k);
    if (lane == 0) { per[b] = a / nll_div; per[n + b] = k / kl_div; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, k = 0.f;
    for (int b = 0; b < n; ++b) { a += per[b]; k += per[n + b]; }
    a /= (float)n; k /= (float)n;
    if (beta_dev) beta = *beta_dev;
    out3[0] = a + beta * k; out3[1] = a; out3[2] = k;
  }
}

// free-bits ELBO (vae.py:443-457): kl = sum_j max(fb, mean_b S[b][j]) / kl_div over all (layer, channel) columns j;
// chan_mask[j] = d max(fb, m_j) / d m_j  (1 above the threshold, 0 below, 1/2 on a tie as torch.maximum).
// The NLL is assembled exactly as in elbo_finalize_kernel.
__global__ __launch_bounds__(256) void elbo_finalize_fb_kernel(int n, const float* nll_part, int nll_count, float nll_div,
                                                               const float* kl_bc, int ncol, float kl_div, float free_bits,
                                                               float beta, const float* beta_dev, float* out3, float* chan_mask) {
  extern __shared__ float per[];  // [n + 256]
  float* colsum = per + n;
  for (int b = threadIdx.x; b < n; b += 256) {
    float a = 0.f;
    for (int j = 0; j < nll_count; ++j) a += nll_part[(int64_t)b * nll_count + j];
    per[b] = a / nll_div;
  }
  float k = 0.f;
  for (int j = threadIdx.x; j < ncol; j += 256) {
    float m = 0.f;
    for (int b = 0; b < n; ++b) m += kl_bc[(int64_t)b * ncol + j];
    m /= (float)n;
    k += fmaxf(free_bits, m);
    chan_mask[j] = m > free_bits ? 1.f : (m == free_bits ? 0.5f : 0.f);
  }
  colsum[threadIdx.x] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, kk = 0.f;
    for (int b = 0; b < n; ++b) a += per[b];
    for (int t = 0; t < 256; ++t) kk += colsum[t];
    a /= (float)n; kk /= kl_div;
    if (beta_dev) beta = *beta_dev;
    out3[0] = a + beta * kk; out3[1] = a; out3[2] = kk;
  }
}

__global__ __launch_bounds__(256) void cf_pixels_kernel(int64_t count, const float* x, const float* rec_loc, const float* rec_scale,
                                                        const float* cf_loc, const float* cf_scale, float* cf_x, float* sum_x,
                                                        float* sum_x2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    const float u = (x[i] - rec_loc[i]) / fmaxf(rec_scale[i], 1e-12f);
    const float v = fminf(fmaxf(cf_loc[i] + cf_scale[i] * u, -1.f), 1.f);
    cf_x[i] = v;
    if (sum_x) sum_x[i] += v;
    if (sum_x2) sum_x2[i] += v * v;
  }
}

static inline int like_grid(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace cgen

using namespace cgen;

extern "C" int cgen_like_chunks(int32_t h, int32_t w) { return ceil_div((int64_t)h * w, LIKE_PIX); }

extern "C" int cgen_dgauss_nll_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x,
                                   float* nll_part, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dgauss_nll_fwd: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && params.p && x.p && nll_part && params.c >= 2 * c, "cgen_dgauss_nll_fwd: bad args");
  DgP p;
  memset(&p, 0, sizeof(p));
  p.n = n; p.h = h; p.w = w; p.c = c; p.ar = (c == 3 && params.c >= 9); p.params = mk(params); p.x = mk(x); p.part = nll_part;
  dim3 grid(cgen_like_chunks(h, w), n);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dgauss_nll_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(dgauss_nll_fwd_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_dgauss_nll_fwd");
}

extern "C" int cgen_dgauss_nll_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x,
                                   const float* coef_dev, int32_t coef_stride, cgen_view g_params, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dgauss_nll_bwd: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && params.p && x.p && coef_dev && g_params.p && params.c >= 2 * c && g_params.c >= (c == 3 && params.c >= 9 ? 9 : 2 * c), "cgen_dgauss_nll_bwd: bad args");
  DgP p;
  memset(&p, 0, sizeof(p));
  p.n = n; p.h = h; p.w = w; p.c = c; p.ar = (c == 3 && params.c >= 9); p.params = mk(params); p.x = mk(x); p.g = mk(g_params); p.coef = coef_dev; p.coef_stride = coef_stride;
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dgauss_nll_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(dgauss_nll_bwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_dgauss_nll_bwd");
}

extern "C" int cgen_dgauss_sample(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, float logt,
                                  const uint64_t* rng, uint32_t stream_id, float* x_nchw, float* scale_nchw, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dgauss_sample: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && params.p && x_nchw && scale_nchw && params.c >= 2 * c, "cgen_dgauss_sample: bad args");
  const int ar = (c == 3 && params.c >= 9);
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dgauss_sample_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, ar, mk(params), logt, rng, stream_id, x_nchw, scale_nchw);
  else hipLaunchKernelGGL(dgauss_sample_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, ar, mk(params), logt, rng, stream_id, x_nchw, scale_nchw);
  return check_launch("cgen_dgauss_sample");
}

extern "C" int cgen_dgauss_params(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x, float logt,
                                  float* loc_nchw, float* logscale_nchw, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dgauss_params: bad dtype");
  CGEN_REQUIRE(params.p && loc_nchw && logscale_nchw && c >= 1 && c <= 3 && params.c >= (c == 3 ? 9 : 2 * c) && (!x.p || x.c == c), "cgen_dgauss_params: bad args");
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dgauss_params_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(params), mk(x), logt, loc_nchw, logscale_nchw);
  else hipLaunchKernelGGL(dgauss_params_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(params), mk(x), logt, loc_nchw, logscale_nchw);
  return check_launch("cgen_dgauss_params");
}

extern "C" int cgen_dmol_nll_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view logits, cgen_view x, float* nll_part,
                                 cgen_stream_t stream) {
  const bool low_bit = (dtype & CGEN_DMOL_LOW_BIT) != 0;
  dtype &= ~CGEN_DMOL_LOW_BIT;
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dmol_nll_fwd: bad dtype");
  CGEN_REQUIRE(logits.p && x.p && nll_part && logits.c == 100 && x.c == 3, "cgen_dmol_nll_fwd: bad args");
  DmP p;
  memset(&p, 0, sizeof(p));
  p.hb = low_bit ? 1.f / 31.f : 1.f / 255.f; p.logc = low_bit ? 2.7408400239252009f : DM_LOG_127_5;  // log 15.5 | log 127.5
  p.n = n; p.h = h; p.w = w; p.logits = mk(logits); p.x = mk(x); p.part = nll_part;
  dim3 grid(cgen_like_chunks(h, w), n);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dmol_nll_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(dmol_nll_fwd_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_dmol_nll_fwd");
}

extern "C" int cgen_dmol_nll_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view logits, cgen_view x,
                                 const float* coef_dev, int32_t coef_stride, cgen_view g_logits, cgen_stream_t stream) {
  const bool low_bit = (dtype & CGEN_DMOL_LOW_BIT) != 0;
  dtype &= ~CGEN_DMOL_LOW_BIT;
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dmol_nll_bwd: bad dtype");
  CGEN_REQUIRE(logits.p && x.p && coef_dev && g_logits.p && logits.c == 100 && g_logits.c == 100, "cgen_dmol_nll_bwd: bad args");
  DmP p;
  memset(&p, 0, sizeof(p));
  p.hb = low_bit ? 1.f / 31.f : 1.f / 255.f; p.logc = low_bit ? 2.7408400239252009f : DM_LOG_127_5;
  p.n = n; p.h = h; p.w = w; p.logits = mk(logits); p.x = mk(x); p.g = mk(g_logits); p.coef = coef_dev; p.coef_stride = coef_stride;
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dmol_nll_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(dmol_nll_bwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_dmol_nll_bwd");
}

extern "C" int cgen_dmol_decode(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view logits, int32_t mode,
                                const uint64_t* rng, uint32_t stream_id, float logt, float* x_nchw, float* scale_nchw,
                                cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_dmol_decode: bad dtype");
  CGEN_REQUIRE(logits.p && logits.c == 100 && x_nchw && scale_nchw && ((mode >= 0 && mode <= 2) || (mode >= 11 && mode <= 19)) && (mode != 2 || rng), "cgen_dmol_decode: bad args");
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(dmol_decode_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, mk(logits), mode, rng, stream_id, logt, x_nchw, scale_nchw);
  else hipLaunchKernelGGL(dmol_decode_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, mk(logits), mode, rng, stream_id, logt, x_nchw, scale_nchw);
  return check_launch("cgen_dmol_decode");
}

extern "C" int cgen_elbo_finalize(int32_t n, const float* nll_part, int32_t nll_count, float nll_div, const float* kl_part,
                                  int32_t kl_count, float kl_div, float beta, const float* beta_dev, float* out3, cgen_stream_t stream) {
  CGEN_REQUIRE(n > 0 && n <= 8192 && nll_part && out3 && (kl_count == 0 || kl_part), "cgen_elbo_finalize: bad args");
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(1024), 2 * n * sizeof(float), (hipStream_t)stream, n, nll_part, nll_count,
                     nll_div, kl_part, kl_count, kl_div, beta, beta_dev, out3);
  return check_launch("cgen_elbo_finalize");
}

extern "C" int cgen_cf_pixels(int64_t count, const float* x, const float* rec_loc, const float* rec_scale, const float* cf_loc,
                              const float* cf_scale, float* cf_x, float* sum_x, float* sum_x2, cgen_stream_t stream) {
  CGEN_REQUIRE(count >= 0 && x && rec_loc && rec_scale && cf_loc && cf_scale && cf_x, "cgen_cf_pixels: null");
  if (count == 0) return CGEN_OK;
  hipLaunchKernelGGL(cf_pixels_kernel, dim3(like_grid(count)), dim3(256), 0, (hipStream_t)stream, count, x, rec_loc, rec_scale,
                     cf_loc, cf_scale, cf_x, sum_x, sum_x2);
  return check_launch("cgen_cf_pixels");
}

extern "C" int cgen_cf_dgauss_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view rec_params, cgen_view cf_params,
                                  cgen_view x, const float* g_cfx_nchw, float gscale, cgen_view g_rec_params, cgen_view g_cf_params,
                                  cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_cf_dgauss_bwd: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && rec_params.p && cf_params.p && x.p && g_cfx_nchw && g_rec_params.p && g_cf_params.p &&
                   rec_params.c >= 2 * c && cf_params.c == rec_params.c && g_rec_params.c == rec_params.c && g_cf_params.c == rec_params.c,
               "cgen_cf_dgauss_bwd: bad args");
  const int ar = (c == 3 && rec_params.c >= 9);
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(cf_dgauss_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, ar, mk(rec_params), mk(cf_params), mk(x), g_cfx_nchw, gscale, mk(g_rec_params), mk(g_cf_params));
  else hipLaunchKernelGGL(cf_dgauss_bwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, ar, mk(rec_params), mk(cf_params), mk(x), g_cfx_nchw, gscale, mk(g_rec_params), mk(g_cf_params));
  return check_launch("cgen_cf_dgauss_bwd");
}

extern "C" int cgen_elbo_finalize_fb(int32_t n, const float* nll_part, int32_t nll_count, float nll_div, const float* kl_bc, int32_t ncol,
                                     float kl_div, float free_bits, float beta, const float* beta_dev, float* out3, float* chan_mask,
                                     cgen_stream_t stream) {
  CGEN_REQUIRE(n > 0 && n <= 8192 && nll_part && nll_count > 0 && kl_bc && ncol > 0 && out3 && chan_mask, "cgen_elbo_finalize_fb: bad args");
  hipLaunchKernelGGL(elbo_finalize_fb_kernel, dim3(1), dim3(256), (n + 256) * sizeof(float), (hipStream_t)stream, n, nll_part, nll_count,
                     nll_div, kl_bc, ncol, kl_div, free_bits, beta, beta_dev, out3, chan_mask);
  return check_launch("cgen_elbo_finalize_fb");
}

extern "C" int cgen_gauss_nll_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x, cgen_view u,
                                  const uint64_t* rng, uint32_t stream_id, float* nll_part, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_gauss_nll_fwd: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && params.p && x.p && nll_part && params.c >= 2 * c && (u.p || rng), "cgen_gauss_nll_fwd: bad args");
  GsP p;
  memset(&p, 0, sizeof(p));
  p.n = n; p.h = h; p.w = w; p.c = c; p.params = mk(params); p.x = mk(x); p.u = mk(u); p.rng = rng; p.stream_id = stream_id; p.part = nll_part;
  dim3 grid(cgen_like_chunks(h, w), n);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(gauss_nll_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gauss_nll_fwd_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_gauss_nll_fwd");
}

extern "C" int cgen_gauss_nll_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x, cgen_view u,
                                  const uint64_t* rng, uint32_t stream_id, const float* coef_dev, int32_t coef_stride,
                                  cgen_view g_params, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_gauss_nll_bwd: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && params.p && x.p && coef_dev && g_params.p && params.c >= 2 * c && g_params.c >= 2 * c && (u.p || rng),
               "cgen_gauss_nll_bwd: bad args");
  GsP p;
  memset(&p, 0, sizeof(p));
  p.n = n; p.h = h; p.w = w; p.c = c; p.params = mk(params); p.x = mk(x); p.u = mk(u); p.g = mk(g_params); p.rng = rng; p.stream_id = stream_id;
  p.coef = coef_dev; p.coef_stride = coef_stride;
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(gauss_nll_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gauss_nll_bwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_gauss_nll_bwd");
}

extern "C" int cgen_gauss_sample(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, float logt,
                                 const uint64_t* rng, uint32_t stream_id, float* x_nchw, float* scale_nchw, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_gauss_sample: bad dtype");
  CGEN_REQUIRE((c == 1 || c == 3) && params.p && x_nchw && scale_nchw && params.c >= 2 * c, "cgen_gauss_sample: bad args");
  const int grid = like_grid((int64_t)n * h * w);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(gauss_sample_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(params), logt, rng, stream_id, x_nchw, scale_nchw);
  else hipLaunchKernelGGL(gauss_sample_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(params), logt, rng, stream_id, x_nchw, scale_nchw);
  return check_launch("cgen_gauss_sample");
}
