// Per-stochastic-layer Gaussian kernels: fused reparameterise + KL (forward / backward), prior sampling and the
// counterfactual mediator mix.  HBM-bound: one read of (q_loc,q_ls,p_loc,p_ls[,eps]), one write of z, and a
// deterministic two-stage per-sample reduction of the KL (wave shuffles -> LDS -> one partial per block).
#include <initializer_list>

#include "common.h"

namespace cgen {

}  // namespace cgen
#include "latent_bodies.inc"
namespace cgen {

__global__ __launch_bounds__(256) void reparam_kl_fwd_vec8_kernel(LatP p) {
  __shared__ float sm[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  uint64_t seed = 0, off = 0;
  if (!p.eps_in.p) { seed = p.rng[0]; off = p.rng[1]; }
  const float kl_acc = reparam_kl_fwd_vec8_item(p, b, chunk, threadIdx.x, seed, off);
  const float tot = block_sum_256(kl_acc, sm);
  if (threadIdx.x == 0) p.kl_part[(int64_t)b * p.kl_stride + chunk] = tot;
}

__global__ __launch_bounds__(256) void reparam_kl_bwd_vec8_kernel(LatBwdP p) {
  if ((int)blockIdx.x >= p.main_blocks) {  // rider blocks
    const int64_t tot = (int64_t)p.n * p.h * p.w * (p.ride_c >> 3);
    const int nb = gridDim.x - p.main_blocks;
    for (int64_t g = (int64_t)(blockIdx.x - p.main_blocks) * 256 + threadIdx.x; g < tot; g += (int64_t)nb * 256) reparam_ride_item(p, g);
    return;
  }
  const int64_t total = (int64_t)p.n * ((p.h * p.w * p.c) >> 3);
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)p.main_blocks * 256) reparam_kl_bwd_vec8_item(p, g);
}

template <typename T>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(LatP p) {
  __shared__ float sm[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int per = p.h * p.w * p.c;
  uint64_t seed = 0, off = 0;
  if (!p.eps_in.p) { seed = p.rng[0]; off = p.rng[1]; }
  float kl_acc = 0.f;
  for (int i = 0; i < LAT_CHUNK / 256; ++i) {
    const int e = chunk * LAT_CHUNK + i * 256 + threadIdx.x;
    if (e < per) {
      int y, x, ch;
      lat_decode(e, p.w, p.c, y, x, ch);
      const float ql = Elem<T>::ld(vptr<T>(p.q_loc, b, y, x) + ch);
      const float qs = Elem<T>::ld(vptr<T>(p.q_ls, b, y, x) + ch) + p.logt;
      const float pl = Elem<T>::ld(vptr<T>(p.p_loc, b, y, x) + ch);
      const float ps = Elem<T>::ld(vptr<T>(p.p_ls, b, y, x) + ch) + p.logt;
      float eps;
      if (p.eps_in.p) eps = Elem<T>::ld(vptr<T>(p.eps_in, b, y, x) + ch);
      else eps = Philox::normal1(seed, off, p.stream_id, (uint64_t)b * per + e);
      const float sq = expf(qs);
      Elem<T>::st(vptr<T>(p.z, b, y, x) + ch, ql + sq * eps);
      if (p.eps_out.p) Elem<T>::st(vptr<T>(p.eps_out, b, y, x) + ch, eps);
      // vae.py:18-25, same association as the reference: -0.5 + p - q + 0.5*(e^{q}^2 + d^2)/e^{p}^2
      const float sp = expf(ps);
      const float d = ql - pl;
      kl_acc += -0.5f + ps - qs + 0.5f * (sq * sq + d * d) / (sp * sp);
    }
  }
  const float tot = block_sum_256(kl_acc, sm);
  if (threadIdx.x == 0) p.kl_part[(int64_t)b * p.kl_stride + chunk] = tot;
}

template <typename T>
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(LatBwdP p) {
  const int per = p.h * p.w * p.c;
  const int64_t total = (int64_t)p.n * per;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int b = (int)(g / per), e = (int)(g % per);
    int y, x, ch;
    lat_decode(e, p.w, p.c, y, x, ch);
    const float ql = Elem<T>::ld(vptr<T>(p.q_loc, b, y, x) + ch);
    const float qs = Elem<T>::ld(vptr<T>(p.q_ls, b, y, x) + ch) + p.logt;
    const float pl = Elem<T>::ld(vptr<T>(p.p_loc, b, y, x) + ch);
    const float ps = Elem<T>::ld(vptr<T>(p.p_ls, b, y, x) + ch) + p.logt;
    const float k = p.coef[(int64_t)b * p.coef_stride] * (p.chan_scale ? p.chan_scale[ch] : 1.f);
    const float e2q = expf(2.f * qs), ie2p = expf(-2.f * ps);
    const float d = ql - pl;
    float gql = k * d * ie2p;
    float gqs = k * (e2q * ie2p - 1.f);
    const float gpl = -k * d * ie2p;
    const float gps = k * (1.f - (e2q + d * d) * ie2p);
    if (p.gz.p) {
      const float gzv = Elem<T>::ld(vptr<T>(p.gz, b, y, x) + ch);
      const float zv = Elem<T>::ld(vptr<T>(p.z, b, y, x) + ch);
      gql += gzv;
      gqs += gzv * (zv - ql);  // dz/dq_ls = e^{q_ls} eps = z - q_loc
    }
    T* o;
    o = vptr<T>(p.g_q_loc, b, y, x) + ch; Elem<T>::st(o, p.acc_q ? Elem<T>::ld(o) + gql : gql);
    o = vptr<T>(p.g_q_ls, b, y, x) + ch;  Elem<T>::st(o, p.acc_q ? Elem<T>::ld(o) + gqs : gqs);
    o = vptr<T>(p.g_p_loc, b, y, x) + ch; Elem<T>::st(o, p.acc_p ? Elem<T>::ld(o) + gpl : gpl);
    o = vptr<T>(p.g_p_ls, b, y, x) + ch;  Elem<T>::st(o, p.acc_p ? Elem<T>::ld(o) + gps : gps);
  }
}

// ============================================================================= latent layer: reparameterise + KL + z_proj in one launch
// The decoder's stochastic layer (vae.py:264-294) is  z = q_loc + exp(q_ls) eps;  kl;  h' = z_proj(cat[z, pa]) + h + p_feat.
// z_proj is a 1x1 conv with K = z_dim + ctx <= 32: ONE MFMA K-step.  On the launch-bound decoder chain (~5.5 us of
// every dependent launch is dispatch) the separate z_proj launch -- and, backward, its data-gradient launch -- cost more than
// their work.  Here a wave owns 16 pixels: lane (pixel fr = lane & 15, K group fg = lane >> 4) computes the 8 z channels
// fg * 8.. of its pixel (fg < 2) or fetches 8 parent channels (fg >= 2) -- exactly the B fragment of
// v_mfma_f32_16x16x32_bf16 -- and the weight image rows are the A fragments: no LDS, no shuffles.  z is stored once (bf16) for
// z_feat_proj, the weight gradients and the backward pass, with the same Philox indexing as the unfused kernel.
typedef float lat_f32x4 __attribute__((ext_vector_type(4)));
__device__ uint4 lat_zero16[2];  // zeros: source of absent operands (loads are never issued under a condition)

template <typename T>
__global__ __launch_bounds__(256) void sample_gaussian_kernel(int n, int h, int w, int c, View loc, View ls, View eps_in,
                                                              const uint64_t* rng, uint32_t stream_id, float logt, View z) {
  const int per = h * w * c;
  const int64_t total = (int64_t)n * per;
  uint64_t seed = 0, off = 0;
  if (!eps_in.p) { seed = rng[0]; off = rng[1]; }
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int b = (int)(g / per), e = (int)(g % per);
    int y, x, ch;
    lat_decode(e, w, c, y, x, ch);
    const float l = Elem<T>::ld(vptr<T>(loc, b, y, x) + ch);
    const float s = Elem<T>::ld(vptr<T>(ls, b, y, x) + ch) + logt;
    const float eps = eps_in.p ? Elem<T>::ld(vptr<T>(eps_in, b, y, x) + ch) : Philox::normal1(seed, off, stream_id, (uint64_t)g);
    Elem<T>::st(vptr<T>(z, b, y, x) + ch, l + expf(s) * eps);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mediator_kernel(int n, int h, int w, int c, View z, View q_loc, View q_ls, View p_loc,
                                                       View p_ls, float alpha, float t, float logt, int linear_var, View out) {
  const int per = h * w * c;
  const int64_t total = (int64_t)n * per;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int b = (int)(g / per), e = (int)(g % per);
    int y, x, ch;
    lat_decode(e, w, c, y, x, ch);
    const float zv = Elem<T>::ld(vptr<T>(z, b, y, x) + ch);
    const float ql = Elem<T>::ld(vptr<T>(q_loc, b, y, x) + ch);
    const float qsc = expf(Elem<T>::ld(vptr<T>(q_ls, b, y, x) + ch) + logt);
    const float pl = Elem<T>::ld(vptr<T>(p_loc, b, y, x) + ch);
    const float psc = expf(Elem<T>::ld(vptr<T>(p_ls, b, y, x) + ch) + logt);
    const float u = (zv - ql) / qsc;
    const float r_loc = alpha * ql + (1.f - alpha) * pl;
    // vae.py:505 weights the variances by a^2 / (1-a)^2; simple_vae.py:385 by a / (1-a)
    const float wq = linear_var ? alpha : alpha * alpha, wp = linear_var ? (1.f - alpha) : (1.f - alpha) * (1.f - alpha);
    float r_scale = sqrtf(wq * qsc * qsc + wp * psc * psc);
    if (t > 0.f) r_scale *= t;
    Elem<T>::st(vptr<T>(out, b, y, x) + ch, r_loc + r_scale * u);
  }
}

// free bits (vae.py:443-449): S[b][c] = sum_{h,w} KL(q || p)[b,h,w,c] of one stochastic layer.  One workgroup per sample;
// a thread owns channel (tid % c) of pixels tid / c, tid / c + 256 / c, ... (c divides 256), fixed-order LDS tree.
template <typename T>
__global__ __launch_bounds__(256) void kl_channel_sums_kernel(int h, int w, int c, View q_loc, View q_ls, View p_loc, View p_ls, float logt,
                                                              float* out, int out_stride) {
  __shared__ float red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ch = tid % c, lanes = 256 / c, slot = tid / c;
  float acc = 0.f;
  for (int px = slot; px < h * w; px += lanes) {
    const int y = px / w, x = px - y * w;
    const float ql = Elem<T>::ld(vptr<T>(q_loc, b, y, x) + ch);
    const float qs = Elem<T>::ld(vptr<T>(q_ls, b, y, x) + ch) + logt;
    const float pl = Elem<T>::ld(vptr<T>(p_loc, b, y, x) + ch);
    const float ps = Elem<T>::ld(vptr<T>(p_ls, b, y, x) + ch) + logt;
    const float d = ql - pl;
    acc += -0.5f + ps - qs + 0.5f * (expf(2.f * qs) + d * d) * expf(-2.f * ps);
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = lanes >> 1; s > 0; s >>= 1) {
    if (slot < s) red[tid] += red[tid + s * c];
    __syncthreads();
  }
  if (slot == 0) out[(int64_t)b * out_stride + ch] = red[tid];
}

// element-wise KL(q || p) of diagonal Gaussians on flat f32 arrays (vae.py:14-25), no reduction
__global__ __launch_bounds__(256) void gaussian_kl_map_kernel(int64_t total, const float* q_loc, const float* q_ls, const float* p_loc,
                                                              const float* p_ls, float* out) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const float ql = q_loc[g], qs = q_ls[g], pl = p_loc[g], ps = p_ls[g];
    const float eq = expf(qs), ep = expf(ps), d = ql - pl;
    out[g] = -0.5f + ps - qs + 0.5f * (eq * eq + d * d) / (ep * ep);
  }
}

static inline int lat_grid(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace cgen

using namespace cgen;

// the 8-channel fast paths need 16-byte aligned views, 8-granular channel counts and 32-bit offsets
static inline bool lat_vec8_ok(int n, int c, std::initializer_list<cgen_view> vs) {
  if (c % 8) return false;
  for (const cgen_view& v : vs) {
    if (!v.p) continue;
    if (!cgen::vec16_ok(v, 2)) return false;
    if ((int64_t)n * v.sn >= ((int64_t)1 << 31)) return false;
  }
  return true;
}

extern "C" int cgen_reparam_kl_chunks(int32_t h, int32_t w, int32_t c) { return ceil_div((int64_t)h * w * c, LAT_CHUNK); }

extern "C" int cgen_reparam_kl_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                                   cgen_view p_loc, cgen_view p_ls, cgen_view eps_in, const uint64_t* rng, uint32_t stream_id,
                                   float logt, cgen_view z, cgen_view eps_out, float* kl_part, int32_t kl_stride,
                                   cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_reparam_kl_fwd: bad dtype");
  CGEN_REQUIRE(q_loc.p && q_ls.p && p_loc.p && p_ls.p && z.p && kl_part, "cgen_reparam_kl_fwd: null view");
  CGEN_REQUIRE(eps_in.p || rng, "cgen_reparam_kl_fwd: need eps or rng");
  LatP p;
  p.n = n; p.h = h; p.w = w; p.c = c;
  p.q_loc = mk(q_loc); p.q_ls = mk(q_ls); p.p_loc = mk(p_loc); p.p_ls = mk(p_ls);
  p.eps_in = mk(eps_in); p.z = mk(z); p.eps_out = mk(eps_out);
  p.rng = rng; p.stream_id = stream_id; p.logt = logt; p.kl_part = kl_part; p.kl_stride = kl_stride;
  dim3 grid(cgen_reparam_kl_chunks(h, w, c), n);
  if (dtype == CGEN_F32) hipLaunchKernelGGL(reparam_kl_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (lat_vec8_ok(n, c, {q_loc, q_ls, p_loc, p_ls, eps_in, z, eps_out})) hipLaunchKernelGGL(reparam_kl_fwd_vec8_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(reparam_kl_fwd_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("cgen_reparam_kl_fwd");
}

static int reparam_kl_bwd_impl(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls, cgen_view p_loc,
                               cgen_view p_ls, cgen_view z, float logt, cgen_view gz, const float* kl_coef_dev, int32_t coef_stride,
                               const float* kl_chan_scale, cgen_view g_q_loc, cgen_view g_q_ls, cgen_view g_p_loc, cgen_view g_p_ls,
                               int32_t acc_q, int32_t acc_p, const cgen_view* ride_src, const cgen_view* ride_dst, int32_t ride_acc,
                               cgen_stream_t stream, const char* who) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "%s: bad dtype", who);
  CGEN_REQUIRE(q_loc.p && q_ls.p && p_loc.p && p_ls.p && kl_coef_dev && g_q_loc.p && g_q_ls.p && g_p_loc.p && g_p_ls.p, "%s: null view", who);
  CGEN_REQUIRE(!gz.p || z.p, "%s: gz given without z", who);
  LatBwdP p;
  memset(&p, 0, sizeof(p));
  p.n = n; p.h = h; p.w = w; p.c = c;
  p.q_loc = mk(q_loc); p.q_ls = mk(q_ls); p.p_loc = mk(p_loc); p.p_ls = mk(p_ls); p.z = mk(z); p.gz = mk(gz);
  p.g_q_loc = mk(g_q_loc); p.g_q_ls = mk(g_q_ls); p.g_p_loc = mk(g_p_loc); p.g_p_ls = mk(g_p_ls);
  p.coef = kl_coef_dev; p.chan_scale = kl_chan_scale; p.coef_stride = coef_stride; p.acc_q = acc_q; p.acc_p = acc_p; p.logt = logt;
  const int grid = lat_grid((int64_t)n * h * w * c);
  const bool vec8 = dtype == CGEN_F16 && lat_vec8_ok(n, c, {q_loc, q_ls, p_loc, p_ls, z, gz, g_q_loc, g_q_ls, g_p_loc, g_p_ls});
  if (ride_src) {
    CGEN_REQUIRE(vec8 && ride_dst && ride_src->p && ride_dst->p && ride_src->c == ride_dst->c && ride_src->c % 8 == 0 &&
                     lat_vec8_ok(n, ride_src->c, {*ride_src, *ride_dst}),
                 "%s: the rider needs the 16-byte bf16 path (8-channel multiples, 16-byte aligned views)", who);
    p.ride_src = mk(*ride_src); p.ride_dst = mk(*ride_dst); p.ride_c = ride_src->c; p.ride_acc = ride_acc;
  }
  if (dtype == CGEN_F32) hipLaunchKernelGGL(reparam_kl_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else if (vec8) {
    p.main_blocks = lat_grid((int64_t)n * h * w * c / 8);
    const int ride_blocks = ride_src ? lat_grid((int64_t)n * h * w * ride_src->c / 8) : 0;
    hipLaunchKernelGGL(reparam_kl_bwd_vec8_kernel, dim3(p.main_blocks + ride_blocks), dim3(256), 0, (hipStream_t)stream, p);
  } else hipLaunchKernelGGL(reparam_kl_bwd_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch(who);
}

extern "C" int cgen_reparam_kl_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                                   cgen_view p_loc, cgen_view p_ls, cgen_view z, float logt, cgen_view gz,
                                   const float* kl_coef_dev, int32_t coef_stride, const float* kl_chan_scale, cgen_view g_q_loc,
                                   cgen_view g_q_ls, cgen_view g_p_loc, cgen_view g_p_ls, int32_t acc_q, int32_t acc_p,
                                   cgen_stream_t stream) {
  return reparam_kl_bwd_impl(dtype, n, h, w, c, q_loc, q_ls, p_loc, p_ls, z, logt, gz, kl_coef_dev, coef_stride, kl_chan_scale, g_q_loc, g_q_ls,
                             g_p_loc, g_p_ls, acc_q, acc_p, nullptr, nullptr, 0, stream, "cgen_reparam_kl_bwd");
}

extern "C" int cgen_reparam_kl_bwd_rider(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                                         cgen_view p_loc, cgen_view p_ls, cgen_view z, float logt, cgen_view gz,
                                         const float* kl_coef_dev, int32_t coef_stride, const float* kl_chan_scale, cgen_view g_q_loc,
                                         cgen_view g_q_ls, cgen_view g_p_loc, cgen_view g_p_ls, int32_t acc_q, int32_t acc_p,
                                         cgen_view ride_src, cgen_view ride_dst, int32_t ride_acc, cgen_stream_t stream) {
  return reparam_kl_bwd_impl(dtype, n, h, w, c, q_loc, q_ls, p_loc, p_ls, z, logt, gz, kl_coef_dev, coef_stride, kl_chan_scale, g_q_loc, g_q_ls,
                             g_p_loc, g_p_ls, acc_q, acc_p, &ride_src, &ride_dst, ride_acc, stream, "cgen_reparam_kl_bwd_rider");
}


extern "C" int cgen_sample_gaussian(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view loc, cgen_view ls,
                                    cgen_view eps_in, const uint64_t* rng, uint32_t stream_id, float logt, cgen_view z,
                                    cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_sample_gaussian: bad dtype");
  CGEN_REQUIRE(loc.p && ls.p && z.p && (eps_in.p || rng), "cgen_sample_gaussian: bad args");
  const int grid = lat_grid((int64_t)n * h * w * c);
  if (dtype == CGEN_F32)
    hipLaunchKernelGGL(sample_gaussian_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(loc), mk(ls), mk(eps_in), rng, stream_id, logt, mk(z));
  else
    hipLaunchKernelGGL(sample_gaussian_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(loc), mk(ls), mk(eps_in), rng, stream_id, logt, mk(z));
  return check_launch("cgen_sample_gaussian");
}

extern "C" int cgen_mediator_mix(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view z, cgen_view q_loc,
                                 cgen_view q_ls, cgen_view p_loc, cgen_view p_ls, float alpha, float t, float logt,
                                 int32_t linear_var, cgen_view out, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_mediator_mix: bad dtype");
  CGEN_REQUIRE(z.p && q_loc.p && q_ls.p && p_loc.p && p_ls.p && out.p, "cgen_mediator_mix: null view");
  const int grid = lat_grid((int64_t)n * h * w * c);
  if (dtype == CGEN_F32)
    hipLaunchKernelGGL(mediator_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(z), mk(q_loc), mk(q_ls), mk(p_loc), mk(p_ls), alpha, t, logt, linear_var, mk(out));
  else
    hipLaunchKernelGGL(mediator_kernel<h16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, h, w, c, mk(z), mk(q_loc), mk(q_ls), mk(p_loc), mk(p_ls), alpha, t, logt, linear_var, mk(out));
  return check_launch("cgen_mediator_mix");
}

extern "C" int cgen_gaussian_kl_map(int64_t count, const float* q_loc, const float* q_ls, const float* p_loc, const float* p_ls,
                                    float* out, cgen_stream_t stream) {
  CGEN_REQUIRE(count >= 0 && q_loc && q_ls && p_loc && p_ls && out, "cgen_gaussian_kl_map: bad args");
  if (count == 0) return CGEN_OK;
  hipLaunchKernelGGL(cgen::gaussian_kl_map_kernel, dim3(cgen::lat_grid(count)), dim3(256), 0, (hipStream_t)stream, count, q_loc, q_ls, p_loc, p_ls, out);
  return cgen::check_launch("cgen_gaussian_kl_map");
}

extern "C" int cgen_kl_channel_sums(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                                    cgen_view p_loc, cgen_view p_ls, float logt, float* out, int32_t out_stride, cgen_stream_t stream) {
  CGEN_REQUIRE(dtype == CGEN_F32 || dtype == CGEN_F16, "cgen_kl_channel_sums: bad dtype");
  CGEN_REQUIRE(q_loc.p && q_ls.p && p_loc.p && p_ls.p && out && out_stride >= c, "cgen_kl_channel_sums: bad args");
  CGEN_REQUIRE(c >= 1 && c <= 256 && 256 % c == 0, "cgen_kl_channel_sums: the latent width must divide 256 (got %d)", c);
  using namespace cgen;
  if (dtype == CGEN_F32) hipLaunchKernelGGL(kl_channel_sums_kernel<float>, dim3(n), dim3(256), 0, (hipStream_t)stream, h, w, c, mk(q_loc), mk(q_ls), mk(p_loc), mk(p_ls), logt, out, out_stride);
  else hipLaunchKernelGGL(kl_channel_sums_kernel<h16_t>, dim3(n), dim3(256), 0, (hipStream_t)stream, h, w, c, mk(q_loc), mk(q_ls), mk(p_loc), mk(p_ls), logt, out, out_stride);
  return check_launch("cgen_kl_channel_sums");
}
