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

struct LatZpFwdP {
  LatP l;
  View pa, hres, pfeat, out;  // pa: [n,h,w,ctx] (ctx8 <= 16); hres / pfeat: optional residuals; out: [n,h,w,co]
  const h16_t* w;            // z_proj forward image: rows ceil16(co) x krow, column = concat8 index (z 0..15, parents 16..)
  const float* bias;
  int co, krow, rows, pa_groups;  // pa_groups: 8-channel groups of the parents (0, 1 or 2)
};

__global__ __launch_bounds__(512) void reparam_zproj_fwd_kernel(LatZpFwdP q) {  // 8 waves x 16 pixels = the 128 pixels of a KL chunk, one pass
  __shared__ float sm[8];
  const LatP& p = q.l;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  const int npix = p.h * p.w, per = npix * 16;
  uint64_t seed = 0, off = 0;
  if (!p.eps_in.p) { seed = p.rng[0]; off = p.rng[1]; }
  float kl_acc = 0.f;
  const int co16 = (q.co + 15) >> 4;
  {
    const int pix = chunk * (LAT_CHUNK / 16) + wave * 16 + fr;
    const bool valid = pix < npix;
    const int y = valid ? pix / p.w : 0, x = valid ? pix - y * p.w : 0;
    union { uint4 u; h16x8 v; } bfrag;
    bfrag.u = make_uint4(0, 0, 0, 0);
    if (fg < 2) {
      const int ch = fg * 8;
      if (valid) {
        float ql[8], qs[8], pl[8], ps[8], eps[8], z[8];
        unpack8(ld8(p.q_loc, off8(p.q_loc, b, y, x, ch)), ql);
        unpack8(ld8(p.q_ls, off8(p.q_ls, b, y, x, ch)), qs);
        unpack8(ld8(p.p_loc, off8(p.p_loc, b, y, x, ch)), pl);
        unpack8(ld8(p.p_ls, off8(p.p_ls, b, y, x, ch)), ps);
        if (p.eps_in.p) {
          unpack8(ld8(p.eps_in, off8(p.eps_in, b, y, x, ch)), eps);
        } else {
          const uint64_t i0 = (uint64_t)b * per + (uint64_t)pix * 16 + ch;  // as reparam_kl_fwd_vec8_kernel: same draws
          float n4[4];
          Philox::normal4(seed, off, p.stream_id, i0 >> 2, n4);
          eps[0] = n4[0]; eps[1] = n4[1]; eps[2] = n4[2]; eps[3] = n4[3];
          Philox::normal4(seed, off, p.stream_id, (i0 >> 2) + 1, n4);
          eps[4] = n4[0]; eps[5] = n4[1]; eps[6] = n4[2]; eps[7] = n4[3];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float qq = qs[k] + p.logt, pp = ps[k] + p.logt;
          const float sq = expf(qq), sp = expf(pp), d = ql[k] - pl[k];
          z[k] = ql[k] + sq * eps[k];
          kl_acc += -0.5f + pp - qq + 0.5f * (sq * sq + d * d) / (sp * sp);
        }
        bfrag.u = pack8(z);
        st8(p.z, off8(p.z, b, y, x, ch), bfrag.u);
        if (p.eps_out.p) st8(p.eps_out, off8(p.eps_out, b, y, x, ch), pack8(eps));
      }
    } else if (fg - 2 < q.pa_groups) {
      if (valid) bfrag.u = ld8(q.pa, off8(q.pa, b, y, x, (fg - 2) * 8));
    }
    // ---- h'[co] = W[co][0:32] . [z | pa] + bias + h + p_feat: one MFMA per 16 output channels.  All operands of up to
    // eight channel tiles are requested first (one memory round trip per batch, not per tile), then MFMAs + stores.
    const h16_t* wl = q.w + fg * 8;
    const int o_out = off8(q.out, b, y, x, 0), o_h = off8(q.hres, b, y, x, 0), o_f = off8(q.pfeat, b, y, x, 0);
#pragma unroll 1
    for (int ct0 = 0; ct0 < co16; ct0 += 8) {
      uint4 aw[8];
      float4 bb[8];
      uint2 rh[8], rf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ct = ct0 + j;
        const int row = min(ct * 16 + fr, q.rows - 1);
        const int c0 = ct * 16 + fg * 4;
        const bool live = valid && c0 < q.co;
        aw[j] = *(const uint4*)(wl + (int64_t)row * q.krow);
        bb[j] = *(const float4*)((q.bias && c0 + 4 <= q.co) ? (const void*)(q.bias + c0) : (const void*)lat_zero16);
        rh[j] = *(const uint2*)((live && q.hres.p) ? (const void*)((const h16_t*)q.hres.p + o_h + c0) : (const void*)lat_zero16);
        rf[j] = *(const uint2*)((live && q.pfeat.p) ? (const void*)((const h16_t*)q.pfeat.p + o_f + c0) : (const void*)lat_zero16);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c0 = (ct0 + j) * 16 + fg * 4;
        union { uint4 u; h16x8 v; } a;
        a.u = aw[j];
        lat_f32x4 acc = {bb[j].x, bb[j].y, bb[j].z, bb[j].w};
        acc = mfma_h16(a.v, bfrag.v, acc, 0, 0, 0);
        if (valid && c0 < q.co) {
          const float v0 = acc[0] + h_lo(rh[j].x) + h_lo(rf[j].x);
          const float v1 = acc[1] + h_hi(rh[j].x) + h_hi(rf[j].x);
          const float v2 = acc[2] + h_lo(rh[j].y) + h_lo(rf[j].y);
          const float v3 = acc[3] + h_hi(rh[j].y) + h_hi(rf[j].y);
          uint2 o;
          o.x = f2h_pk(v0, v1); o.y = f2h_pk(v2, v3);
          *(uint2*)((h16_t*)q.out.p + o_out + c0) = o;
        }
      }
    }
  }
  kl_acc = wave_sum(kl_acc);
  if (lane == 0) sm[wave] = kl_acc;
  __syncthreads();
  if (threadIdx.x == 0) p.kl_part[(int64_t)b * p.kl_stride + chunk] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) + ((sm[4] + sm[5]) + (sm[6] + sm[7]));
}

// Backward: g_z = W_z^T g_h' (K = co, the data gradient of z_proj's z segment) + the part already in grad(z) (z_feat_proj's),
// then the reparameterisation / KL gradient on it.  A wave owns 16 pixels; the MFMA result leaves lane (pixel fr, K group fg)
// with z channels fg * 4 .. fg * 4 + 3 of its pixel, and the element-wise part runs on those four.
struct LatZpBwdP {
  LatBwdP l;
  View gh;            // grad of z_proj's output [n,h,w,co]
  const h16_t* wdg;  // z_proj data-gradient image of the z segment: rows 16 (z channel) x krow_dg, column = co
  int co, krow_dg, groups;  // groups: 16-pixel groups in total
};

__global__ __launch_bounds__(256) void reparam_zproj_bwd_kernel(LatZpBwdP q) {
  const LatBwdP& p = q.l;
  if ((int)blockIdx.x >= p.main_blocks) {  // rider blocks (as reparam_kl_bwd_vec8_kernel)
    const int rg = p.ride_c >> 3, per = p.h * p.w * rg;
    const int64_t tot = (int64_t)p.n * per;
    const int nb = gridDim.x - p.main_blocks;
    for (int64_t g = (int64_t)(blockIdx.x - p.main_blocks) * 256 + threadIdx.x; g < tot; g += (int64_t)nb * 256) {
      const int b = (int)(g / per), g8 = (int)(g - (int64_t)b * per);
      const int pix = g8 / rg, ch = (g8 - pix * rg) * 8;
      const int y = pix / p.w, x = pix - y * p.w;
      uint4 v = ld8(p.ride_src, off8(p.ride_src, b, y, x, ch));
      const int o = off8(p.ride_dst, b, y, x, ch);
      if (p.ride_acc) {
        float a[8], t[8];
        unpack8(v, a);
        unpack8(ld8(p.ride_dst, o), t);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += t[k];
        v = pack8(a);
      }
      st8(p.ride_dst, o, v);
    }
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  const int npix = p.h * p.w, gps = (npix + 15) >> 4;  // 16-pixel groups per sample (a group never straddles two samples)
  const int nks = (q.co + 31) >> 5;
  for (int g = blockIdx.x * 4 + wave; g < q.groups; g += p.main_blocks * 4) {
    const int b = g / gps, pix = (g - b * gps) * 16 + fr;
    const bool valid = pix < npix;
    const int y = valid ? pix / p.w : 0, x = valid ? pix - y * p.w : 0;
    lat_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int o_g = off8(q.gh, b, y, x, 0);
    const h16_t* wl = q.wdg + (int64_t)fr * q.krow_dg + fg * 8;
    const int ch = fg * 4;
    // element-wise operands of this lane's four z channels: requested now, consumed after the MFMAs
    auto ld4u = [&](const View& v, bool on) -> uint2 {
      return *(const uint2*)((on && valid && v.p) ? (const void*)((const h16_t*)v.p + off8(v, b, y, x, ch)) : (const void*)lat_zero16);
    };
    const uint2 u_ql = ld4u(p.q_loc, true), u_qs = ld4u(p.q_ls, true), u_pl = ld4u(p.p_loc, true), u_ps = ld4u(p.p_ls, true), u_z = ld4u(p.z, true);
    const uint2 u_gz = ld4u(p.gz, true);
    const uint2 u_a1 = ld4u(p.g_q_loc, p.acc_q != 0), u_a2 = ld4u(p.g_q_ls, p.acc_q != 0), u_a3 = ld4u(p.g_p_loc, p.acc_p != 0), u_a4 = ld4u(p.g_p_ls, p.acc_p != 0);
#pragma unroll 1
    for (int ks0 = 0; ks0 < nks; ks0 += 8) {
      uint4 aw[8], bw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ks = min(ks0 + j, nks - 1);
        const int c = (ks0 + j) * 32 + fg * 8;
        aw[j] = *(const uint4*)(wl + ks * 32);  // (the image is zero beyond co: its row length is padded by 32)
        bw[j] = *(const uint4*)((valid && c < q.co) ? (const void*)((const h16_t*)q.gh.p + o_g + c) : (const void*)lat_zero16);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        union { uint4 u; h16x8 v; } a, bq;
        a.u = aw[j]; bq.u = bw[j];
        acc = mfma_h16(a.v, bq.v, acc, 0, 0, 0);  // (steps past nks multiply by a zero fragment)
      }
    }
    if (!valid) continue;
    auto un4 = [](const uint2 t, float (&o)[4]) {
      o[0] = h_lo(t.x); o[1] = h_hi(t.x);
      o[2] = h_lo(t.y); o[3] = h_hi(t.y);
    };
    auto st4f = [&](const View& v, const float (&o)[4]) {
      uint2 t;
      t.x = f2h_pk(o[0], o[1]); t.y = f2h_pk(o[2], o[3]);
      *(uint2*)((h16_t*)v.p + off8(v, b, y, x, ch)) = t;
    };
    float ql[4], qs[4], pl[4], ps[4], zv[4], gz[4] = {acc[0], acc[1], acc[2], acc[3]};
    un4(u_ql, ql); un4(u_qs, qs); un4(u_pl, pl); un4(u_ps, ps); un4(u_z, zv);
    {
      float t[4];
      un4(u_gz, t);  // (zeros when grad(z) has no other contribution)
#pragma unroll
      for (int k = 0; k < 4; ++k) gz[k] += t[k];
    }
    const float k0 = p.coef[(int64_t)b * p.coef_stride];
    float o1[4], o2[4], o3[4], o4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float kk = k0 * (p.chan_scale ? p.chan_scale[ch + k] : 1.f);
      const float qq = qs[k] + p.logt, pp = ps[k] + p.logt;
      const float e2q = expf(2.f * qq), ie2p = expf(-2.f * pp), d = ql[k] - pl[k];
      o1[k] = kk * d * ie2p + gz[k];
      o2[k] = kk * (e2q * ie2p - 1.f) + gz[k] * (zv[k] - ql[k]);
      o3[k] = -kk * d * ie2p;
      o4[k] = kk * (1.f - (e2q + d * d) * ie2p);
    }
    {
      float t[4];  // (zeros unless accumulating)
      un4(u_a1, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) o1[k] += t[k];
      un4(u_a2, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) o2[k] += t[k];
      un4(u_a3, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) o3[k] += t[k];
      un4(u_a4, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) o4[k] += t[k];
    }
    st4f(p.g_q_loc, o1); st4f(p.g_q_ls, o2); st4f(p.g_p_loc, o3); st4f(p.g_p_ls, o4);
  }
}

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


static inline bool lat_v8(const cgen_view& v) {  // 8-byte vector access on a bf16 view
  return !v.p || (((uintptr_t)v.p % 8 == 0) && v.sn % 4 == 0 && v.sh % 4 == 0 && v.sw % 4 == 0);
}

extern "C" int cgen_latent_zproj_supported(const cgen_latent_zproj_args* a) {
  if (!a || a->dtype != CGEN_F16 || a->c != 16 || a->n <= 0 || a->h <= 0 || a->w <= 0 || a->co <= 0 || a->co % 4) return 0;
  const int pa8 = a->pa.p ? (a->pa.c + 7) / 8 * 8 : 0;
  if (pa8 > 16 || (a->pa.p && !(a->pa.c % 8 == 0 || a->pa.cpad >= pa8))) return 0;
  if (!lat_vec8_ok(a->n, 16, {a->q_loc, a->q_ls, a->p_loc, a->p_ls, a->eps_in, a->z, a->eps_out})) return 0;
  if (a->pa.p && !lat_vec8_ok(a->n, 8, {a->pa})) return 0;
  if (!lat_v8(a->hres) || !lat_v8(a->pfeat) || !lat_v8(a->out) || !lat_v8(a->gout)) return 0;
  if (a->gout.p && !lat_vec8_ok(a->n, 8, {a->gout})) return 0;  // 16-byte fragment loads of grad(out)
  if (!lat_v8(a->gz) || !lat_v8(a->g_q_loc) || !lat_v8(a->g_q_ls) || !lat_v8(a->g_p_loc) || !lat_v8(a->g_p_ls)) return 0;
  for (const cgen_view* v : {&a->out, &a->hres, &a->pfeat, &a->gout})
    if (v->p && ((int64_t)a->n * v->sn + (int64_t)a->h * v->sh) * 2 >= ((int64_t)1 << 31)) return 0;
  return 1;
}

extern "C" int cgen_latent_zproj_fwd(const cgen_latent_zproj_args* a, cgen_stream_t stream) {
  CGEN_REQUIRE(cgen_latent_zproj_supported(a), "cgen_latent_zproj_fwd: shape / layout not served (ask cgen_latent_zproj_supported first)");
  CGEN_REQUIRE(a->q_loc.p && a->q_ls.p && a->p_loc.p && a->p_ls.p && a->z.p && a->kl_part && a->out.p && a->w_fwd, "cgen_latent_zproj_fwd: null view");
  CGEN_REQUIRE(a->eps_in.p || a->rng, "cgen_latent_zproj_fwd: need eps or rng");
  LatZpFwdP q;
  memset(&q, 0, sizeof(q));
  LatP& p = q.l;
  p.n = a->n; p.h = a->h; p.w = a->w; p.c = 16;
  p.q_loc = mk(a->q_loc); p.q_ls = mk(a->q_ls); p.p_loc = mk(a->p_loc); p.p_ls = mk(a->p_ls);
  p.eps_in = mk(a->eps_in); p.z = mk(a->z); p.eps_out = mk(a->eps_out);
  p.rng = a->rng; p.stream_id = a->stream_id; p.logt = a->logt; p.kl_part = a->kl_part; p.kl_stride = a->kl_stride;
  q.pa = mk(a->pa); q.hres = mk(a->hres); q.pfeat = mk(a->pfeat); q.out = mk(a->out);
  q.w = (const h16_t*)a->w_fwd; q.bias = a->bias; q.co = a->co;
  const int pa8 = a->pa.p ? (a->pa.c + 7) / 8 * 8 : 0;
  q.pa_groups = pa8 / 8;
  q.krow = ((16 + pa8 + 31) / 32) * 32 + 32;
  q.rows = (a->co + 15) / 16 * 16;
  dim3 grid(cgen_reparam_kl_chunks(a->h, a->w, 16), a->n);
  hipLaunchKernelGGL(reparam_zproj_fwd_kernel, grid, dim3(512), 0, (hipStream_t)stream, q);
  return check_launch("cgen_latent_zproj_fwd");
}

extern "C" int cgen_latent_zproj_bwd(const cgen_latent_zproj_args* a, cgen_stream_t stream) {
  CGEN_REQUIRE(cgen_latent_zproj_supported(a), "cgen_latent_zproj_bwd: shape / layout not served (ask cgen_latent_zproj_supported first)");
  CGEN_REQUIRE(a->q_loc.p && a->q_ls.p && a->p_loc.p && a->p_ls.p && a->z.p && a->kl_coef_dev && a->g_q_loc.p && a->g_q_ls.p && a->g_p_loc.p &&
                   a->g_p_ls.p && a->gout.p && a->w_dgrad, "cgen_latent_zproj_bwd: null view");
  LatZpBwdP q;
  memset(&q, 0, sizeof(q));
  LatBwdP& p = q.l;
  p.n = a->n; p.h = a->h; p.w = a->w; p.c = 16;
  p.q_loc = mk(a->q_loc); p.q_ls = mk(a->q_ls); p.p_loc = mk(a->p_loc); p.p_ls = mk(a->p_ls); p.z = mk(a->z); p.gz = mk(a->gz);
  p.g_q_loc = mk(a->g_q_loc); p.g_q_ls = mk(a->g_q_ls); p.g_p_loc = mk(a->g_p_loc); p.g_p_ls = mk(a->g_p_ls);
  p.coef = a->kl_coef_dev; p.chan_scale = a->kl_chan_scale; p.coef_stride = a->coef_stride; p.acc_q = a->acc_q; p.acc_p = a->acc_p; p.logt = a->logt;
  q.gh = mk(a->gout); q.wdg = (const h16_t*)a->w_dgrad; q.co = a->co;
  q.krow_dg = (((a->co + 7) / 8 * 8) + 31) / 32 * 32 + 32;
  const int gps = (a->h * a->w + 15) / 16;
  q.groups = a->n * gps;
  p.main_blocks = std::min((q.groups + 3) / 4, 2048);
  int ride_blocks = 0;
  if (a->ride_src.p) {
    CGEN_REQUIRE(a->ride_dst.p && a->ride_src.c == a->ride_dst.c && a->ride_src.c % 8 == 0 && lat_vec8_ok(a->n, a->ride_src.c, {a->ride_src, a->ride_dst}),
                 "cgen_latent_zproj_bwd: the rider needs 8-channel multiples and 16-byte aligned views");
    p.ride_src = mk(a->ride_src); p.ride_dst = mk(a->ride_dst); p.ride_c = a->ride_src.c; p.ride_acc = a->ride_acc;
    ride_blocks = lat_grid((int64_t)a->n * a->h * a->w * a->ride_src.c / 8);
  }
  hipLaunchKernelGGL(reparam_zproj_bwd_kernel, dim3(p.main_blocks + ride_blocks), dim3(256), 0, (hipStream_t)stream, q);
  return check_launch("cgen_latent_zproj_bwd");
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
