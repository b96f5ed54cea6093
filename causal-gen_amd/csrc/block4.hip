// Fused DEFAULT Block for gfx950 (MI355X): the four convolutions of vae.py:57-71 (the non-"light" version: morphomnist, cmnist,
// mimic224) per launch -- and the same kernel as the Block's data gradient.
//   forward (vae.py:57-71, 73-84):   t0 = b0 + W0 . gelu(cat(segs))        1x1, C -> b
//                                    t1 = b1 + W1 * gelu(t0)               3x3, b -> b, zero padding
//                                    t2 = b2 + W2 * gelu(t1)               3x3, b -> b
//                                    out = b3 + W3 . gelu(t2) (+ res)      1x1, b -> Co
//   data gradient (aten::convolution_backward x4 + gelu_backward x4, input part; aux = the forward tensors):
//                                    g2 = (W3^T . g_out) gelu'(t2),  g1 = (W2^T * g2) gelu'(t1),  g0 = (W1^T * g1) gelu'(t0),
//                                    g_x[k] = (W0[k]^T . g0) gelu'(x_k) (+ accumulated gradient)      for every differentiable segment k
// Both directions are the same pipeline  1x1 -> 3x3 -> 3x3 -> 1x1  with a per-phase post-operation, so they share the body.
//
// Shape of the kernel.  These Blocks are HBM- and launch-bound, not MFMA-bound (the bottleneck is C / 4: 3.3 kFLOP per pixel at 224x224
// against 9.2 k of the light Block): four launches move 4.5 C bytes per pixel (every intermediate tensor is written and read back with
// its halo), one launch moves 2 C + 0.75 C (the three intermediates are still written once, centre pixels only: the weight gradients
// and the backward pass need them).  So the design is the simple one:
//   * one workgroup (4 waves) per 8 x 16 tile of the OUTPUT; phase 0 runs on the 12 x 20 halo tile, phase 1 on 10 x 18, phases 2 / 3 on
//     8 x 16; between phases the activated tile lives in LDS only (three tiles, pixel stride an odd number of 16-byte groups: the
//     32x32x16 B-operand reads are conflict-free, MI355X_MICROARCH.md);
//   * phase 0 takes its B operand STRAIGHT from global memory in MFMA lane order (a lane owns 32 contiguous bytes of one pixel per
//     32-channel chunk: the K axis of the weight image is permuted to match), GELU applied in registers: no staging, no barrier;
//   * weights are fragment-ordered images (cgen_weight_prep modes 8-13; one contiguous KiB per wave load, L2-resident);
//   * no persistent loop, no hand-counted waits: up to 4 workgroups per CU (26-79 KB of LDS, <= 128 VGPRs at b <= 16) hide each
//     other's latencies, the compiler schedules the loads.
#include <stddef.h>
#include <stdlib.h>

#include "common.h"

namespace cgen {

typedef float b4_f32x16 __attribute__((ext_vector_type(16)));

__device__ uint4 g_b4zero[2];  // 32 bytes of zeros: source of out-of-image / padding loads

#define B4_TH 8
#define B4_TW 16
#define B4_P0 20   // row pitch (pixels) of the phase-0 tile: 12 x 20
#define B4_P1 18   // phase-1 tile: 10 x 18
#define B4_N0 240
#define B4_N1 180
#define B4_N2 128
#define B4_BIAS_BYTES (3 * 64 * 4 + 256 * 4)

struct B4Div { uint32_t mul, shift; };
static inline B4Div b4_mkdiv(uint32_t d) {
  B4Div f;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t sh = 0;
  while ((1u << sh) < d) ++sh;
  f.shift = sh;
  f.mul = (uint32_t)((((uint64_t)1 << (32 + sh)) + d - 1) / d - ((uint64_t)1 << 32));
  return f;
}
__device__ __forceinline__ int b4_div(int n, const B4Div& f) { return (int)(((uint64_t)__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shift); }

struct B4V { const char* p; int sn, sh, sw; };  // 32-bit BYTE strides (the host checks every view spans < 2^31 bytes)
struct B4Out {
  const char* w;      // phase-3 fragment image [32-row block][16-channel group of the bottleneck][lane][8]
  const float* bias;  // [Co] or null
  B4V out, aux, res;  // aux: gelu' source (data gradient); res: added (forward: the residual; data gradient: the accumulated gradient)
  int Co, nmb;
  int out_rem, res_rem;  // remainder planes of a residual trunk (byte offsets from out.p / res.p, 0 = none): value = hi + rem
};
struct B4P {
  int N, H, W, nseg;
  int nch0, bc8, nout, xcd_order;  // phase-0 chunks of 32 input channels; bottleneck width rounded up to 8 (stored channel groups)
  int tiles_x, tiles_y, ntiles, pad1;
  int inv[3][2];  // ceil(65536 / columns) of phases 0, 1, 2 (tile width + 4, + 2, + 0) for a full-width tile [0] and the last tile of a row [1]
  B4Div d_tx, d_ty;
  B4V seg[3];
  int seg_nch[3], seg_c8[3];
  const char* w[3];       // fragment images of phases 0, 1, 2
  const float* bias[3];   // forward: [b] each, or null
  int nbias[3], pad2;
  B4V mid[3], aux[3];     // mid[k]: written by phase k (centre pixels); aux[k]: gelu' source of phase k (data gradient)
  B4Out o[3];
};

// ---- GELU by table.  These Blocks are VALU-bound on the activation (measured: 1900 VALU instructions per wave and tile against 28
// MFMAs with the erf polynomial of common.h, ~20 instructions per element): a workgroup builds a 385-entry table of second-order Taylor
// coefficients on the grid x0 = i / 32, |x0| <= 6, in LDS at its start (erff / expf, 1.5 entries per thread) and an evaluation is
// clamp, scale, round, subtract, convert, one 16-byte LDS read and two FMAs.  Truncation error: |f'''| d^3 / 6 with |d| <= 1 / 64:
// 2.5e-7 for the normal CDF, 8e-7 for gelu' -- the size of the erf approximation it replaces (1.5e-7), far below binary16 resolution.
// forward table: Phi(x);  data-gradient table: gelu'(x) = Phi(x) + x phi(x)  (gelu'' = phi (2 - x^2), gelu''' = phi (x^3 - 4 x))
#define B4_LUT_N 385
#define B4_LUT_BYTES (B4_LUT_N * 16)
template <bool FWD, int NT>
__device__ __forceinline__ void b4_lut_fill(float4* lut, const int tid) {
  for (int i = tid; i < B4_LUT_N; i += NT) {
    const float x0 = (float)(i - 192) * (1.f / 32.f);
    const float cdf = 0.5f * (1.f + erff(x0 * CGEN_SQRT1_2)), pdf = CGEN_INV_SQRT_2PI * __expf(-0.5f * x0 * x0);
    float4 c;
    if (FWD) { c.x = cdf; c.y = pdf * (1.f / 32.f); c.z = -x0 * pdf * (0.5f / 1024.f); }
    else { c.x = cdf + x0 * pdf; c.y = pdf * (2.f - x0 * x0) * (1.f / 32.f); c.z = pdf * (x0 * x0 * x0 - 4.f * x0) * (0.5f / 1024.f); }
    c.w = 0.f;
    lut[i] = c;
  }
}
// table value at x (`lut0` points at the entry of x0 = 0); beyond |x| = 6 the function is constant to 1e-9
__device__ __forceinline__ float b4_lut(const float x, const float4* __restrict__ lut0) {
  const float t = __builtin_amdgcn_fmed3f(x, -6.f, 6.f) * 32.f;
  const float r = __builtin_rintf(t);
  const float d = t - r;
  const float4 c = lut0[(int)r];
  return fmaf(d, fmaf(d, c.z, c.y), c.x);
}
// gelu of eight / four packed binary16 values
__device__ __forceinline__ uint32_t b4_gelu2(const uint32_t w, const float4* __restrict__ lut0) {
  const float a = h_lo(w), b = h_hi(w);
  return f2h_pk(a * b4_lut(a, lut0), b * b4_lut(b, lut0));
}
__device__ __forceinline__ uint4 b4_gelu8(const uint4 x, const float4* __restrict__ lut0) {
  return make_uint4(b4_gelu2(x.x, lut0), b4_gelu2(x.y, lut0), b4_gelu2(x.z, lut0), b4_gelu2(x.w, lut0));
}
__device__ __forceinline__ uint2 b4_gelu4(const uint2 x, const float4* __restrict__ lut0) { return make_uint2(b4_gelu2(x.x, lut0), b4_gelu2(x.y, lut0)); }
// gelu' of four / eight packed pre-activations (the data-gradient table)
__device__ __forceinline__ void b4_gelu4_bwd(const uint2 x, float* d, const float4* __restrict__ lut0) {
  d[0] = b4_lut(h_lo(x.x), lut0); d[1] = b4_lut(h_hi(x.x), lut0); d[2] = b4_lut(h_lo(x.y), lut0); d[3] = b4_lut(h_hi(x.y), lut0);
}
__device__ __forceinline__ void b4_gelu8_bwd(const uint4 x, float* d, const float4* __restrict__ lut0) {
  b4_gelu4_bwd(make_uint2(x.x, x.y), d, lut0);
  b4_gelu4_bwd(make_uint2(x.z, x.w), d + 4, lut0);
}
__device__ __forceinline__ void b4_unpack8(const uint4 x, float* v) {
  v[0] = h_lo(x.x); v[1] = h_hi(x.x); v[2] = h_lo(x.y); v[3] = h_hi(x.y);
  v[4] = h_lo(x.z); v[5] = h_hi(x.z); v[6] = h_lo(x.w); v[7] = h_hi(x.w);
}
__device__ __forceinline__ uint4 b4_pack8(const float* v) {
  uint4 o;
  o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]); o.z = f2h_pk(v[4], v[5]); o.w = f2h_pk(v[6], v[7]);
  return o;
}
__device__ __forceinline__ b4_f32x16 b4_mfma(const h16x8 a, const h16x8 b, const b4_f32x16 c) {
#ifdef CGEN_H16_BF16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ h16x8 b4_as_h(const uint4 v) {
  union { uint4 u; h16x8 h; } c;
  c.u = v;
  return c.h;
}

// Post-operation of phases 0-2 for one pixel group and one 32-row block.  The weight rows are permuted (cgen_weight_prep modes 8-11) so
// that the lane (pixel, kg) holds HW = (real channels of the block) / 2 CONSECUTIVE channels in its first HW accumulators: channels
// chb .. chb + HW - 1, chb = 32 block + HW kg -- all 64 lanes work on real channels whatever the bottleneck width (4 channels per lane
// at b = 8), in 8-byte units of four.
//   forward:        v = acc + bias;  mid <- rn16(v) (centre pixels);  LDS <- gelu(rn16(v)), zero outside the image
//   data gradient:  v = acc * gelu'(aux);  mid <- rn16(v) (centre);  LDS <- rn16(v), zero outside the image
template <bool FWD, int HW>
__device__ __forceinline__ void b4_post(const b4_f32x16& acc, const int chb, const float* __restrict__ biasL, const float4* __restrict__ lut0,
                                        const B4V& mid, const uint2* apre, const int n, const int iy, const int ix, const bool inimg, const bool centre,
                                        const bool write, char* __restrict__ dst) {
  constexpr int NU = HW / 4;
  uint2 h[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float v[4] = {acc[4 * u], acc[4 * u + 1], acc[4 * u + 2], acc[4 * u + 3]};
    if constexpr (FWD) {
      const float4 bq = *(const float4*)(biasL + chb + 4 * u);
      v[0] += bq.x; v[1] += bq.y; v[2] += bq.z; v[3] += bq.w;
    } else {
      const uint2 t = apre[u];  // (requested before the group's K loop: b4_aux_load)
      float d[4];
      b4_gelu4_bwd(t, d, lut0);
      v[0] *= d[0]; v[1] *= d[1]; v[2] *= d[2]; v[3] *= d[3];
    }
    h[u] = make_uint2(f2h_pk(v[0], v[1]), f2h_pk(v[2], v[3]));
  }
  if (centre) {
    char* mp = (char*)mid.p + (n * mid.sn + iy * mid.sh + ix * mid.sw) + chb * 2;
    if constexpr (HW % 8 == 0) {
#pragma unroll
      for (int u = 0; u < NU; u += 2) *(uint4*)(mp + 8 * u) = make_uint4(h[u].x, h[u].y, h[u + 1].x, h[u + 1].y);
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) *(uint2*)(mp + 8 * u) = h[u];
    }
  }
  if (write) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if constexpr (FWD) h[u] = b4_gelu4(h[u], lut0);
      if (!inimg) h[u] = make_uint2(0, 0);
    }
    if constexpr (HW % 8 == 0) {
#pragma unroll
      for (int u = 0; u < NU; u += 2) *(uint4*)(dst + 8 * u) = make_uint4(h[u].x, h[u].y, h[u + 1].x, h[u + 1].y);
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) *(uint2*)(dst + 8 * u) = h[u];
    }
  }
}
// (the post-operation of 32-row block MB of a bottleneck of NB8 eight-channel groups)
template <bool FWD, int NB8, int MB>
__device__ __forceinline__ void b4_post_mb(const b4_f32x16& acc, const int kg, const float* __restrict__ biasL, const float4* __restrict__ lut0,
                                           const B4V& mid, const uint2* apre, const int n, const int iy, const int ix, const bool inimg, const bool centre,
                                           const bool write, char* __restrict__ pix) {
  constexpr int NREAL = (8 * NB8 - 32 * MB) < 32 ? (8 * NB8 - 32 * MB) : 32, HW = NREAL / 2;
  const int chb = 32 * MB + HW * kg;
  b4_post<FWD, HW>(acc, chb, biasL, lut0, mid, apre, n, iy, ix, inimg, centre, write, pix + chb * 2);
}
// (data gradient) the gelu' sources of a pixel group, requested BEFORE its K loop: up to four 8-byte units per 32-row block and lane
template <bool FWD, int NB8, int MB>
__device__ __forceinline__ void b4_aux_load(uint2* apre, const B4V& aux, const int n, const int iy, const int ix, const bool inimg, const int kg) {
  constexpr int NREAL = (8 * NB8 - 32 * MB) < 32 ? (8 * NB8 - 32 * MB) : 32, HW = NREAL / 2;
  if constexpr (!FWD) {
    const char* ap = aux.p + (n * aux.sn + iy * aux.sh + ix * aux.sw) + (32 * MB + HW * kg) * 2;
#pragma unroll
    for (int u = 0; u < HW / 4; ++u) apre[u] = *(const uint2*)(inimg ? ap + 8 * u : (const char*)g_b4zero);
  }
}
// a pixel group wholly outside the image: its activated values are zeros (the lane pair of a pixel clears its NB16 32-byte groups)
template <int NB16>
__device__ __forceinline__ void b4_clear(char* __restrict__ pix, const int kg, const bool live) {
  if (live) {
#pragma unroll
    for (int q = 0; q < NB16; ++q) *(uint4*)(pix + 32 * q + 16 * kg) = make_uint4(0, 0, 0, 0);
  }
}

// A 3x3 phase over the ACTUAL extent of the tile (edge tiles and small images enumerate only the pixels they have): input tile Uin
// (row pitch PIN pixels) -> rows x cols pixels (cols * inv >> 16 divides) of the tile Uout (row pitch POUT) whose pixel (0, 0) is image
// pixel (y0 - OFF, x0 - OFF); PH: phase index (1 or 2).  Pixel groups of 32 are dealt to the four waves.
template <bool FWD, int NB8, int NW, int PIN, int POUT, int OFF, int PH>
__device__ __forceinline__ void b4_conv3(const B4P& p, const char* __restrict__ Uin, char* __restrict__ Uout, const float* __restrict__ biasL,
                                         const float4* __restrict__ lut0, const int n, const int y0, const int x0, const int th, const int tw,
                                         const int inv, const int wave, const int lane) {
  constexpr int NB16 = (NB8 + 1) / 2, NMB = (NB16 + 1) / 2, PS = (2 * NB16 + 1) * 16;
  const int px = lane & 31, kg = lane >> 5;
  const int cols = tw + 2 * OFF, nout = (th + 2 * OFF) * cols;
  const char* const wl = p.w[PH] + lane * 16;
  for (int g = wave; 32 * g < nout; g += NW) {
    const int m = 32 * g + px, mc = min(m, nout - 1);
    const int my = (mc * inv) >> 16, mx = mc - my * cols;
    const int iy = y0 - OFF + my, ix = x0 - OFF + mx;
    const bool live = m < nout;
    const bool inimg = live && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    char* const pix = Uout + (my * POUT + mx) * PS;
    if (__builtin_amdgcn_ballot_w64(inimg) == 0) { b4_clear<NB16>(pix, kg, live); continue; }
    const char* const bsrc = Uin + (my * PIN + mx) * PS + kg * 16;
    uint2 apre[NMB][4];
    b4_aux_load<FWD, NB8, 0>(apre[0], p.aux[PH], n, iy, ix, inimg, kg);
    if constexpr (NMB > 1) b4_aux_load<FWD, NB8, 1>(apre[1], p.aux[PH], n, iy, ix, inimg, kg);
    b4_f32x16 acc[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int g16 = 0; g16 < NB16; ++g16) {
        const h16x8 B = *(const h16x8*)(bsrc + ((tap / 3) * PIN + tap % 3) * PS + g16 * 32);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) acc[mb] = b4_mfma(*(const h16x8*)(wl + ((mb * 9 + tap) * NB16 + g16) * 1024), B, acc[mb]);
      }
    }
    const bool centre = inimg && my >= OFF && my < OFF + th && mx >= OFF && mx < OFF + tw;
    b4_post_mb<FWD, NB8, 0>(acc[0], kg, biasL + 64 * PH, lut0, p.mid[PH], apre[0], n, iy, ix, inimg, centre, live, pix);
    if constexpr (NMB > 1) b4_post_mb<FWD, NB8, 1>(acc[1], kg, biasL + 64 * PH, lut0, p.mid[PH], apre[1], n, iy, ix, inimg, centre, live, pix);
  }
}

// NW: waves per workgroup -- 4, or 8 where the launch has few tiles (<= 1024: one round of resident workgroups, the time of the launch is
// the dependent chain of ONE tile, and twice the waves halve the pixel groups each has to walk)
// (the kernel body: workgroup `tile` of problem p)
template <bool FWD, int NB8, int NW>
__device__ __forceinline__ void blk4_body(const B4P& p, const int tile) {
  constexpr int NT = 64 * NW;
  constexpr int NB16 = (NB8 + 1) / 2, NMB = (NB16 + 1) / 2, PS = (2 * NB16 + 1) * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const U0 = smem;
  char* const U1 = U0 + B4_N0 * PS;
  char* const U2 = U1 + B4_N1 * PS;
  float* const biasL = (float*)(U2 + B4_N2 * PS);  // [3][64] + [256] (forward)
  float4* const lut = (float4*)((char*)biasL + B4_BIAS_BYTES);
  const float4* const lut0 = lut + 192;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kg = lane >> 5;
  const int b1 = b4_div(tile, p.d_tx), tx = tile - b1 * p.tiles_x;
  const int n = b4_div(b1, p.d_ty);
  const int y0 = (b1 - n * p.tiles_y) * B4_TH, x0 = tx * B4_TW;
  const int th = min(B4_TH, p.H - y0), tw = min(B4_TW, p.W - x0);  // the tile's actual extent
  const int lastx = tx == p.tiles_x - 1 ? 1 : 0;
  const char* const zero = (const char*)g_b4zero;

  b4_lut_fill<FWD, NT>(lut, tid);
  if constexpr (FWD) {
    if (tid < 192) {
      const int ph = tid >> 6, c = tid & 63;
      biasL[tid] = (p.bias[ph] != nullptr && c < p.nbias[ph]) ? p.bias[ph][c] : 0.f;
    }
    if (tid < 256) biasL[192 + tid] = (p.o[0].bias != nullptr && tid < p.o[0].Co) ? p.o[0].bias[tid] : 0.f;
  }
  if constexpr (NB8 & 1) {  // the upper half of the last 16-channel group is never written: K padding of phases 1-3 (zero weights, but 0 x NaN = NaN)
    for (int q = tid; q < B4_N0 + B4_N1 + B4_N2; q += NT) *(uint4*)(U0 + q * PS + 16 * NB8) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  // ------------------------------------------------------------------ phase 0: 1x1 over the halo tile ((th + 4) x (tw + 4) pixels), B operand from global
  {
    const int cols = tw + 4, n0 = (th + 4) * cols, inv = p.inv[0][lastx];
    const int nch = p.nch0, k1 = p.seg_nch[0];
    const int wmb = nch * 2048;                                        // bytes per 32-row block of the image
    for (int g = wave; 32 * g < n0; g += NW) {
      const int m = 32 * g + px, mc = min(m, n0 - 1);
      const int hy = (mc * inv) >> 16, hx = mc - hy * cols;
      const int iy = y0 - 2 + hy, ix = x0 - 2 + hx;
      const bool live = m < n0;
      const bool ok = live && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      char* const pix = U0 + (hy * B4_P0 + hx) * PS;
      if (__builtin_amdgcn_ballot_w64(ok) == 0) { b4_clear<NB16>(pix, kg, live); continue; }
      uint2 apre[NMB][4];
      b4_aux_load<FWD, NB8, 0>(apre[0], p.aux[0], n, iy, ix, ok, kg);
      if constexpr (NMB > 1) b4_aux_load<FWD, NB8, 1>(apre[1], p.aux[0], n, iy, ix, ok, kg);
      b4_f32x16 acc[NMB];
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
      // this lane's pixel in every segment (absent segments are never dereferenced)
      const char* const sp0 = p.seg[0].p + (n * p.seg[0].sn + iy * p.seg[0].sh + ix * p.seg[0].sw) + kg * 16;
      const char* const sp1 = p.seg[1].p + (n * p.seg[1].sn + iy * p.seg[1].sh + ix * p.seg[1].sw) + kg * 16;
      const char* const sp2 = p.seg[2].p + (n * p.seg[2].sn + iy * p.seg[2].sh + ix * p.seg[2].sw) + kg * 16;
      const char* const wl = p.w[0] + lane * 16;
      const int c80 = p.seg_c8[0], c81 = p.seg_c8[1], c82 = p.seg_c8[2];  // (locals: a lambda that captured the kernarg struct would force a scratch copy of it)
      // One chunk = 32 input channels.  K order: step s, lane half kg, element e <-> channel 32 jl + 16 s + 8 kg + e (the two lanes of a
      // pixel read 32 contiguous bytes per step); a step that lies wholly beyond the segment's channels is skipped (wave-uniform).
      // The chunks run through a two-stage software pipeline: the loads (activations + weight fragments) of chunk j + 1 are in flight
      // under the GELU + MFMAs of chunk j -- with one chunk at a time a 12-chunk posterior Block was 12 dependent memory round trips.
      struct Stage { uint4 b0, b1; h16x8 A[2][NMB]; bool two; };
      // (the segment walk is incremental state -- written as selects over (sp0, sp1, sp2) by the chunk's segment number, hipcc builds a
      //  table of the three pointers in SCRATCH and indexes it)
      const char* cur = sp0;
      int c8 = c80, left = k1, jl = 0, sgi = 0;
      const int nc1 = p.seg_nch[1];
      auto issue = [&](const int j, Stage& st) {
        if (jl == left) {
          jl = 0;
          if (sgi == 0) { cur = sp1; c8 = c81; left = nc1; } else { cur = sp2; c8 = c82; left = 1 << 20; }
          ++sgi;
        }
        const char* const src = cur + 64 * jl;
        st.two = 32 * jl + 16 < c8;
        st.b0 = *(const uint4*)((ok && 32 * jl + 8 * kg < c8) ? src : zero);
        st.b1 = *(const uint4*)((ok && 32 * jl + 16 + 8 * kg < c8) ? src + 32 : zero);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) st.A[s][mb] = *(const h16x8*)(wl + j * 2048 + mb * wmb + s * 1024);
        ++jl;
      };
      auto consume = [&](Stage& st) {
        if constexpr (FWD) st.b0 = b4_gelu8(st.b0, lut0);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) acc[mb] = b4_mfma(st.A[0][mb], b4_as_h(st.b0), acc[mb]);
        if (st.two) {
          if constexpr (FWD) st.b1 = b4_gelu8(st.b1, lut0);
#pragma unroll
          for (int mb = 0; mb < NMB; ++mb) acc[mb] = b4_mfma(st.A[1][mb], b4_as_h(st.b1), acc[mb]);
        }
      };
      Stage sa, sb;
      issue(0, sa);
      for (int j = 0; j < nch; j += 2) {
        if (j + 1 < nch) issue(j + 1, sb);
        consume(sa);
        if (j + 2 < nch) issue(j + 2, sa);
        if (j + 1 < nch) consume(sb);
      }
      const bool centre = ok && hy >= 2 && hy < 2 + th && hx >= 2 && hx < 2 + tw;
      b4_post_mb<FWD, NB8, 0>(acc[0], kg, biasL, lut0, p.mid[0], apre[0], n, iy, ix, ok, centre, live, pix);
      if constexpr (NMB > 1) b4_post_mb<FWD, NB8, 1>(acc[1], kg, biasL, lut0, p.mid[0], apre[1], n, iy, ix, ok, centre, live, pix);
    }
  }
  __syncthreads();
  // ------------------------------------------------------------------ phases 1, 2: 3x3 over LDS tiles
  b4_conv3<FWD, NB8, NW, B4_P0, B4_P1, 1, 1>(p, U0, U1, biasL, lut0, n, y0, x0, th, tw, p.inv[1][lastx], wave, lane);
  __syncthreads();
  b4_conv3<FWD, NB8, NW, B4_P1, B4_TW, 0, 2>(p, U1, U2, biasL, lut0, n, y0, x0, th, tw, p.inv[2][lastx], wave, lane);
  __syncthreads();
  // ------------------------------------------------------------------ phase 3: 1x1 to every output, epilogue from the accumulators
  const int n3 = th * tw, inv3 = p.inv[2][lastx];
  for (int g = wave; 32 * g < n3; g += NW) {
    const int m = 32 * g + px, mc = min(m, n3 - 1);
    const int oy = (mc * inv3) >> 16, ox = mc - oy * tw;
    const int iy = y0 + oy, ix = x0 + ox;
    const bool ok = m < n3;
    h16x8 B[NB16];
#pragma unroll
    for (int g16 = 0; g16 < NB16; ++g16) B[g16] = *(const h16x8*)(U2 + (oy * B4_TW + ox) * PS + g16 * 32 + kg * 16);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      if (o < p.nout) {
        const B4Out& O = p.o[o];
        const int pixo = n * O.out.sn + iy * O.out.sh + ix * O.out.sw;
        const int pixa = n * O.aux.sn + iy * O.aux.sh + ix * O.aux.sw;
        const int pixr = n * O.res.sn + iy * O.res.sh + ix * O.res.sw;
        const char* wl = O.w + lane * 16;
        for (int mb = 0; mb < O.nmb; ++mb) {
          const int chb = 32 * mb + 16 * kg;
          uint4 ra[2], rr[2], rr2[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const bool okq = ok && chb + 8 * q < O.Co;
            ra[q] = rr[q] = rr2[q] = make_uint4(0, 0, 0, 0);
            if (!FWD) ra[q] = *(const uint4*)(okq ? O.aux.p + pixa + chb * 2 + 16 * q : zero);
            if (O.res.p != nullptr) rr[q] = *(const uint4*)(okq ? O.res.p + pixr + chb * 2 + 16 * q : zero);
            if (FWD && O.res_rem) rr2[q] = *(const uint4*)(okq ? O.res.p + O.res_rem + pixr + chb * 2 + 16 * q : zero);
          }
          b4_f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
          for (int g16 = 0; g16 < NB16; ++g16) acc = b4_mfma(*(const h16x8*)(wl + (mb * NB16 + g16) * 1024), B[g16], acc);
          float v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = acc[j];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (chb + 8 * q < O.Co) {  // (the kg half of a partial last block: no work on padding channels)
              float r[8];
              if constexpr (FWD) {
                const float4 b0 = *(const float4*)(biasL + 192 + chb + 8 * q), b1 = *(const float4*)(biasL + 192 + chb + 8 * q + 4);
                v[8 * q] += b0.x; v[8 * q + 1] += b0.y; v[8 * q + 2] += b0.z; v[8 * q + 3] += b0.w;
                v[8 * q + 4] += b1.x; v[8 * q + 5] += b1.y; v[8 * q + 6] += b1.z; v[8 * q + 7] += b1.w;
              } else {
                b4_gelu8_bwd(ra[q], r, lut0);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * q + e] *= r[e];
              }
              if (O.res.p != nullptr) {
                b4_unpack8(rr[q], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * q + e] += r[e];
                if (FWD && O.res_rem) {
                  b4_unpack8(rr2[q], r);
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[8 * q + e] += r[e];
                }
              }
              if (ok) {
                const uint4 h = b4_pack8(v + 8 * q);
                *(uint4*)((char*)O.out.p + pixo + chb * 2 + 16 * q) = h;
                if (FWD && O.out_rem) {
                  float hv[8];
                  b4_unpack8(h, hv);
#pragma unroll
                  for (int e = 0; e < 8; ++e) hv[e] = v[8 * q + e] - hv[e];
                  *(uint4*)((char*)O.out.p + O.out_rem + pixo + chb * 2 + 16 * q) = b4_pack8(hv);
                }
              }
            }
          }
        }
      }
    }
  }
}

template <bool FWD, int NB8, int NW>
__global__ __launch_bounds__(64 * NW, NB8 <= 2 ? 4 : (NB8 <= 4 ? (NW == 8 ? 4 : 3) : 2)) void blk4_kernel(const B4P p) {
  // XCD-contiguous tile order: the dispatcher deals consecutive workgroups round-robin to the eight XCDs (each with its own L2), so
  // with tile = workgroup index no L2 ever sees two neighbouring tiles and every halo column comes over the fabric again (measured:
  // fetch 2.7 x the input at 224x224).  Workgroup i takes tile (i mod 8) * (ntiles / 8) + i / 8: an XCD walks a contiguous range.
  const int bid = blockIdx.x, per = p.ntiles >> 3;
  blk4_body<FWD, NB8, NW>(p, (p.xcd_order && bid < 8 * per) ? (bid & 7) * per + (bid >> 3) : bid);
}
// Two independent DATA-GRADIENT problems of one instance in one launch: workgroups 0 .. na - 1 take the first, the rest the second
// (the backward of a decoder layer's posterior and prior Blocks, vae.py:240-301: at <= 28x28 a launch is <= 256 tiles and its time is
// ONE tile's dependent chain -- two launches back to back are two chains, one launch is one)
template <int NB8>
__global__ __launch_bounds__(256, NB8 <= 2 ? 4 : (NB8 <= 4 ? 3 : 2)) void blk4_pair_kernel(const B4P pa, const B4P pb, const int na) {
  if ((int)blockIdx.x < na) blk4_body<false, NB8, 4>(pa, blockIdx.x);
  else blk4_body<false, NB8, 4>(pb, blockIdx.x - na);
}

// ----------------------------------------------------------------------------- host side
static bool b4_view(const cgen_view& v, int n, int h, int w, B4V& o) {
  o.p = (const char*)v.p; o.sn = o.sh = o.sw = 0;
  if (!v.p) return true;
  const int64_t ext = ((int64_t)n * v.sn + (int64_t)(h + B4_TH + 4) * v.sh + (int64_t)(w + B4_TW + 4) * v.sw + v.c + 64) * 2;
  if (ext >= ((int64_t)1 << 31) || v.sn < 0 || v.sh < 0 || v.sw < 0) return false;
  if (((uintptr_t)v.p % 16) || (v.sn * 2) % 16 || (v.sh * 2) % 16 || (v.sw * 2) % 16) return false;
  o.sn = (int)(v.sn * 2); o.sh = (int)(v.sh * 2); o.sw = (int)(v.sw * 2);
  return true;
}
// every 16-byte channel group up to ceil8(c) can be read whole
static bool b4_groups_ok(const cgen_view& v) { return v.c % 8 == 0 || v.cpad >= ((v.c + 7) & ~7); }

static int b4_fill(const cgen_block4_args* a, B4P& p, int& nb8) {
  if (!a || a->dtype != CGEN_F16 || a->nseg < 1 || a->nseg > 3 || a->n <= 0 || a->h < 1 || a->w < 1) return 0;
  if (a->nout < 1 || a->nout > 3 || a->b < 1 || a->b > 64) return 0;
  const bool fwd = a->fwd != 0;
  if (fwd && a->nout != 1) return 0;
  memset(&p, 0, sizeof(p));
  p.N = a->n; p.H = a->h; p.W = a->w; p.nseg = a->nseg; p.nout = a->nout;
  p.bc8 = (a->b + 7) & ~7;
  nb8 = (a->b + 7) / 8;
  if (nb8 == 7) return 0;  // (no instance: 56 channels would be written as 64)
  int nch = 0;
  for (int s = 0; s < a->nseg; ++s) {
    const cgen_view& v = a->seg[s];
    if (!v.p || v.c <= 0 || !b4_groups_ok(v) || !b4_view(v, a->n, a->h, a->w, p.seg[s])) return 0;
    p.seg_nch[s] = (v.c + 31) / 32;
    p.seg_c8[s] = (v.c + 7) & ~7;
    nch += p.seg_nch[s];
  }
  p.nch0 = nch;
  for (int k = 0; k < 3; ++k) {
    if (!a->wimg[k]) return 0;
    p.w[k] = (const char*)a->wimg[k];
    p.bias[k] = fwd ? a->bias[k] : nullptr;
    p.nbias[k] = a->b;
    const cgen_view& m = a->mid[k];
    if (!m.p || m.c != a->b || !b4_groups_ok(m) || !b4_view(m, a->n, a->h, a->w, p.mid[k])) return 0;
    if (fwd) { if (a->mid_aux[k].p) return 0; }
    else {
      const cgen_view& x = a->mid_aux[k];
      if (!x.p || x.c != a->b || !b4_groups_ok(x) || !b4_view(x, a->n, a->h, a->w, p.aux[k])) return 0;
    }
  }
  for (int o = 0; o < a->nout; ++o) {
    const cgen_block3_out& s = a->o[o];
    B4Out& d = p.o[o];
    if (!s.w || !s.out.p || s.out.c <= 0 || s.out.c % 8 != 0 || (fwd && s.out.c > 256)) return 0;
    d.w = (const char*)s.w; d.bias = fwd ? s.bias : nullptr;
    d.Co = s.out.c; d.nmb = (s.out.c + 31) / 32;
    if (!b4_view(s.out, a->n, a->h, a->w, d.out)) return 0;
    if (fwd) { if (s.aux.p) return 0; }
    else if (!s.aux.p || s.aux.c != s.out.c || !b4_view(s.aux, a->n, a->h, a->w, d.aux)) return 0;
    if (s.res1.p && (s.res1.c != s.out.c || !b4_view(s.res1, a->n, a->h, a->w, d.res))) return 0;
    if (s.out_rem < 0 || s.out_rem >= ((int64_t)1 << 30) || s.res1_rem < 0 || s.res1_rem >= ((int64_t)1 << 30)) return 0;
    if (!fwd && (s.out_rem || s.res1_rem)) return 0;
    d.out_rem = (int)s.out_rem; d.res_rem = s.res1.p ? (int)s.res1_rem : 0;
  }
  p.tiles_x = ceil_div(p.W, B4_TW); p.tiles_y = ceil_div(p.H, B4_TH);
  const int64_t nt = (int64_t)p.N * p.tiles_x * p.tiles_y;
  if (nt >= ((int64_t)1 << 30)) return 0;
  p.ntiles = (int)nt;
  static const int xcd_env = [] { const char* e = getenv("CGEN_BLK4_XCD"); return e ? atoi(e) : 1; }();
  p.xcd_order = xcd_env && nt >= 512;  // (launches of at least two tiles per CU: below that placement decides nothing)
  p.d_tx = b4_mkdiv(p.tiles_x); p.d_ty = b4_mkdiv(p.tiles_y);
  {
    const int twl = p.W - B4_TW * (p.tiles_x - 1);  // width of a row's last tile
    for (int k = 0; k < 3; ++k) {
      p.inv[k][0] = 65536 / (B4_TW + 4 - 2 * k) + 1;
      p.inv[k][1] = 65536 / (twl + 4 - 2 * k) + 1;
    }
    if (p.tiles_x == 1) for (int k = 0; k < 3; ++k) p.inv[k][0] = p.inv[k][1];
  }
  return 1;
}

template <bool FWD, int NB8>
static void b4_launch(const B4P& p, hipStream_t st) {
  constexpr int PS = (2 * ((NB8 + 1) / 2) + 1) * 16;
  const size_t lds = (size_t)(B4_N0 + B4_N1 + B4_N2) * PS + B4_BIAS_BYTES + B4_LUT_BYTES;
  static const int nw_env = [] { const char* e = getenv("CGEN_BLK4_NW"); return e ? atoi(e) : 0; }();  // (4 / 8: force; measurements only)
  // (the data gradient runs next to the background weight-gradient kernel: an eight-wave workgroup has to find eight free wave slots on
  //  one CU at once and waits for that kernel's workgroups to retire -- mimic224 22.4 -> 24.7 ms/step with eight waves there)
  const bool eight = nw_env == 84 ? FWD && p.ntiles <= 1024 : (nw_env ? nw_env == 8 : FWD && p.ntiles <= 1024);
  if (eight) {
    (void)hipFuncSetAttribute((const void*)blk4_kernel<FWD, NB8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((blk4_kernel<FWD, NB8, 8>), dim3(p.ntiles), dim3(512), lds, st, p);
  } else {
    (void)hipFuncSetAttribute((const void*)blk4_kernel<FWD, NB8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((blk4_kernel<FWD, NB8, 4>), dim3(p.ntiles), dim3(256), lds, st, p);
  }
}
template <bool FWD>
static void b4_launch_dir(const B4P& p, const int nb8, hipStream_t st) {
  switch (nb8) {
    case 1: b4_launch<FWD, 1>(p, st); break;
    case 2: b4_launch<FWD, 2>(p, st); break;
    case 3: b4_launch<FWD, 3>(p, st); break;
    case 4: b4_launch<FWD, 4>(p, st); break;
    case 5: b4_launch<FWD, 5>(p, st); break;
    case 6: b4_launch<FWD, 6>(p, st); break;
    default: b4_launch<FWD, 8>(p, st); break;  // (56 / 64 channels: the image pads 56 to 64)
  }
}

}  // namespace cgen

using namespace cgen;

extern "C" int cgen_block4_supported(const cgen_block4_args* a) {
  B4P p;
  int nb8;
  return b4_fill(a, p, nb8);
}

template <int NB8>
static void b4_launch_pair(const B4P& pa, const B4P& pb, hipStream_t st) {
  constexpr int PS = (2 * ((NB8 + 1) / 2) + 1) * 16;
  const size_t lds = (size_t)(B4_N0 + B4_N1 + B4_N2) * PS + B4_BIAS_BYTES + B4_LUT_BYTES;
  (void)hipFuncSetAttribute((const void*)blk4_pair_kernel<NB8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((blk4_pair_kernel<NB8>), dim3(pa.ntiles + pb.ntiles), dim3(256), lds, st, pa, pb, pa.ntiles);
}
static int b4_pair_fill(const cgen_block4_args* a, const cgen_block4_args* b, B4P& pa, B4P& pb, int& nb8) {
  int nb8b = 0;
  if (!a || !b || a->fwd || b->fwd) return 0;
  if (!b4_fill(a, pa, nb8) || !b4_fill(b, pb, nb8b) || nb8 != nb8b) return 0;
  return (int64_t)pa.ntiles + pb.ntiles < ((int64_t)1 << 30);
}
extern "C" int cgen_block4_pair_supported(const cgen_block4_args* a, const cgen_block4_args* b) {
  B4P pa, pb;
  int nb8;
  return b4_pair_fill(a, b, pa, pb, nb8);
}
extern "C" int cgen_block4_pair(const cgen_block4_args* a, const cgen_block4_args* b, cgen_stream_t stream) {
  B4P pa, pb;
  int nb8 = 0;
  CGEN_REQUIRE(b4_pair_fill(a, b, pa, pb, nb8), "cgen_block4_pair: two data-gradient problems of one bottleneck class (ask cgen_block4_pair_supported first)");
  switch (nb8) {
    case 1: b4_launch_pair<1>(pa, pb, (hipStream_t)stream); break;
    case 2: b4_launch_pair<2>(pa, pb, (hipStream_t)stream); break;
    case 3: b4_launch_pair<3>(pa, pb, (hipStream_t)stream); break;
    case 4: b4_launch_pair<4>(pa, pb, (hipStream_t)stream); break;
    case 5: b4_launch_pair<5>(pa, pb, (hipStream_t)stream); break;
    case 6: b4_launch_pair<6>(pa, pb, (hipStream_t)stream); break;
    default: b4_launch_pair<8>(pa, pb, (hipStream_t)stream); break;
  }
  return check_launch("cgen_block4_pair");
}

extern "C" int cgen_block4(const cgen_block4_args* a, cgen_stream_t stream) {
  B4P p;
  int nb8 = 0;
  CGEN_REQUIRE(b4_fill(a, p, nb8), "cgen_block4: shape / layout not served by the fused default-Block kernel (ask cgen_block4_supported first)");
  if (a->fwd) b4_launch_dir<true>(p, nb8, (hipStream_t)stream);
  else b4_launch_dir<false>(p, nb8, (hipStream_t)stream);
  static const bool trace = getenv("CGEN_CONV_TRACE") != nullptr;
  if (trace) fprintf(stderr, "blk4[%s] %dx%dx%d chunks %d b %d Co %d nseg %d nout %d | grid %d\n", a->fwd ? "fwd" : "bwd", a->n, a->h, a->w, p.nch0, a->b, p.o[0].Co, a->nseg, a->nout, p.ntiles);
  return check_launch("cgen_block4");
}
