// Per-image STAGE INTERPRETER (bf16): one launch walks a whole run of low-resolution layers.
//
// Why.  At <= 12x12 (ukbb192 / mimic) and at every resolution of the 32x32 models a layer has a few hundred MFMAs per image:
// one launch per conv is a pure latency chain -- dispatch + prologue + operand round trips, ~10 us each, ~900 per step
// (DESIGN 3.7).  Samples are independent, and a whole low-resolution image fits one CU's LDS.  So the engine hands this
// kernel a LIST of consecutive ops (convs with their fused prologue / epilogue, pooling, upsampling + bias, reparameterise +
// KL and its gradient, gradient copies) and ONE workgroup per image executes the list front to back with workgroup barriers
// between ops instead of kernel boundaries: every tensor an op reads was written by the same workgroup, so
// __syncthreads() is all the ordering there is (no grid barrier, no cross-CU hand-off -- MI355X_MICROARCH.md prices those
// at 4-7 us, no better than the launch they would replace).
//
// A conv inside the list (vae.py:53-71 Block convs, :165-167 z_proj / z_feat_proj, their data gradients):
//   * the image's whole input -- every segment of the virtual torch.cat, zero halo -- is staged ONCE by global->LDS DMA
//     (linear layout, 16-byte channel groups, odd group count per pixel), activation applied in place;
//   * the weights of a pass's <= 4 channel tiles stream through a two-slot LDS ring shared by the eight waves: one DMA
//     instruction = 8 rows x 128 bytes of the weight image (full cache lines) landing in MFMA-fragment order (conflict-free
//     ds_read_b128); chunk c + 1 is in flight under chunk c's MFMAs (the step lambda takes the slots as __restrict__ pointers,
//     so hipcc keeps vmcnt out of the loop); a wave holds <= 4 x 2 accumulator tiles and walks the whole K axis, pixel
//     fragments from the staged image through a host-built K-step table (tap shifts) that sits in LDS too -- no cross-wave
//     reduction, deterministic;
//   * bias is the accumulators' initial value; the epilogue ((.) * act'(aux) + res1 + res2, bf16 rounding, zero fill of the
//     padding channels) goes straight from the accumulators to global memory, 8 bytes per lane.
// The element-wise ops run the SAME bodies as their stand-alone kernels (elementwise_bodies.inc, latent_bodies.inc) over
// one image.  Everything is planned on the host (cgen_stage_plan) into one device blob the engine caches per op list.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "elementwise_bodies.inc"
#include "latent_bodies.inc"

namespace cgen {

typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef short st_s16x2 __attribute__((ext_vector_type(2)));

#define ST_THREADS 512
#define ST_WAVES 8
#define ST_PB 4   // pixel groups per register block
#define ST_TB 2   // channel tiles per register block
#define ST_KC 8     // K-steps per chunk: all of a chunk's weight fragments are in flight together

__device__ uint4 st_zero16[4];  // DMA / load source of everything that is "zero" (halo, padding groups, absent operands)

struct SDiv { uint32_t mul, shift; };
static inline SDiv mk_sdiv(uint32_t d) {  // q = (umulhi(n, mul) + n) >> shift, exact for 0 <= n < 2^31 (round-up method)
  SDiv f;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t sh = 0;
  while ((1u << sh) < d) ++sh;
  f.shift = sh;
  f.mul = (uint32_t)((((uint64_t)1 << (32 + sh)) + d - 1) / d - ((uint64_t)1 << 32));
  return f;
}
__device__ __forceinline__ int sdiv(int n, const SDiv& f) { return (int)(((uint64_t)__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shift); }

struct SView {  // per-image view: 64-bit base, 32-bit ELEMENT strides (the host checked that every offset fits)
  char* p;
  int sn, sh, sw, c, cpad, pad0;
};

struct StConv {
  int H, W, KS, halo, nseg, act, dact, Co;
  int C8, Gs, Wp, HW, ngroups, npieces, G, T;
  int nk, krow, k_lo, rows_pad, pixstride, zoff, ktab_off, pad0;
  int PW, TW, npp, ntp, KC, woff, wslot, ktab_bytes;  // wave grid (PW x TW waves, each <= ST_PB pixel groups x ST_TB tiles per pass), passes, K-steps per weight chunk, LDS offset / slot size of the weight ring
  SView seg[CGEN_MAX_SEG];
  int seg_koff[CGEN_MAX_SEG];
  SView out, aux, res1, res2;
  const h16_t* w;
  const float* bias;
  const char* next_w;  // weight image of the next conv of the list (L2 warm-up), or this op's own
  int next_w_bytes, pad1;
  SDiv d_gs, d_wp, d_w;
};

struct StElem {
  Shape4 s;      // n == 1 (one image); h, w, c of the tensor the body iterates over
  int d, hi, wi, accumulate, c_from, vec, pad0, pad1;
  float ish, isw, alpha, beta;
  const float* src;
  View in, out;  // whole-batch views: the kernel shifts them to its image
};

struct StOp {
  int kind, lds_bytes, pad0, pad1;
  union {
    StConv conv;
    StElem elem;
    LatP lat;
    LatBwdP latb;
  };
};

// ----------------------------------------------------------------------------- device: convolution of one image
__device__ __forceinline__ uint4 st_act_group(uint4 v, const int act) {
  if (act == CGEN_ACT_RELU) {  // bf16 ReLU == signed 16-bit max(x, 0) on the raw bits
    union { uint32_t u; st_s16x2 s; } c;
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c.u = w[e];
      c.s = __builtin_elementwise_max(c.s, (st_s16x2){0, 0});
      w[e] = c.u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
  return gelu8_fwd_h16(v);
}


#define ST_STAMP(k) do { if (stamp) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)

// The op descriptor of the CURRENT op lives in LDS (ST_OPBUF bytes per slot, two slots: the next op's descriptor is DMA'd in
// while this one runs).  Every lane holds a 64-way slice of it (dword j*64 + lane in opw[j]); a field is then ONE v_readlane
// away, wave-uniform in an SGPR -- instead of a chain of dependent scalar loads from global memory (~0.5 us per miss, five
// deep per op when the table is cold).
#define ST_OPBUF 1024
#define ST_KTAB_OFF 2048 // this op's K-step table (<= 2 KiB: 128 K-steps)
#define ST_IMG_OFF 4096  // [op slot 0 | op slot 1 | K-step table] then the staged image, then the two slots of the weight ring
struct OpW { uint32_t w[ST_OPBUF / 256]; };
__device__ __forceinline__ OpW st_load_op(const char* smem, const int slot, const int lane) {
  OpW o;
  const uint32_t* src = (const uint32_t*)(smem + slot * ST_OPBUF) + lane;
#pragma unroll
  for (int j = 0; j < ST_OPBUF / 256; ++j) o.w[j] = src[j * 64];
  return o;
}
#define ST_FIELD_DW(member) ((int)((offsetof(StOp, conv) + offsetof(StConv, member)) / 4))
#define SI(member) ((int)__builtin_amdgcn_readlane((int)ow.w[ST_FIELD_DW(member) / 64], ST_FIELD_DW(member) % 64))
#define SI_AT(member, idx) ((int)__builtin_amdgcn_readlane((int)ow.w[(ST_FIELD_DW(member) + (idx)) / 64], (ST_FIELD_DW(member) + (idx)) % 64))
__device__ __forceinline__ const char* st_ptr(const int lo, const int hi) { return (const char*)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }
#define SP(member) st_ptr(SI_AT(member, 0), SI_AT(member, 1))

__device__ __forceinline__ void st_conv(const OpW& ow, const char* __restrict__ blob, const int n, char* __restrict__ smem, unsigned long long* stamp, uint32_t (&pf)[3]) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kg = lane >> 4;
  char* img = smem + ST_IMG_OFF;

  const int H = SI(H), W = SI(W), halo = SI(halo), Gs = SI(Gs), Wp = SI(Wp), ngroups = SI(ngroups), npieces = SI(npieces);
  // ---- stage the image: every 16-byte group of the halo'd, concatenated input, by LDS-DMA (zeros from the zero page).
  // The loop body is ONE basic block (selects only): with a branch inside, hipcc drains vmcnt at the loop header and the
  // DMAs go out one round trip at a time.
  {
    const SDiv d_gs = {(uint32_t)SI(d_gs.mul), (uint32_t)SI(d_gs.shift)}, d_wp = {(uint32_t)SI(d_wp.mul), (uint32_t)SI(d_wp.shift)};
    const int ko1 = SI_AT(seg_koff, 1), ko2 = SI_AT(seg_koff, 2), ko3 = SI_AT(seg_koff, 3);  // (absent segments: 1 << 30)
    const uint64_t z64 = (uint64_t)(uintptr_t)st_zero16;
    uint64_t b[CGEN_MAX_SEG];
    int sh[CGEN_MAX_SEG], sw[CGEN_MAX_SEG], sc[CGEN_MAX_SEG];
#define ST_SEG(k) b[k] = (uint64_t)(uintptr_t)SP(seg[k].p) + (uint64_t)((int64_t)n * SI(seg[k].sn) * 2); sh[k] = SI(seg[k].sh); sw[k] = SI(seg[k].sw); sc[k] = SI(seg[k].c);
    ST_SEG(0) ST_SEG(1) ST_SEG(2) ST_SEG(3)
#undef ST_SEG
    if (wave == ST_WAVES - 2) {  // this op's K-step table -> LDS (read back with ds_read in the K loop: no global round trip there)
      const char* kt_src = blob + SI(ktab_off);
      const int ktb = SI(ktab_bytes);
      for (int o = 0; o < ktb; o += 1024)
        __builtin_amdgcn_global_load_lds((gbl_ptr)(kt_src + o + lane * 16), (lds_ptr)(smem + ST_KTAB_OFF + o), 16, 0, 0);
    }
    for (int pi = wave; pi < npieces; pi += ST_WAVES) {
      const int q = pi * 64 + lane;
      const int pix = sdiv(q, d_gs), grp = q - pix * Gs;
      const int yy = sdiv(pix, d_wp), xx = pix - yy * Wp;
      const int y = yy - halo, x = xx - halo;
      const int c8 = grp * 8;
      const bool s1 = c8 >= ko1, s2 = c8 >= ko2, s3 = c8 >= ko3;
      const uint64_t base = s3 ? b[3] : (s2 ? b[2] : (s1 ? b[1] : b[0]));
      const int vsh = s3 ? sh[3] : (s2 ? sh[2] : (s1 ? sh[1] : sh[0]));
      const int vsw = s3 ? sw[3] : (s2 ? sw[2] : (s1 ? sw[1] : sw[0]));
      const int vsc = s3 ? sc[3] : (s2 ? sc[2] : (s1 ? sc[1] : sc[0]));
      const int ko = s3 ? ko3 : (s2 ? ko2 : (s1 ? ko1 : 0));
      const int cs = c8 - ko;
      const int ok = (int)(q < ngroups) & (int)((unsigned)y < (unsigned)H) & (int)((unsigned)x < (unsigned)W) & (int)(cs < vsc);
      const uint64_t m = (uint64_t)0 - (uint64_t)ok;  // all ones / zero
      const uint64_t cand = base + (uint64_t)(int64_t)((y * vsh + x * vsw + cs) * 2);
      const uint64_t src = (cand & m) | (z64 & ~m);
      __builtin_amdgcn_global_load_lds((gbl_ptr)(uintptr_t)src, (lds_ptr)(img + pi * 1024), 16, 0, 0);
    }
  }
  ST_STAMP(1);
  __syncthreads();  // (hipcc drains vmcnt before the barrier: the image has landed)
  ST_STAMP(2);
  const int act = SI(act);
  if (act != CGEN_ACT_NONE) {
    uint4* im = (uint4*)img;
    for (int q = tid; q < ngroups; q += ST_THREADS) im[q] = st_act_group(im[q], act);
    __syncthreads();
  }

  ST_STAMP(3);
  const int Co = SI(Co), HW = SI(HW), nk = SI(nk), krow = SI(krow), k_lo = SI(k_lo), rows_pad = SI(rows_pad), pixstride = SI(pixstride), zoff = SI(zoff);
  const int G = SI(G), T = SI(T), PW = SI(PW), TW = SI(TW), npp = SI(npp), ntp = SI(ntp), KC = SI(KC), woff = SI(woff), wslot = SI(wslot);
  const int dact = SI(dact), out_cpad = SI(out.cpad);
  const SDiv d_w = {(uint32_t)SI(d_w.mul), (uint32_t)SI(d_w.shift)};
  const h16_t* wimg = (const h16_t*)SP(w);
  const float* bias = (const float*)SP(bias);
  const char* outp = SP(out.p) + (int64_t)n * SI(out.sn) * 2;
  const char* auxp0 = SP(aux.p);
  const char* r1p0 = SP(res1.p);
  const char* r2p0 = SP(res2.p);
  const char* auxp = auxp0 ? auxp0 + (int64_t)n * SI(aux.sn) * 2 : nullptr;
  const char* r1p = r1p0 ? r1p0 + (int64_t)n * SI(res1.sn) * 2 : nullptr;
  const char* r2p = r2p0 ? r2p0 + (int64_t)n * SI(res2.sn) * 2 : nullptr;
  const int out_sh = SI(out.sh), out_sw = SI(out.sw), aux_sh = SI(aux.sh), aux_sw = SI(aux.sw);
  const int r1_sh = SI(res1.sh), r1_sw = SI(res1.sw), r2_sh = SI(res2.sh), r2_sw = SI(res2.sw);
  const int nchunks = (nk + KC - 1) / KC, KCP = KC >> 1;
  char* ring = smem + woff;
  const char* ktl = smem + ST_KTAB_OFF;
  // weight ring, producer side.  One DMA instruction = 8 rows x 128 bytes (two K-steps, FULL cache lines) of the weight image
  // -> 1 KiB of LDS; this lane fetches (row r8, K-step parity kpar, K group kg8) into slot `lane`, and that slot is where the
  // consuming lane (r & 7 == r8, kg == kg8, odd / even K-step) reads it back: conflict-free by construction.
  const int r8 = lane & 7, kg8 = (lane >> 3) & 3, kpar = lane >> 5;
  const int wlane = r8 * krow + kpar * 32 + kg8 * 8;  // element offset of this lane's 16 bytes inside (8-row block, K-step pair)
  // consumer side: lane (r, kg) of an MFMA A fragment
  const int wrd = (r >> 3) * 1024 + ((r & 7) + 8 * kg) * 16;
  const int ip = wave % PW, it = wave / PW;
  const bool wave_on = wave < PW * TW;

  for (int tp = 0; tp < ntp; ++tp) {
    const int tbase = tp * TW * ST_TB;
    const int tiles_pass = min(TW * ST_TB, T - tbase);
    const int nfrag = tiles_pass * KC;  // 1-KiB DMA instructions per chunk
    auto issue_w = [&](char* __restrict__ dst, const int c) {
      for (int f = wave; f < nfrag; f += ST_WAVES) {
        const int blk = f & 1, kp = (f >> 1) % KCP, j = (f >> 1) / KCP;
        const int row8 = min((tbase + j) * 16 + blk * 8, rows_pad - 8);            // (rows past the image: clamped, never stored)
        const int kst = min(c * KC + 2 * kp, ((nk + 1) & ~1) - 2);                  // (K-step pairs past the end: the last pair again, never used)
        const h16_t* src = wimg + ((int64_t)row8 * krow + k_lo + kst * 32 + wlane);
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(dst + f * 1024), 16, 0, 0);
      }
    };
    const int t0 = tbase + it * ST_TB;
    const int nt = wave_on ? max(0, min(ST_TB, min(T, tbase + tiles_pass) - t0)) : 0;
    st_f32x4 binit[ST_TB];
#pragma unroll
    for (int j = 0; j < ST_TB; ++j) {
      const int co = (t0 + j) * 16 + kg * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bias[max(0, min(co + e, Co - 1))];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = (co + e < Co) ? bv[e] : 0.f;
      }
      binit[j] = (st_f32x4){bv[0], bv[1], bv[2], bv[3]};
    }
    const int wtile = (t0 - tbase) * KCP * 2048;  // this wave's first tile inside a ring slot
    for (int pp = 0; pp < npp; ++pp) {
      const int g0 = (pp * PW + ip) * ST_PB;
      const int np = (wave_on && nt > 0) ? max(0, min(ST_PB, G - g0)) : 0;
      int pbase[ST_PB], py[ST_PB], px[ST_PB];
      bool pv[ST_PB];
#pragma unroll
      for (int i = 0; i < ST_PB; ++i) {
        int p = (g0 + i) * 16 + r;
        pv[i] = i < np && p < HW;
        p = max(0, min(p, HW - 1));
        py[i] = sdiv(p, d_w);
        px[i] = p - py[i] * W;
        pbase[i] = (py[i] * Wp + px[i]) * pixstride;
      }
      st_f32x4 acc[ST_PB][ST_TB];
#pragma unroll
      for (int i = 0; i < ST_PB; ++i)
#pragma unroll
        for (int j = 0; j < ST_TB; ++j) acc[i][j] = binit[j];

      // ---- K loop: the weights of chunk c + 1 (all tiles of this pass, shared by the eight waves) travel global -> LDS
      // while chunk c is multiplied out of the other ring slot; pixel fragments come from the staged image.  The ring slots,
      // the image and the K-step table are handed to the step as __restrict__ pointers (hipcc would otherwise drain the DMA
      // in flight before every LDS read it cannot prove disjoint -- DESIGN 3.7).
      __syncthreads();  // every wave is done with both slots (previous pass) before chunk 0 overwrites slot 0
      issue_w(ring, 0);
      auto step = [&](const char* __restrict__ cur, char* __restrict__ nxt, const char* __restrict__ imgp, const char* __restrict__ ktp, const int c) {
        if (c + 1 < nchunks) issue_w(nxt, c + 1);
        const int nkc = min(KC, nk - c * KC);
        for (int d = 0; d < nkc; ++d) {
          const int ks = c * KC + d;
          const int ko = *(const int*)(ktp + (ks * 4 + kg) * 4);
          h16x8 wfr[ST_TB];
#pragma unroll
          for (int j = 0; j < ST_TB; ++j)
            if (j < nt) wfr[j] = *(const h16x8*)(cur + wtile + (j * KCP + (d >> 1)) * 2048 + (d & 1) * 512 + wrd);
          h16x8 a[ST_PB];
#pragma unroll
          for (int i = 0; i < ST_PB; ++i)
            if (i < np) a[i] = *(const h16x8*)(imgp + (ko < 0 ? zoff : pbase[i] + ko));
#pragma unroll
          for (int i = 0; i < ST_PB; ++i)
            if (i < np) {
#pragma unroll
              for (int j = 0; j < ST_TB; ++j)
                if (j < nt) acc[i][j] = mfma_h16(wfr[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
      };
      for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // chunk c has landed (hipcc drains vmcnt here); every wave is done reading the slot chunk c + 1 goes to
        step(ring + (c & 1) * wslot, ring + ((c & 1) ^ 1) * wslot, img, ktl, c);
      }

      // epilogue operands of the block (requested together, one round trip)
      uint2 ea[ST_PB][ST_TB], er1[ST_PB][ST_TB], er2[ST_PB][ST_TB];
#pragma unroll
      for (int i = 0; i < ST_PB; ++i)
#pragma unroll
        for (int j = 0; j < ST_TB; ++j) {
          const int co = (t0 + j) * 16 + kg * 4;
          const bool live = pv[i] && j < nt && co + 4 <= Co;
          ea[i][j] = er1[i][j] = er2[i][j] = make_uint2(0, 0);
          if (auxp) ea[i][j] = *(const uint2*)(live ? auxp + (py[i] * aux_sh + px[i] * aux_sw + co) * 2 : (const char*)st_zero16);
          if (r1p) er1[i][j] = *(const uint2*)(live ? r1p + (py[i] * r1_sh + px[i] * r1_sw + co) * 2 : (const char*)st_zero16);
          if (r2p) er2[i][j] = *(const uint2*)(live ? r2p + (py[i] * r2_sh + px[i] * r2_sw + co) * 2 : (const char*)st_zero16);
        }
      ST_STAMP(4);
      // ---- epilogue: lane owns channels co .. co+3 of pixel (py, px) of every block tile
#pragma unroll
      for (int i = 0; i < ST_PB; ++i) {
        if (!pv[i]) continue;
        const int o_out = (py[i] * out_sh + px[i] * out_sw) * 2;
#pragma unroll
        for (int j = 0; j < ST_TB; ++j) {
          if (j >= nt) continue;
          const int co = (t0 + j) * 16 + kg * 4;
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          if (co + 4 <= Co) {
            if (auxp) {
              const uint2 a2 = ea[i][j];
              const float av[4] = {h_lo(a2.x), h_hi(a2.x), h_lo(a2.y), h_hi(a2.y)};
              if (dact == CGEN_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = av[e] > 0.f ? v[e] : 0.f;
              } else if (dact == CGEN_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float cdf, pdf;
                  gelu_terms_fast(av[e], cdf, pdf);
                  v[e] *= cdf + av[e] * pdf;
                }
              }
            }
            if (r1p) {
              const uint2 t = er1[i][j];
              v[0] += h_lo(t.x); v[1] += h_hi(t.x); v[2] += h_lo(t.y); v[3] += h_hi(t.y);
            }
            if (r2p) {
              const uint2 t = er2[i][j];
              v[0] += h_lo(t.x); v[1] += h_hi(t.x); v[2] += h_lo(t.y); v[3] += h_hi(t.y);
            }
            uint2 o;
            o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]);
            *(uint2*)(outp + o_out + co * 2) = o;
          } else {  // ragged width: element by element; channels [Co, out.cpad) are written as zeros
            const int o_aux = (py[i] * aux_sh + px[i] * aux_sw) * 2, o_r1 = (py[i] * r1_sh + px[i] * r1_sw) * 2, o_r2 = (py[i] * r2_sh + px[i] * r2_sw) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ce = co + e;
              if (ce < Co) {
                float u = v[e];
                if (auxp) u *= act_bwd(dact, h2f(*(const h16_t*)(auxp + o_aux + ce * 2)));
                if (r1p) u += h2f(*(const h16_t*)(r1p + o_r1 + ce * 2));
                if (r2p) u += h2f(*(const h16_t*)(r2p + o_r2 + ce * 2));
                *(h16_t*)(outp + o_out + ce * 2) = f2h(u);
              } else if (ce < out_cpad) {
                *(h16_t*)(outp + o_out + ce * 2) = 0;
              }
            }
          }
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------- device: element-wise ops of one image
__device__ __forceinline__ View st_shift(View v, const int n) {
  if (v.p) v.p += (int64_t)n * v.sn * 2;
  return v;
}

template <int V>
__device__ __forceinline__ void st_elem(const int kind, const StElem& e, const int n) {
  typedef h16_t T;
  const int64_t g0 = threadIdx.x, gs = ST_THREADS;
  const View in = st_shift(e.in, n), out = st_shift(e.out, n);
  switch (kind) {
    case CGEN_ST_AVGPOOL_FWD: avgpool_fwd_body<T, V>(g0, gs, e.s, e.d, in, out); break;
    case CGEN_ST_AVGPOOL_BWD: avgpool_bwd_body<T, V>(g0, gs, e.s, e.d, in, out, e.accumulate); break;
    case CGEN_ST_UPSAMPLE_FWD: upsample_fwd_body<T, V>(g0, gs, e.s, e.hi, e.wi, e.ish, e.isw, in, e.src, out); break;
    case CGEN_ST_UPSAMPLE_BWD: upsample_bwd_body<T, V>(g0, gs, e.s, e.hi, e.wi, e.ish, e.isw, in, out, e.accumulate); break;
    case CGEN_ST_BCAST: batch_broadcast_body<T, V>(g0, gs, e.s, e.src, out); break;
    default: axpby_body<T, V>(g0, gs, e.s, in, out, e.alpha, e.beta, e.c_from, e.accumulate); break;
  }
}

// reparameterise + KL of one image: the stand-alone kernel's per-thread item over the image's chunks, two chunks at a time
// (one per 256-thread half), each half reduced in block_sum_256's order -> the same partials bit for bit
__device__ __forceinline__ void st_reparam_fwd(const LatP& p, const int n, float* red) {
  const int tid = threadIdx.x, half = tid >> 8, t = tid & 255;
  const int nchunks = __builtin_amdgcn_readfirstlane((p.h * p.w * p.c + LAT_CHUNK - 1) / LAT_CHUNK);
  uint64_t seed = 0, off = 0;
  if (!p.eps_in.p) { seed = p.rng[0]; off = p.rng[1]; }
  for (int c0 = 0; c0 < nchunks; c0 += 2) {
    const int chunk = c0 + half;
    float v = 0.f;
    if (chunk < nchunks) v = reparam_kl_fwd_vec8_item(p, n, chunk, t, seed, off);
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (t == 0 && chunk < nchunks) p.kl_part[(int64_t)n * p.kl_stride + chunk] = (red[half * 4] + red[half * 4 + 1]) + (red[half * 4 + 2] + red[half * 4 + 3]);
    __syncthreads();
  }
}

__device__ __forceinline__ void st_reparam_bwd(const LatBwdP& p, const int n) {
  const int per8 = (p.h * p.w * p.c) >> 3;
  for (int g = threadIdx.x; g < per8; g += ST_THREADS) reparam_kl_bwd_vec8_item(p, (int64_t)n * per8 + g);
  if (p.ride_src.p) {
    const int rper = p.h * p.w * (p.ride_c >> 3);
    for (int g = threadIdx.x; g < rper; g += ST_THREADS) reparam_ride_item(p, (int64_t)n * rper + g);
  }
}

static_assert(sizeof(StOp) <= ST_OPBUF, "an op descriptor must fit its LDS slot");


__device__ __forceinline__ void st_fetch_op(const char* __restrict__ blob, const int i, char* smem, const int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  // one wave-wide DMA: 64 lanes x 16 bytes = the whole slot (bytes past sizeof(StOp) belong to the next op or the tables:
  // readable, ignored)
  __builtin_amdgcn_global_load_lds((gbl_ptr)(blob + (size_t)i * sizeof(StOp) + lane * 16), (lds_ptr)(smem + (i & 1) * ST_OPBUF), 16, 0, 0);
}

__global__ __launch_bounds__(ST_THREADS) void stage_kernel(const char* __restrict__ blob, const int nops, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[ST_WAVES];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) st_fetch_op(blob, 0, smem, lane);
  __syncthreads();
  for (int i = 0; i < nops; ++i) {
    unsigned long long* stamp = (stamps && n == 0 && threadIdx.x == 0 && i < 256) ? stamps + 8 * i : nullptr;  // CGEN_STAGE_STAMPS: 100 MHz clock per phase, workgroup 0
    ST_STAMP(0);
    const int slot = i & 1;
    const OpW ow = st_load_op(smem, slot, lane);
    const int kind = __builtin_amdgcn_readlane((int)ow.w[0], 0);
    uint32_t pf[3] = {0, 0, 0};
    if (kind == CGEN_ST_CONV) {
      if (wave == ST_WAVES - 1 && i + 1 < nops) st_fetch_op(blob, i + 1, smem, lane);  // lands with this op's first barrier
      st_conv(ow, blob, n, smem, stamp, pf);
      ST_STAMP(5);
    } else {
      // element-wise / latent ops: the descriptor is copied out of LDS (per-lane registers), then the next one is fetched
      const char* od = smem + slot * ST_OPBUF;
      if (kind == CGEN_ST_REPARAM_FWD) {
        const LatP p = *(const LatP*)(od + offsetof(StOp, lat));
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the copy is in registers before the slot's neighbour is overwritten... (and before any DMA is in flight)
        if (wave == ST_WAVES - 1 && i + 1 < nops) st_fetch_op(blob, i + 1, smem, lane);
        st_reparam_fwd(p, n, red);
      } else if (kind == CGEN_ST_REPARAM_BWD) {
        const LatBwdP p = *(const LatBwdP*)(od + offsetof(StOp, latb));
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (wave == ST_WAVES - 1 && i + 1 < nops) st_fetch_op(blob, i + 1, smem, lane);
        st_reparam_bwd(p, n);
      } else {
        const StElem e = *(const StElem*)(od + offsetof(StOp, elem));
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (wave == ST_WAVES - 1 && i + 1 < nops) st_fetch_op(blob, i + 1, smem, lane);
        if (e.vec) st_elem<4>(kind, e, n);
        else st_elem<1>(kind, e, n);
      }
    }
    __syncthreads();  // the next op reads what this one wrote (same workgroup: workgroup scope is enough) and reuses the LDS
    asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]));  // (the L2 warm-up loads end here)
    ST_STAMP(6);
    if (stamp) stamp[7] = (unsigned long long)kind;
  }
}

// ----------------------------------------------------------------------------- host: eligibility and planning
static inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t span_elems(const cgen_view& v, int n, int h, int w) {
  return (int64_t)(n - 1) * v.sn + (int64_t)(h - 1) * v.sh + (int64_t)(w - 1) * v.sw + std::max(v.c, v.cpad);
}
static bool sview(const cgen_view& v, int n, int h, int w, SView& o) {
  memset(&o, 0, sizeof(o));
  if (!v.p) return true;
  if (v.sn < 0 || v.sh < 0 || v.sw < 0 || v.sn >= ((int64_t)1 << 30) || span_elems(v, n, h, w) * 2 >= ((int64_t)1 << 31)) return false;
  if ((int64_t)(h - 1) * v.sh + (int64_t)(w - 1) * v.sw + std::max(v.c, v.cpad) >= ((int64_t)1 << 30)) return false;
  o.p = (char*)v.p; o.sn = (int)v.sn; o.sh = (int)v.sh; o.sw = (int)v.sw; o.c = v.c; o.cpad = v.cpad;
  return true;
}
static inline bool al(const cgen_view& v, int bytes) {  // base and every stride multiples of `bytes`
  if (!v.p) return true;
  return ((uintptr_t)v.p % bytes) == 0 && (v.sn * 2) % bytes == 0 && (v.sh * 2) % bytes == 0 && (v.sw * 2) % bytes == 0;
}

static int stage_budget() {
  static const int b = [] { const char* e = getenv("CGEN_STAGE_MFMA"); return e ? atoi(e) : 4096; }();
  return b;
}

// fills `o` (without the K-step table); returns the LDS bytes the conv needs, 0 when the op is not served
static int plan_conv(const cgen_conv_args* a, StConv& o, std::vector<int>* ktab) {
  memset(&o, 0, sizeof(o));
  if (a->dtype != CGEN_F16 || (a->ks != 1 && a->ks != 3) || a->nseg < 1 || a->nseg > CGEN_MAX_SEG || a->n < 1 || a->h < 1 || a->w < 1) return 0;
  if (!a->weight || !a->out.p || a->out.c < 1) return 0;
  if (a->out_rem || a->res1_rem) return 0;  // (remainder planes of the f16 residual trunk: the stand-alone kernels serve those)
  o.H = a->h; o.W = a->w; o.KS = a->ks; o.halo = a->ks / 2; o.nseg = a->nseg; o.act = a->act; o.dact = a->dact; o.Co = a->out.c;
  int c8 = 0;
  for (int s = 0; s < a->nseg; ++s) {
    const cgen_view& v = a->seg[s];
    if (!v.p || v.c < 1 || !al(v, 16)) return 0;
    if (v.c % 8 != 0 && v.cpad < pad_to(v.c, 8)) return 0;  // whole 16-byte groups are fetched: ragged widths need zero padding
    if (!sview(v, a->n, a->h, a->w, o.seg[s])) return 0;
    o.seg_koff[s] = c8;
    c8 += pad_to(v.c, 8);
  }
  for (int s = a->nseg; s < CGEN_MAX_SEG; ++s) { o.seg[s] = o.seg[0]; o.seg_koff[s] = 1 << 30; }
  o.C8 = c8;
  if (!al(a->out, 8) || !al(a->aux, 8) || !al(a->res1, 8) || !al(a->res2, 8)) return 0;
  if (!sview(a->out, a->n, a->h, a->w, o.out) || !sview(a->aux, a->n, a->h, a->w, o.aux) || !sview(a->res1, a->n, a->h, a->w, o.res1) ||
      !sview(a->res2, a->n, a->h, a->w, o.res2)) return 0;
  if ((a->aux.p && a->aux.c != o.Co) || (a->res1.p && a->res1.c != o.Co) || (a->res2.p && a->res2.c != o.Co)) return 0;
  o.out.cpad = a->out.cpad;
  const int taps = a->ks * a->ks;
  int tap0 = 0, tap1 = taps;
  if (a->h == 1 && a->w == 1 && a->ks > 1) { tap0 = taps / 2; tap1 = tap0 + 1; }
  o.k_lo = tap0 * c8;
  const int k_hi = tap1 * c8;
  o.nk = (k_hi - o.k_lo + 31) / 32;
  o.krow = pad_to(taps * c8, 32) + 32;
  o.rows_pad = pad_to(o.Co, 16);
  o.Gs = c8 / 8;
  if ((o.Gs & 1) == 0) ++o.Gs;
  o.pixstride = o.Gs * 16;
  o.Wp = a->w + 2 * o.halo;
  const int Hp = a->h + 2 * o.halo;
  o.HW = a->h * a->w;
  o.ngroups = Hp * o.Wp * o.Gs;
  o.npieces = (o.ngroups + 1 + 63) / 64;  // + the zero slot right behind the image
  o.zoff = o.ngroups * 16;
  if (ST_IMG_OFF + o.npieces * 1024 > 150 * 1024) return 0;
  o.G = (o.HW + 15) / 16;
  o.T = (o.Co + 15) / 16;
  if ((int64_t)o.G * o.T * o.nk > stage_budget()) return 0;  // too much work for one CU per image: a chip-wide launch is faster
  o.w = (const h16_t*)a->weight;
  o.bias = a->bias;
  o.next_w = (const char*)a->weight;  // (cgen_stage_plan points it at the next conv of the list)
  o.next_w_bytes = 0;
  o.d_gs = mk_sdiv(o.Gs); o.d_wp = mk_sdiv(o.Wp); o.d_w = mk_sdiv(a->w);
  // ---- pass plan.  PW x TW waves; in one pass a wave owns <= ST_PB pixel groups x ST_TB channel tiles (its accumulators) and
  // walks the whole K axis; the weights of the pass's TW * ST_TB tiles stream through a two-slot LDS ring shared by all waves, KC
  // K-steps per slot.  Passes partition the channel tiles (every weight is fetched once per pixel pass).
  {
    double best = 1e30;
    int bpw = 1, btw = 1;
    static const int cand[][2] = {{8, 1}, {4, 2}, {2, 2}, {4, 1}, {2, 1}, {1, 2}, {1, 1}};
    for (auto& pt : cand) {
      const int pw = pt[0], tw = pt[1];
      if (pw > 1 && (pw / 2) * ST_PB >= o.G && !(pw == 2 && tw == 2)) { /* more pixel columns than the image needs */ }
      const int npp = (o.G + pw * ST_PB - 1) / (pw * ST_PB), ntp = (o.T + tw * ST_TB - 1) / (tw * ST_TB);
      const int gw = std::min(ST_PB, (o.G + pw * npp - 1) / (pw * npp)), tw_t = std::min(ST_TB, (o.T + tw * ntp - 1) / (tw * ntp));
      // per pass: nk K-steps of (gw x tw_t MFMAs + gw + tw_t LDS reads) on the busiest wave, two waves per SIMD when all 8 run
      const double per = (double)o.nk * (gw * tw_t * 16.0 + (gw + tw_t) * 8.0) * (pw * tw > 4 ? 2.0 : 1.0) + 1500.0;
      const double cost = (double)npp * ntp * per;
      if (cost < best) { best = cost; bpw = pw; btw = tw; }
    }
    o.PW = bpw; o.TW = btw;
    o.npp = (o.G + bpw * ST_PB - 1) / (bpw * ST_PB);
    o.ntp = (o.T + btw * ST_TB - 1) / (btw * ST_TB);
  }
  o.ktab_bytes = o.nk * 16;
  if (o.ktab_bytes > ST_IMG_OFF - ST_KTAB_OFF) return 0;
  const int img_bytes = o.npieces * 1024;
  o.woff = ST_IMG_OFF + img_bytes;
  {
    const int tiles_pass = std::min(o.TW * ST_TB, o.T);
    int kc = 16;
    while (kc > 2 && (o.woff + 2 * tiles_pass * kc * 1024 > 156 * 1024 || kc >= 2 * (((o.nk + 1) & ~1)))) kc -= 2;
    if (o.woff + 2 * tiles_pass * kc * 1024 > 156 * 1024) return 0;
    o.KC = kc;
    o.wslot = tiles_pass * kc * 1024;
  }
  const int lds_total = o.woff + 2 * o.wslot;
  if (ktab) {
    ktab->clear();
    for (int ks = 0; ks < o.nk; ++ks)
      for (int g4 = 0; g4 < 4; ++g4) {
        const int kidx = o.k_lo + ks * 32 + g4 * 8;
        if (kidx >= k_hi) { ktab->push_back(-1); continue; }
        const int tap = kidx / c8, cc = kidx - tap * c8;
        const int dy = tap / a->ks, dx = tap - dy * a->ks;
        ktab->push_back((dy * o.Wp + dx) * o.pixstride + cc * 2);
      }
  }
  return lds_total;
}

static bool flat_ok(const cgen_view& v, int n, int h, int w) {
  if (!v.p) return true;
  return v.sn >= 0 && v.sh >= 0 && v.sw >= 0 && span_elems(v, n, h, w) * 2 < ((int64_t)1 << 31);
}
static bool vec4v(const cgen_view& v) { return !v.p || (((uintptr_t)v.p % 8) == 0 && (v.sn * 2) % 8 == 0 && (v.sh * 2) % 8 == 0 && (v.sw * 2) % 8 == 0); }
static inline float inv_scale(int out, int in) { return (float)(1.0 / ((double)out / (double)in)); }

static int plan_elem(int kind, const cgen_stage_elem_args* a, StElem& o) {
  memset(&o, 0, sizeof(o));
  if (a->dtype != CGEN_F16 || a->n < 1 || !a->out.p) return 0;
  const int c = a->out.c;
  int ih = a->h, iw = a->w;  // the tensor the body iterates over
  switch (kind) {
    case CGEN_ST_AVGPOOL_FWD: if (!a->in.p || a->d < 1 || a->in.c != c) return 0; break;
    case CGEN_ST_AVGPOOL_BWD: if (!a->in.p || a->d < 1 || a->in.c != c) return 0; ih = a->h * a->d; iw = a->w * a->d; break;
    case CGEN_ST_UPSAMPLE_FWD: if (!a->in.p || a->in.c != c || a->hi < 1 || a->wi < 1 || a->h < a->hi || a->w < a->wi) return 0; break;
    case CGEN_ST_UPSAMPLE_BWD: if (!a->in.p || a->in.c != c) return 0; ih = a->hi; iw = a->wi; break;
    case CGEN_ST_BCAST: if (!a->src) return 0; break;
    case CGEN_ST_AXPBY: if (a->in.p && a->in.c != c) return 0; break;
    default: return 0;
  }
  const int big_h = std::max(a->h, std::max(ih, a->hi)) * std::max(1, a->d), big_w = std::max(a->w, std::max(iw, a->wi)) * std::max(1, a->d);
  if (!flat_ok(a->in, a->n, big_h, big_w) || !flat_ok(a->out, a->n, big_h, big_w)) return 0;
  o.s = Shape4{1, ih, iw, c};
  o.d = a->d; o.hi = a->hi; o.wi = a->wi; o.accumulate = a->accumulate; o.c_from = a->c_from;
  o.alpha = a->alpha; o.beta = a->beta; o.src = a->src;
  o.in = mk(a->in); o.out = mk(a->out);
  o.vec = (c % 4 == 0 && vec4v(a->in) && vec4v(a->out)) ? 1 : 0;
  if (kind == CGEN_ST_UPSAMPLE_FWD) { o.ish = inv_scale(a->h, a->hi); o.isw = inv_scale(a->w, a->wi); }
  if (kind == CGEN_ST_UPSAMPLE_BWD) { o.hi = a->h; o.wi = a->w; o.ish = inv_scale(a->h, a->hi); o.isw = inv_scale(a->w, a->wi); }
  return 1;
}

static bool lat_ok(int n, int c, std::initializer_list<cgen_view> vs) {
  if (c % 8) return false;
  for (const cgen_view& v : vs) {
    if (!v.p) continue;
    if (!vec16_ok(v, 2)) return false;
    if ((int64_t)n * v.sn >= ((int64_t)1 << 31)) return false;
  }
  return true;
}

static int plan_reparam(const cgen_stage_reparam_args* a, LatP& p) {
  if (a->dtype != CGEN_F16 || !(a->q_loc.p && a->q_ls.p && a->p_loc.p && a->p_ls.p && a->z.p && a->kl_part) || !(a->eps_in.p || a->rng)) return 0;
  if (!lat_ok(a->n, a->c, {a->q_loc, a->q_ls, a->p_loc, a->p_ls, a->eps_in, a->z})) return 0;
  memset(&p, 0, sizeof(p));
  p.n = a->n; p.h = a->h; p.w = a->w; p.c = a->c;
  p.q_loc = mk(a->q_loc); p.q_ls = mk(a->q_ls); p.p_loc = mk(a->p_loc); p.p_ls = mk(a->p_ls);
  p.eps_in = mk(a->eps_in); p.z = mk(a->z);
  p.rng = a->rng; p.stream_id = a->stream_id; p.logt = a->logt; p.kl_part = a->kl_part; p.kl_stride = a->kl_stride;
  return 1;
}

static int plan_reparam_bwd(const cgen_stage_reparam_bwd_args* a, LatBwdP& p) {
  if (a->dtype != CGEN_F16 || !(a->q_loc.p && a->q_ls.p && a->p_loc.p && a->p_ls.p && a->kl_coef_dev && a->g_q_loc.p && a->g_q_ls.p && a->g_p_loc.p && a->g_p_ls.p)) return 0;
  if (a->gz.p && !a->z.p) return 0;
  if (!lat_ok(a->n, a->c, {a->q_loc, a->q_ls, a->p_loc, a->p_ls, a->z, a->gz, a->g_q_loc, a->g_q_ls, a->g_p_loc, a->g_p_ls})) return 0;
  memset(&p, 0, sizeof(p));
  p.n = a->n; p.h = a->h; p.w = a->w; p.c = a->c;
  p.q_loc = mk(a->q_loc); p.q_ls = mk(a->q_ls); p.p_loc = mk(a->p_loc); p.p_ls = mk(a->p_ls); p.z = mk(a->z); p.gz = mk(a->gz);
  p.g_q_loc = mk(a->g_q_loc); p.g_q_ls = mk(a->g_q_ls); p.g_p_loc = mk(a->g_p_loc); p.g_p_ls = mk(a->g_p_ls);
  p.coef = a->kl_coef_dev; p.chan_scale = a->kl_chan_scale; p.coef_stride = a->coef_stride; p.acc_q = a->acc_q; p.acc_p = a->acc_p; p.logt = a->logt;
  if (a->ride_src.p) {
    if (!a->ride_dst.p || a->ride_src.c != a->ride_dst.c || !lat_ok(a->n, a->ride_src.c, {a->ride_src, a->ride_dst})) return 0;
    p.ride_src = mk(a->ride_src); p.ride_dst = mk(a->ride_dst); p.ride_c = a->ride_src.c; p.ride_acc = a->ride_acc;
  }
  return 1;
}

static int plan_op(int kind, const void* args, StOp& op, std::vector<int>* ktab) {
  memset(&op, 0, sizeof(op));
  op.kind = kind;
  int lds = 0;
  switch (kind) {
    case CGEN_ST_CONV: lds = plan_conv((const cgen_conv_args*)args, op.conv, ktab); break;
    case CGEN_ST_REPARAM_FWD: lds = plan_reparam((const cgen_stage_reparam_args*)args, op.lat); break;
    case CGEN_ST_REPARAM_BWD: lds = plan_reparam_bwd((const cgen_stage_reparam_bwd_args*)args, op.latb); break;
    case CGEN_ST_AVGPOOL_FWD: case CGEN_ST_AVGPOOL_BWD: case CGEN_ST_UPSAMPLE_FWD: case CGEN_ST_UPSAMPLE_BWD: case CGEN_ST_BCAST: case CGEN_ST_AXPBY:
      lds = plan_elem(kind, (const cgen_stage_elem_args*)args, op.elem); break;
    default: return 0;
  }
  if (lds > 0 && lds < ST_IMG_OFF) lds = ST_IMG_OFF;  // (the descriptor slots)
  op.lds_bytes = lds;
  return lds;
}

}  // namespace cgen

using namespace cgen;

extern "C" int cgen_stage_accepts(int32_t kind, const void* args) {
  if (!args) return 0;
  StOp op;
  return plan_op(kind, args, op, nullptr);
}

extern "C" int cgen_stage_plan(const int32_t* kinds, const void* const* args, int32_t count, void* blob_host, int64_t capacity,
                               int64_t* blob_bytes, int32_t* lds_bytes) {
  CGEN_REQUIRE(kinds && args && count > 0 && blob_bytes && lds_bytes, "cgen_stage_plan: bad args");
  std::vector<StOp> ops(count);
  std::vector<int> table, kt;
  const int64_t ops_bytes = (int64_t)count * sizeof(StOp);
  int lds = 16;
  for (int i = 0; i < count; ++i) {
    const int l = plan_op(kinds[i], args[i], ops[i], &kt);
    CGEN_REQUIRE(l > 0, "cgen_stage_plan: op %d (kind %d) is not served (ask cgen_stage_accepts first)", i, kinds[i]);
    lds = std::max(lds, l);
    if (kinds[i] == CGEN_ST_CONV) {
      ops[i].conv.ktab_off = (int)(ops_bytes + (int64_t)table.size() * 4);
      table.insert(table.end(), kt.begin(), kt.end());
    }
  }
  for (int i = 0; i < count; ++i) {  // L2 warm-up target of every conv: the weight image of the next conv of the list
    if (kinds[i] != CGEN_ST_CONV) continue;
    for (int j = i + 1; j < count; ++j)
      if (kinds[j] == CGEN_ST_CONV) {
        ops[i].conv.next_w = (const char*)ops[j].conv.w;
        ops[i].conv.next_w_bytes = ops[j].conv.rows_pad * ops[j].conv.krow * 2;
        break;
      }
  }
  table.resize(table.size() + 256, 0);  // (the last descriptor's slot DMA reads a full KiB)
  const int64_t total = ops_bytes + (int64_t)table.size() * 4;
  *blob_bytes = total;
  *lds_bytes = lds;
  if (!blob_host) return CGEN_OK;
  CGEN_REQUIRE(capacity >= total, "cgen_stage_plan: blob buffer too small (%lld < %lld)", (long long)capacity, (long long)total);
  memcpy(blob_host, ops.data(), ops_bytes);
  if (!table.empty()) memcpy((char*)blob_host + ops_bytes, table.data(), table.size() * 4);
  return CGEN_OK;
}

extern "C" int cgen_stage_run(const void* blob_dev, int32_t count, int32_t n_images, int32_t lds_bytes, cgen_stream_t stream) {
  CGEN_REQUIRE(blob_dev && count > 0 && n_images > 0 && lds_bytes >= 0 && lds_bytes <= 159 * 1024, "cgen_stage_run: bad args");
  static bool once = false;
  if (!once) {  // (the kernel also has 32 bytes of static LDS: the dynamic part may take the rest of the CU's 160 KiB)
    (void)hipFuncSetAttribute((const void*)stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipGetLastError();
    once = true;
  }
  unsigned long long* stamps = nullptr;
  { const char* e = getenv("CGEN_STAGE_STAMPS"); if (e) stamps = (unsigned long long*)strtoull(e, nullptr, 0); }
  hipLaunchKernelGGL(stage_kernel, dim3(n_images), dim3(ST_THREADS), (size_t)lds_bytes, (hipStream_t)stream, (const char*)blob_dev, count, stamps);
  return check_launch("cgen_stage_run");
}
