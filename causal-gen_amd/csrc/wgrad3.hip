// Streaming weight gradient for gfx950 (MI355X), round 5.  Replaces wgrad_tile_mega_kernel (conv.hip) for every shape it serves.
//   dW[co][dy][dx][ci] = sum_px G[px][co] * act(X)[px + (dy-1, dx-1)][ci]        (aten::convolution_backward, weight part;
//                                                                                  reference: src/vae.py:53-55, 63-65 through autograd)
//
// GEMM view: the PIXEL axis is the MFMA K dimension (v_mfma_f32_32x32x16_f16, 16 pixels per instruction); both operands sit in
// HBM channels-contiguous (NHWC), so K is strided and the fragments are gathered by the LDS transpose read ds_read_b64_tr_b16
// straight from the NHWC tile (a lane supplies the address of 4 channels of one pixel and receives 4 pixels of one channel).
//
// What is new against the round-2..4 kernel:
//   * The NARROW operand is the shifted one.  A Block's convs are C -> C/4 or C/4 -> C: one operand is 4x narrower.  Writing
//       dW[co][tap][ci] = sum_px P[px][p] * S[px + d(tap)][s]
//     with S = the narrower of (act(X), G) -- for S = G the shift is mirrored, the taps come out flipped (layout 1) -- packs
//     (tap, channel) of S densely into the MFMA N axis at 4-channel granularity (9 x 24 = 216 -> 7 fragments of 32, 96 % full)
//     while P's channels are the M axis (96 = 3 fragments): 21 MFMAs 32x32x16 per 16 pixels where the 16x16x32 tiling needed 54
//     plus 28 half-empty ones; only the narrow tile carries a halo, so the halo re-read costs 8 % of the bytes instead of 40 %.
//   * One workgroup owns the WHOLE [P channels] x [taps x S channels] slab of a pixel range where it fits 4 waves x <= 9
//     fragments (else P is cut into windows): every activation byte is read once per window, not once per (co block x window).
//   * A ring of `nslot` tile buffers filled by untracked global->LDS DMA with counted vmcnt waits and ONE barrier per tile; all
//     lane-dependent address parts are launch constants; ReLU is applied to the fragments in registers (one v_pk_max_i16 per
//     channel pair; GELU keeps an in-LDS pass); bias gradients ride along as v_dot2_f32_f16 sums of the G fragments.
//   * LDS per workgroup <= 48 KB (CGEN_WG3_LDS), so two of them -- or one next to a fused-Block workgroup -- fit a CU.
// Deterministic: every partial slab is written by exactly one wave in a fixed order; cgen_wgrad_reduce sums the slabs in order.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include <atomic>

#include "wgrad3.h"

namespace cgen {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short w3s16x4 __attribute__((ext_vector_type(4)));
typedef short w3s16x2 __attribute__((ext_vector_type(2)));

__device__ uint4 g_w3zero[4];  // DMA source of out-of-image pixels
#ifndef W3_UNIFORM_MODE
#define W3_UNIFORM_MODE 0
#endif
#define W3_PN 6  // DMA instructions per wave and tile whose lane offsets live in registers: P tile, S tile
#define W3_SN 3

static inline W3Div mk_w3div(uint32_t d) {  // round-up method, exact for 0 <= n < 2^31 (as FastDiv in conv.hip)
  W3Div f;
  if (d <= 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t sh = 0;
  while ((1u << sh) < d) ++sh;
  f.shift = sh;
  f.mul = (uint32_t)((((uint64_t)1 << (32 + sh)) + d - 1) / d - ((uint64_t)1 << 32));
  return f;
}
__device__ __forceinline__ int w3div(int n, const W3Div f) {
  return (int)(((uint64_t)__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shift);
}

// one LDS-DMA instruction (16 B per lane -> LDS at `lds` + 16 * lane) the compiler's wait-count pass does not see (block.hip)
__device__ __forceinline__ void w3_dma16(const char* src, const uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");
}
#define W3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// wait until at most n of this wave's DMA requests are outstanding (n is wave-uniform; requests return in order).  s_waitcnt takes
// an immediate: a binary decision tree over 0 .. 47 (six scalar branches; a 48-way switch compiled to a compare chain of ~500
// cycles per tile -- stamps, LABNOTES 10.1); anything above waits for 47, which is more than asked: always safe.
// The tile loop's wait: 16 leaves (a larger count is clamped: waiting for MORE of the requests is always safe, and a wave has 4-12
// per tile).  The loop exists in 91 instances; the 48-leaf tree was a fifth of each one's code, and the flush's mix of instances
// runs out of the instruction cache two CUs share (profiles/r05c: LABNOTES 10.5).
__device__ __forceinline__ void w3_vmwait16(int n) {
#define W3_VMW(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
  if (n < 8) {
    if (n < 4) {
      if (n < 2) { if (n < 1) W3_VMW(0); else W3_VMW(1); } else { if (n < 3) W3_VMW(2); else W3_VMW(3); }
    } else {
      if (n < 6) { if (n < 5) W3_VMW(4); else W3_VMW(5); } else { if (n < 7) W3_VMW(6); else W3_VMW(7); }
    }
  } else {
    if (n < 12) {
      if (n < 10) { if (n < 9) W3_VMW(8); else W3_VMW(9); } else { if (n < 11) W3_VMW(10); else W3_VMW(11); }
    } else {
      if (n < 14) { if (n < 13) W3_VMW(12); else W3_VMW(13); } else { if (n < 15) W3_VMW(14); else W3_VMW(15); }
    }
  }
#undef W3_VMW
}
__device__ __forceinline__ void w3_vmwait(int n) {
  n = n > 47 ? 47 : n;
  if (n < 24) {
    if (n < 12) {
      if (n < 6) {
        if (n < 3) {
          if (n < 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          } else {
            if (n < 2) {
              asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
          }
        } else {
          if (n < 4) {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          } else {
            if (n < 5) {
              asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            }
          }
        }
      } else {
        if (n < 9) {
          if (n < 7) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          } else {
            if (n < 8) {
              asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
          }
        } else {
          if (n < 10) {
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
          } else {
            if (n < 11) {
              asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            }
          }
        }
      }
    } else {
      if (n < 18) {
        if (n < 15) {
          if (n < 13) {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          } else {
            if (n < 14) {
              asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            }
          }
        } else {
          if (n < 16) {
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
          } else {
            if (n < 17) {
              asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            }
          }
        }
      } else {
        if (n < 21) {
          if (n < 19) {
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          } else {
            if (n < 20) {
              asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            }
          }
        } else {
          if (n < 22) {
            asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
          } else {
            if (n < 23) {
              asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
            }
          }
        }
      }
    }
  } else {
    if (n < 36) {
      if (n < 30) {
        if (n < 27) {
          if (n < 25) {
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
          } else {
            if (n < 26) {
              asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
            }
          }
        } else {
          if (n < 28) {
            asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
          } else {
            if (n < 29) {
              asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(29)" ::: "memory");
            }
          }
        }
      } else {
        if (n < 33) {
          if (n < 31) {
            asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
          } else {
            if (n < 32) {
              asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
          }
        } else {
          if (n < 34) {
            asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
          } else {
            if (n < 35) {
              asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(35)" ::: "memory");
            }
          }
        }
      }
    } else {
      if (n < 42) {
        if (n < 39) {
          if (n < 37) {
            asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
          } else {
            if (n < 38) {
              asm volatile("s_waitcnt vmcnt(37)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
            }
          }
        } else {
          if (n < 40) {
            asm volatile("s_waitcnt vmcnt(39)" ::: "memory");
          } else {
            if (n < 41) {
              asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
            }
          }
        }
      } else {
        if (n < 45) {
          if (n < 43) {
            asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
          } else {
            if (n < 44) {
              asm volatile("s_waitcnt vmcnt(43)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
            }
          }
        } else {
          if (n < 46) {
            asm volatile("s_waitcnt vmcnt(45)" ::: "memory");
          } else {
            if (n < 47) {
              asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(47)" ::: "memory");
            }
          }
        }
      }
    }
  }
}

__device__ __forceinline__ h16x8 w3_tr(const uint32_t a0, const uint32_t a1) {
  typedef w3s16x4 __attribute__((address_space(3))) * lp;
  union { w3s16x4 h[2]; h16x8 v; uint32_t w[4]; } u;
#ifdef W3_ABL_NOREAD  // (ablation build, wrong results: fragments made from the addresses, no LDS traffic)
  u.w[0] = a0; u.w[1] = a1; u.w[2] = a0 ^ 0x3c00u; u.w[3] = a1 ^ 0x3c00u;
  asm volatile("" : "+v"(u.w[0]), "+v"(u.w[1]), "+v"(u.w[2]), "+v"(u.w[3]));
#else
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a0);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)a1);
#endif
  return u.v;
}
__device__ __forceinline__ h16x8 w3_relu8(const h16x8 v) {  // ReLU on the raw bits (-0 -> +0; a NaN with the sign bit set -> 0, as block.hip)
  union { h16x8 h; w3s16x2 s[4]; } c;
  c.h = v;
#pragma unroll
  for (int e = 0; e < 4; ++e) c.s[e] = __builtin_elementwise_max(c.s[e], (w3s16x2){0, 0});
  return c.h;
}
// GELU of the eight binary16 values of a 16-byte group, INLINE (common.h's gelu8_fwd_h16 is a real function: a call inside the tile
// loop makes every live register around it caller-saved -- spills in the fragment blocks that have none to spare)
__device__ __forceinline__ uint4 w3_gelu8(const uint4 x) {
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
__device__ __forceinline__ f32x16 w3_mfma(const h16x8 a, const h16x8 b, const f32x16 c) {
#ifdef W3_ABL_NOMFMA  // (ablation build, wrong results: the fragments stay live, no matrix instruction)
  asm volatile("" :: "v"(a), "v"(b));
  return c;
#endif
#ifdef CGEN_H16_BF16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// s + (sum of the 8 values of a fragment register set), f32 accumulate
__device__ __forceinline__ float w3_sum8(const h16x8 v, float s) {
#ifdef CGEN_H16_BF16
#pragma unroll
  for (int e = 0; e < 8; ++e) s += (float)v[e];
  return s;
#else
  union { h16x8 h; h16x2_t p[4]; } c;
  c.h = v;
  const h16x2_t one = {(h16n_t)1.0f, (h16n_t)1.0f};
#pragma unroll
  for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_fdot2(c.p[e], one, s, false);
  return s;
#endif
}

// real (OIHW) input-channel index of 8-granular concatenated channel c8i of X, -1 for padding: a select chain over the segment
// table held in scalars (statically indexed fields only -- a by-value table indexed by a lane-dependent number ends up in scratch,
// and per-row lookups through global memory made hipcc hoist hundreds of loads into registers in the unrolled epilogue)
struct W3XMap { int nseg, cx8, k1, k2, k3, o0, o1, o2, o3, c0, c1, c2, c3; };
__device__ __forceinline__ W3XMap w3_xmap(const Wg3P* __restrict__ gp) {
  W3XMap m;
  m.nseg = gp->nsegx; m.cx8 = gp->cx8;
  m.k1 = gp->x_k8[1]; m.k2 = gp->x_k8[2]; m.k3 = gp->x_k8[3];
  m.o0 = gp->x_off[0]; m.o1 = gp->x_off[1]; m.o2 = gp->x_off[2]; m.o3 = gp->x_off[3];
  m.c0 = gp->segx[0].c; m.c1 = gp->segx[1].c; m.c2 = gp->segx[2].c; m.c3 = gp->segx[3].c;
  return m;
}
__device__ __forceinline__ int w3_xreal(const W3XMap& m, const int c8i) {
  int cs = c8i, off = m.o0, cnt = m.c0;
  if (m.nseg > 1 && c8i >= m.k1) { cs = c8i - m.k1; off = m.o1; cnt = m.c1; }
  if (m.nseg > 2 && c8i >= m.k2) { cs = c8i - m.k2; off = m.o2; cnt = m.c2; }
  if (m.nseg > 3 && c8i >= m.k3) { cs = c8i - m.k3; off = m.o3; cnt = m.c3; }
  return (c8i < m.cx8 && cs < cnt) ? off + cs : -1;
}

struct W3Lane {
  const char* base;  // this lane's source for (image 0, row 0, column 0): segment pointer + its channel group
  int sn, sh, sw;    // byte strides
  int pl;            // pixel of the lane inside a DMA instruction
  bool active, data;
};

// lane -> (pixel inside the instruction, 16-byte channel group) -> segment / channel of operand `o`; `is_x`: the operand is the
// (concatenated) conv input, else the output gradient; channels [c0, c0 + width) of the operand are staged
__device__ __forceinline__ W3Lane w3_lane(const Wg3P* __restrict__ gp, const W3Op& o, const bool is_x, const int c0, const int width, const int lane) {
  W3Lane L;
  L.pl = w3div(lane, o.d_gpp);
  const int grp = lane - L.pl * o.gpp;
  L.active = L.pl < o.ppi;
  const int c = c0 + grp * 8;
  const int nsegx = gp->nsegx;
  int si = 0;
#pragma unroll
  for (int k = 1; k < CGEN_MAX_SEG; ++k) si += (is_x && k < nsegx && c >= gp->x_k8[k]) ? 1 : 0;
  const View* sv = is_x ? &gp->segx[si] : &gp->g;
  const int k0 = is_x ? gp->x_k8[si] : 0;
  L.data = L.active && grp * 8 < width && c < o.c8;
  L.base = sv->p + (c - k0) * 2;
  L.sn = (int)sv->sn * 2; L.sh = (int)sv->sh * 2; L.sw = (int)sv->sw * 2;
  return L;
}

// the tile of operand `o` whose first pixel is (n, y0, x0) -> LDS at `dst`: instructions first, first + 4, ... (the general, slow form:
// a division and the image-border test per lane and instruction)
__device__ __forceinline__ void w3_issue_op(const W3Op& o, const W3Lane& L, const int n, const int y0, const int x0, const int H, const int W,
                                            const uint32_t dst, const int first) {
  const char* const zero = (const char*)g_w3zero;
  const int noff = n * L.sn;
  const int step = o.ppi * o.sp;
  for (int i = first; i < o.ninstr; i += 4) {
    const int lin = i * o.ppi + L.pl;
    const int ry = w3div(lin, o.d_row), rx = lin - ry * o.rowpx;
    const int gy = y0 + ry, gx = x0 + rx;
    const bool ok = L.data && lin < o.npx && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    const char* src = ok ? L.base + (noff + gy * L.sh + gx * L.sw) : zero;
    if (L.active) w3_dma16(src, __builtin_amdgcn_readfirstlane(dst + i * step));
  }
}

template <int MPW, int NSW>
__device__ __forceinline__ void wg3_body(const Wg3P* __restrict__ gp, const int sp_i, const int pwin, const int swin) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr)smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (fields of the problem record are read where they are used -- scalar loads through the uniform pointer; only what the tile
  //  loop needs stays live: the first version copied ~80 scalars up front and the loop ran on v_readlane'd SGPR spills)
  const int WK = gp->WK, WN = gp->WN;
  const int lwk = __builtin_ctz(WK), lwn = __builtin_ctz(WN);
  const int wk = wave & (WK - 1), wn = (wave >> lwk) & (WN - 1), wm = wave >> (lwk + lwn);
  const int H = gp->H, W = gp->W;
  const bool x_is_s = gp->x_is_s != 0;
  const int c0P = pwin * gp->pwin_c;
  const int widthP = min(gp->pwin_c, gp->P.c8 - c0P);
  const int th = gp->th, tw = gp->tw;
  const int nslot = gp->nslot, slot_bytes = gp->slot_bytes, pbytes = gp->P.bytes;
  const int ksteps = gp->ksteps;

  // ---- DMA lane constants.  Everything about a lane's share of a tile but the tile origin is a launch constant: the byte offset of
  // its pixel of the wave's k-th P / S instruction (poff / soff), that pixel's (row, column) inside the S tile for the image-border
  // test (syx, 6 bits each), and whether the slot carries data at all (vmask).  Per tile and instruction that leaves one 64-bit add
  // under an exec mask.  Instructions beyond the W3_PN / W3_SN kept in registers, and ragged P tiles, take the slow form.
  int poff[W3_PN], soff[W3_SN];
  uint32_t syx[(W3_SN + 1) / 2], vmask = 0;
  W3Lane LP, LS;
  int ninP, ninS, stepP, stepS, haloS, s_rows, s_rowpx;
  {
    const W3Op P = gp->P, S = gp->S;
    LP = w3_lane(gp, P, !x_is_s, c0P, widthP, lane);
    LS = w3_lane(gp, S, x_is_s, 0, S.c8, lane);
    ninP = P.ninstr; ninS = S.ninstr; stepP = P.ppi * P.sp; stepS = S.ppi * S.sp; haloS = S.halo; s_rows = S.rows; s_rowpx = S.rowpx;
#pragma unroll
    for (int k = 0; k < (W3_SN + 1) / 2; ++k) syx[k] = 0;
#pragma unroll
    for (int k = 0; k < W3_PN; ++k) {
      const int lin = (wave + 4 * k) * P.ppi + LP.pl;
      const int ry = w3div(lin, P.d_row), rx = lin - ry * P.rowpx;
      poff[k] = ry * LP.sh + rx * LP.sw;
      if (LP.data && lin < P.npx && wave + 4 * k < P.ninstr) vmask |= 1u << k;
    }
#pragma unroll
    for (int k = 0; k < W3_SN; ++k) {
      const int lin = (wave + 4 * k) * S.ppi + LS.pl;
      const int ry = w3div(lin, S.d_row), rx = lin - ry * S.rowpx;
      soff[k] = ry * LS.sh + rx * LS.sw;
      syx[k >> 1] |= (uint32_t)((ry & 63) << 6 | (rx & 63)) << (12 * (k & 1));
      if (LS.data && lin < S.npx && wave + 4 * k < S.ninstr) vmask |= 1u << (8 + k);
    }
  }
  const char* const zero = (const char*)g_w3zero;
  // tile walk: the NEXT tile to request, as (image, tile row, tile column), advanced incrementally
  const int tiles_x = gp->tiles_x, tiles_y = gp->tiles_y;
  const int t_begin = sp_i * gp->tps, t_end = min(gp->ntiles, t_begin + gp->tps);
  int q_n, q_ty, q_tx;
  {
    const int b1 = w3div(t_begin, gp->d_tx);
    q_tx = t_begin - b1 * tiles_x;
    q_n = w3div(b1, gp->d_ty);
    q_ty = b1 - q_n * tiles_y;
  }
  // The requests of a tile are SPREAD over the first K16-steps of the tile being consumed (positions 0..3): a burst of ~19 KiB right
  // after the barrier fills the CU's memory queue (a CU drains ~10 B/cycle, its share of HBM) and every further request stalls its
  // wave at issue -- ~2000 cycles per tile in which the wave computes nothing (stamps, LABNOTES 10.1); on the full chip the spread
  // form measured 8 % faster than the burst.  dma_begin: per-tile state; dma_pos<p>: the register-offset requests of position p.
  const char* d_bp = nullptr;
  const char* d_bs = nullptr;
  uint32_t d_sb = 0;
  int d_n = 0, d_y0 = 0, d_x0 = 0;
  bool d_inside = false, d_on = false, d_pfast = false;
  auto dma_begin = [&](const int slot) {
    d_n = q_n; d_y0 = q_ty * th; d_x0 = q_tx * tw;
    if (++q_tx == tiles_x) { q_tx = 0; if (++q_ty == tiles_y) { q_ty = 0; ++q_n; } }
    d_sb = lds0 + slot * slot_bytes;
    d_on = true;
    const int ys = d_y0 - haloS, xs = d_x0 - haloS;
    d_inside = ys >= 0 && xs >= 0 && ys + s_rows <= H && xs + s_rowpx <= W;  // (wave-uniform)
    d_bs = LS.base + (d_n * LS.sn + ys * LS.sh + xs * LS.sw);
    d_pfast = d_y0 + th <= H && d_x0 + tw <= W;  // P has no halo: a tile inside the image needs no per-lane test at all
    if (d_pfast) d_bp = LP.base + (d_n * LP.sn + d_y0 * LP.sh + d_x0 * LP.sw);
    else w3_issue_op(gp->P, LP, d_n, d_y0, d_x0, H, W, d_sb, wave);  // ragged P tile: the slow form, now
  };
  auto dma_p = [&](const int k) {
    if (wave + 4 * k < ninP) {
      const char* b = d_bp;
      asm volatile("" : "+v"(b));  // (opaque: or hipcc forms all nine addresses of a tile ahead of the K loop and keeps 18 registers live across it)
      if ((vmask >> k) & 1) w3_dma16(b + poff[k], __builtin_amdgcn_readfirstlane(d_sb + (wave + 4 * k) * stepP));
    }
  };
  auto dma_s = [&](const int k) {
    if (wave + 4 * k < ninS) {
      const uint32_t dst = __builtin_amdgcn_readfirstlane(d_sb + pbytes + (wave + 4 * k) * stepS);
      const char* d_bs_ = d_bs;
      asm volatile("" : "+v"(d_bs_));
      if (d_inside) {
        if ((vmask >> (8 + k)) & 1) w3_dma16(d_bs_ + soff[k], dst);
      } else {
        const uint32_t f = syx[k >> 1] >> (12 * (k & 1));
        const int gy = d_y0 - haloS + (int)((f >> 6) & 63), gx = d_x0 - haloS + (int)(f & 63);
        const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const char* src = ok ? d_bs_ + soff[k] : zero;
        if ((vmask >> (8 + k)) & 1) w3_dma16(src, dst);
      }
    }
  };
  auto dma_pos = [&](auto pc) {  // position p of the tile's first four K16-steps: requests k with k % 4 == p (P), (W3_PN + k) % 4 == p (S)
    constexpr int p = decltype(pc)::value;
    if (!d_on) return;
#pragma unroll
    for (int k = 0; k < W3_PN; ++k)
      if (k % 4 == p && d_pfast) dma_p(k);
#pragma unroll
    for (int k = 0; k < W3_SN; ++k)
      if ((W3_PN + k) % 4 == p) dma_s(k);
  };
  auto dma_rest = [&]() {  // the instructions beyond the register-offset ones (slow form), end of the tile's requests
    if (!d_on) return;
    if (d_pfast && wave + 4 * W3_PN < ninP) w3_issue_op(gp->P, LP, d_n, d_y0, d_x0, H, W, d_sb, wave + 4 * W3_PN);
    if (wave + 4 * W3_SN < ninS) w3_issue_op(gp->S, LS, d_n, d_y0 - haloS, d_x0 - haloS, H, W, d_sb + pbytes, wave + 4 * W3_SN);
    d_on = false;
  };
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
  typedef std::integral_constant<int, 2> P2;
  typedef std::integral_constant<int, 3> P3;
  auto issue_all = [&](const int slot) {  // a whole tile at once (prologue)
    dma_begin(slot);
    dma_pos(P0()); dma_pos(P1()); dma_pos(P2()); dma_pos(P3());
    dma_rest();
  };
  // DMA instructions this wave issues per tile (instruction i goes to wave i & 3)
  const int ipt = ((ninP + 3 - wave) >> 2) + ((ninS + 3 - wave) >> 2);

  // ---- fragment lane constants.  16-lane group g: K half g >> 1 (pixels 8 (g >> 1) .. + 8 of the K16-step), channel / column half
  // g & 1; lane t = 4 r + q of the group supplies (pixel r of the read, channels 4 q .. 4 q + 3) and receives column t's 4 pixels.
  const int spP = gp->P.sp, spS = gp->S.sp;
  const int fs0 = swin * (WN * NSW);  // first S fragment of this workgroup's column window
  uint32_t aP0, aS0[NSW], aS1[NSW];  // running LDS addresses of this lane's fragment reads (first / second read of a pair: P second = + 4 spP)
  uint32_t bmask = 0;  // S fragments of this wave that hold columns of the centre tap (bias gradient when grad_out is the S operand)
  {
    const int g = lane >> 4, t16 = lane & 15, r = t16 >> 2, q = t16 & 3;
    const int kh = g >> 1, ch16 = g & 1;
    const int klin0 = 8 * kh + r;  // pixel of the FIRST read inside the K16-step (linear over the tile's pixels); second: + 4
    aP0 = lds0 + klin0 * spP + (ch16 * 16 + 4 * q) * 2 + (wm * MPW) * 64 + wk * (16 * spP);
    const int cs8 = gp->S.c8, taps = gp->taps, ksz = gp->ks, ncols = taps * cs8, NS = gp->NS, rowpx = gp->S.rowpx;
    const int py0 = tw == 16 ? 0 : klin0 >> 3, px0 = tw == 16 ? klin0 : klin0 & 7;           // first read's pixel inside the K-step
    const int py1 = tw == 16 ? 0 : (klin0 + 4) >> 3, px1 = tw == 16 ? klin0 + 4 : (klin0 + 4) & 7;
#pragma unroll
    for (int j = 0; j < NSW; ++j) {
      const int fsj = fs0 + wn * NSW + j;
      const int fs = min(fsj, NS - 1);
      int n = fs * 32 + ch16 * 16 + 4 * q;
      if (n >= ncols) n = 0;  // padding column: any valid address (never stored)
      const int tap = n / cs8, c = n - tap * cs8;
      const int ey = tap / ksz, ex = tap - ksz * ey;
      const uint32_t sbase = lds0 + pbytes + wk * (gp->kst_rows * rowpx * spS);
      aS0[j] = sbase + ((py0 + ey) * rowpx + px0 + ex) * spS + c * 2;
      aS1[j] = sbase + ((py1 + ey) * rowpx + px1 + ex) * spS + c * 2;
      const int ctr = taps >> 1;
      if (fsj < NS && fsj * 32 < (ctr + 1) * cs8 && fsj * 32 + 32 > ctr * cs8) bmask |= 1u << j;
    }
  }
  const int kstepP = 16 * spP * WK, kstepS = gp->kst_rows * gp->S.rowpx * spS * WK;  // address advance per K16-step of this wave
  const int sp4 = 4 * spP;
  const int nkw = wk < ksteps ? (ksteps - wk + WK - 1) >> lwk : 0;  // K16-steps of a tile this wave takes = fragment loads per tile
  const int act = gp->act;
  const bool relu_p = act == CGEN_ACT_RELU && !x_is_s, relu_s = act == CGEN_ACT_RELU && x_is_s;
  const bool gelu = act == CGEN_ACT_GELU;
  float* const pb_ptr = gp->pb;
  const bool bias_p = pb_ptr != nullptr && x_is_s && wn == 0 && swin == 0;   // G is the P operand: row sums of my P fragments
  const bool bias_s = pb_ptr != nullptr && !x_is_s && wm == 0 && pwin == 0 && bmask != 0;  // G is the S operand: column sums of my centre-tap fragments
  // K-phase instance of this wave (see kphase below): 0 plain, 1 ReLU on P, 2 bias from S, 3 both | 4 ReLU on S, 5 bias from P, 6 both
  // -- chosen per PROBLEM, not per wave: the waves without bias duty add up sums nobody stores (4 dot products per fragment and
  // K16-step) rather than run a second copy of the loop next to their neighbours'
  const bool bias_any = pb_ptr != nullptr;
  const bool w3_uni = W3_UNIFORM_MODE;
  const bool bp_ = w3_uni ? bias_any : bias_p, bs_ = w3_uni ? bias_any : bias_s;
  const int kmode = __builtin_amdgcn_readfirstlane(x_is_s ? (relu_s ? (bp_ ? 6 : 4) : (bp_ ? 5 : 0)) : ((relu_p ? 1 : 0) | (bs_ ? 2 : 0)));

  f32x16 acc[MPW][NSW];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int j = 0; j < NSW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int NB = MPW > NSW ? MPW : NSW;
  float bsum[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bsum[i] = 0.f;

  // optional cycle stamps (CGEN_WG3_STAMPS=<device address of 256 u64>, tools/wg3_stamps.py): first lane of every wave of the
  // launch's first block, 4 stamps per tile for the first 14 tiles after 2 set-up stamps
#ifdef CGEN_WG3_STAMPS_BUILD  // (a debug build, -DCGEN_WG3_STAMPS_BUILD: the stamps cost registers and ~20 instructions per tile)
  unsigned long long* const stamp = (gp->stamps != nullptr && blockIdx.x == 0 && sp_i == 0 && pwin == 0 && swin == 0 && lane == 0) ? gp->stamps + wave * 64 : nullptr;
  int nst = 0;
#define W3_STAMP() do { if (stamp && nst < 62) stamp[nst++] = __builtin_readcyclecounter(); } while (0)
#else
#define W3_STAMP() do {} while (0)
#endif
  W3_STAMP();
  // ---- tile ring: tiles t .. t + nslot - 2 are in flight / resident while tile t is consumed; one barrier per tile
  const int ahead = nslot - 1;
  const int dbg = gp->dbg;  // ablation (CGEN_WG3_DBG, wrong results): 1 no DMA, 2 no fragment reads / MFMAs, 4 burst DMA
  for (int k = 0; k < ahead; ++k)
    if (t_begin + k < t_end && !(dbg & 1)) issue_all(k);
  int slot = 0;
  W3_STAMP();
  // The tile loop is instantiated once per (ReLU on P / on S / none) x (bias sums from P / from S / none) and entered through ONE
  // switch: with the four flag tests inside every K16-step (the form up to round 5b) half of the kernel's time outside the DMA went
  // into those branches -- two waves per SIMD do not hide a taken branch's fetch bubble (tools/bench_wgrad3.py batch with the
  // W3_ABL_* builds: 192^2 32->8 without DMA 132 us, of which 54 us the flag tests, 13 us the fragment reads + MFMAs; LABNOTES 10.5).
  // (The switch sits outside the loop: inside it, seven K phases joining per tile cost the 3x3 block 300 spilled registers.)
  auto tile_loop = [&](auto c_rp, auto c_rs, auto c_bp, auto c_bs) {
  constexpr bool RP = decltype(c_rp)::value, RS = decltype(c_rs)::value, BP = decltype(c_bp)::value, BS = decltype(c_bs)::value;
  for (int t = t_begin; t < t_end; ++t) {
    {  // tile t has landed (this wave's requests; the barrier makes it everybody's)
      const int later = min(t_end - 1 - t, ahead - 1);  // tiles after t already requested
      w3_vmwait16(later * ipt);
      W3_STAMP();
      W3_BARRIER();  // ... and every wave is done with the slot the next request overwrites (tile t - 1's)
      W3_STAMP();
    }
    if (t + ahead < t_end && !(dbg & 1)) {
      int s2 = slot + ahead;
      if (s2 >= nslot) s2 -= nslot;
      dma_begin(s2);
    }
    W3_STAMP();
    const uint32_t bufP = lds0 + slot * slot_bytes, bufS = bufP + pbytes;
    (void)bufS;
#ifndef W3_ABL_NOGELU
    if constexpr (!RP && !RS) if (gelu) {  // in-place activation of the staged X tile (ReLU is applied to the fragments instead)
      const uint32_t xb = x_is_s ? bufS : bufP;
      const int nb16 = (x_is_s ? gp->S.bytes : pbytes) >> 4;
      for (int s = tid; s < nb16; s += 256) {
        uint4* ptr = (uint4*)(smem + (xb - lds0) + s * 16);
        *ptr = w3_gelu8(*ptr);
      }
      W3_BARRIER();
    }
#endif
    // K16-steps of the tile.  Lane addresses advance by increments (two adds per operand base, the P fragments hang off one base by
    // immediate offsets); the fragment reads of step k + 1 are issued before the MFMAs of step k (two register sets) where the
    // fragment block leaves room; the first four steps carry the DMA requests of the tile two ahead.
    auto load = [&](h16x8 (&pf)[MPW], h16x8 (&sf)[NSW]) {
#pragma unroll
      for (int i = 0; i < MPW; ++i) pf[i] = w3_tr(aP0 + i * 64, aP0 + sp4 + i * 64);
#pragma unroll
      for (int j = 0; j < NSW; ++j) sf[j] = w3_tr(aS0[j], aS1[j]);
      aP0 += kstepP;
#pragma unroll
      for (int j = 0; j < NSW; ++j) { aS0[j] += kstepS; aS1[j] += kstepS; }
    };
    {
      auto mac = [&](h16x8 (&pf)[MPW], h16x8 (&sf)[NSW]) {
#ifndef W3_ABL_NOACT
        if constexpr (BP) {
#pragma unroll
          for (int i = 0; i < MPW; ++i) bsum[i] = w3_sum8(pf[i], bsum[i]);
        }
        if constexpr (BS) {  // (every S fragment of the wave: the epilogue takes the centre-tap ones, bmask)
#pragma unroll
          for (int j = 0; j < NSW; ++j) bsum[j] = w3_sum8(sf[j], bsum[j]);
        }
        if constexpr (RP) {
#pragma unroll
          for (int i = 0; i < MPW; ++i) pf[i] = w3_relu8(pf[i]);
        }
        if constexpr (RS) {
#pragma unroll
          for (int j = 0; j < NSW; ++j) sf[j] = w3_relu8(sf[j]);
        }
#endif
#pragma unroll
        for (int i = 0; i < MPW; ++i)
#pragma unroll
          for (int j = 0; j < NSW; ++j) acc[i][j] = w3_mfma(pf[i], sf[j], acc[i][j]);
      };
      constexpr bool PIPE = MPW * NSW <= 6;  // (the 8- and 9-fragment blocks have no registers for a second fragment set)
      // (loop shape matters: ONE back edge, no exit in the middle -- with a break between the two halves hipcc kept two copies of the
      //  accumulators and moved 16 registers per MFMA between them)
      const int nk = (dbg & 2) ? 0 : nkw;
      if constexpr (PIPE) {
        h16x8 pfA[MPW], sfA[NSW], pfB[MPW], sfB[NSW];
        if (nk > 0) load(pfA, sfA);
        int i = 0;
        for (; i + 2 <= nk; i += 2) {
          load(pfB, sfB);
#ifndef W3_ABL_NOPOS
          if (i == 0) dma_pos(P0()); else if (i == 2) dma_pos(P2());
#endif
          mac(pfA, sfA);
          if (i + 2 < nk) load(pfA, sfA);
#ifndef W3_ABL_NOPOS
          if (i == 0) dma_pos(P1()); else if (i == 2) dma_pos(P3());
#endif
          mac(pfB, sfB);
        }
        if (i < nk) mac(pfA, sfA);
      } else {
        for (int i = 0; i < nk; ++i) {
          h16x8 pf[MPW], sf[NSW];
          load(pf, sf);
          if (i == 0) { dma_pos(P0()); dma_pos(P1()); } else if (i == 1) { dma_pos(P2()); dma_pos(P3()); }
          mac(pf, sf);
        }
      }
      {  // request groups the K16-steps above did not reach (short tiles / K-split waves), then the slow-form remainder
        constexpr int per = PIPE ? 2 : 1;  // steps per pair of positions
#ifdef W3_ABL_NOPOS
        const int done = PIPE ? 0 : (nk / per) * 2;
#else
        const int done = (nk / per) * 2;   // positions issued inside the loop (pairs of them)
#endif
        if (done < 2) { dma_pos(P0()); dma_pos(P1()); }
        if (done < 4) { dma_pos(P2()); dma_pos(P3()); }
        dma_rest();
      }
    }
    W3_STAMP();
    {  // the running addresses advanced by this wave's K16-steps: rewind them onto the next slot of the ring (scalar deltas)
      int dslot = slot_bytes;
      if (++slot == nslot) { slot = 0; dslot = -(nslot - 1) * slot_bytes; }
      if (!(dbg & 2)) {
        aP0 += dslot - nkw * kstepP;
#pragma unroll
        for (int j = 0; j < NSW; ++j) { aS0[j] += dslot - nkw * kstepS; aS1[j] += dslot - nkw * kstepS; }
      } else {
        aP0 += dslot;
#pragma unroll
        for (int j = 0; j < NSW; ++j) { aS0[j] += dslot; aS1[j] += dslot; }
      }
    }
  }
  };
  {
    typedef std::integral_constant<bool, false> F_;
    typedef std::integral_constant<bool, true> T_;
    switch (kmode) {  // (uniform per wave; x_is_s decides which operand is X: ReLU and the bias sums sit on opposite operands)
      case 0: tile_loop(F_(), F_(), F_(), F_()); break;
      case 1: tile_loop(T_(), F_(), F_(), F_()); break;   // ReLU on P (= X)
      case 2: tile_loop(F_(), F_(), F_(), T_()); break;   // bias sums from S (= grad_out)
      case 3: tile_loop(T_(), F_(), F_(), T_()); break;
      case 4: tile_loop(F_(), T_(), F_(), F_()); break;   // ReLU on S (= X)
      case 5: tile_loop(F_(), F_(), T_(), F_()); break;   // bias sums from P (= grad_out)
      default: tile_loop(F_(), T_(), T_(), F_()); break;
    }
  }
#undef W3_STAMP
  // (the caller's barrier separates this problem's last LDS reads from the next problem's first DMA)

  // K-split waves (WK > 1: small slabs, every wave takes every WK-th K16-step of a tile): summed here through LDS in the fixed order
  // wk = 0, 1, 2, 3, one source wave at a time, so the workgroup writes ONE slab (the first version wrote WK of them: the 192^2
  // layers then had 768 partial slabs of 9 KB each and cgen_wgrad_reduce walked them serially, LABNOTES 10.2)
  if (WK > 1) {
    W3_BARRIER();  // every wave is done with the tile ring
    float* const red = (float*)smem + (size_t)(wave >> lwk) * ((MPW * NSW * 16 + NB) * 64) + lane;
    for (int src = 1; src < WK; ++src) {
      if (wk == src) {
#pragma unroll
        for (int i = 0; i < MPW; ++i)
#pragma unroll
          for (int j = 0; j < NSW; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[((i * NSW + j) * 16 + e) * 64] = acc[i][j][e];
            asm volatile("" ::: "memory");
          }
#pragma unroll
        for (int i = 0; i < NB; ++i) red[(MPW * NSW * 16 + i) * 64] = bsum[i];
      }
      W3_BARRIER();
      if (wk == 0) {
#pragma unroll
        for (int i = 0; i < MPW; ++i)
#pragma unroll
          for (int j = 0; j < NSW; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] += red[((i * NSW + j) * 16 + e) * 64];
            asm volatile("" ::: "memory");  // (one fragment at a time: hoisting all the loads needs a second accumulator set)
          }
#pragma unroll
        for (int i = 0; i < NB; ++i) bsum[i] += red[(MPW * NSW * 16 + i) * 64];
      }
      W3_BARRIER();
    }
    if (wk != 0) return;
  }

  // ---- write the partial slab straight from the accumulators.  Fragment (i, j): lane l holds rows (e & 3) + 8 (e >> 2) + 4 (l >> 5),
  // column l & 31: consecutive lanes -> consecutive (tap, channel) of S -> 128-byte runs in both layouts.
  int el = lane;  // (opaque from here on: everything the epilogue derives from the lane index is computed AFTER the tile loop -- hipcc hoisted
  asm volatile("" : "+v"(el));  //  those ~50 loop-invariant values above the loop and spilled them around it)
  const int taps = gp->taps, co_n = gp->co, ci_n = gp->ci_total, NS = gp->NS;
  const W3XMap xm = w3_xmap(gp);
  const int ncol = x_is_s ? ci_n : co_n;  // innermost extent of the partial layout
  const int nrow = x_is_s ? co_n : ci_n;
  float* const pw = gp->pw + (size_t)sp_i * ((size_t)co_n * taps * ci_n);
  {
    const int cs8 = gp->S.c8, ncols = taps * cs8;
    int coff[NSW];  // tap * ncol + real column channel, -1: nothing to store
#pragma unroll
    for (int j = 0; j < NSW; ++j) {
      const int fs = fs0 + wn * NSW + j;
      const int n = fs * 32 + (el & 31);
      coff[j] = -1;
      if (fs < NS && n < ncols) {
        const int tap = n / cs8, c8i = n - tap * cs8;
        const int cc = x_is_s ? w3_xreal(xm, c8i) : (c8i < co_n ? c8i : -1);
        if (cc >= 0) coff[j] = tap * ncol + cc;
      }
    }
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
      const int cwin = (wm * MPW + i) * 32;  // first window-relative channel of the fragment's rows
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) {
          const int row = e1 + 8 * e4 + 4 * (el >> 5);
          const int pc = c0P + cwin + row;
          int rr = -1;
          if (cwin + row < widthP) rr = x_is_s ? (pc < co_n ? pc : -1) : w3_xreal(xm, pc);
          if (rr >= 0 && rr < nrow) {
            float* const rowp = pw + (size_t)rr * taps * ncol;
#pragma unroll
            for (int j = 0; j < NSW; ++j)
              if (coff[j] >= 0) rowp[coff[j]] = acc[i][j][e4 * 4 + e1];
          }
        }
      }
    }
  }
  if (bias_p || bias_s) {
    float* const pb = pb_ptr + (size_t)sp_i * co_n;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);  // the two K halves of the fragment
      if (el < 32) {
        if (bias_p && i < MPW) {
          const int cwin = (wm * MPW + i) * 32 + el;
          const int co = c0P + cwin;
          if (cwin < widthP && co < co_n) pb[co] = tot;
        }
        if (bias_s && i < NSW && ((bmask >> i) & 1)) {
          const int fs = fs0 + wn * NSW + i;
          const int n = fs * 32 + el;
          const int cs8 = gp->S.c8;
          const int tap = n / cs8, c = n - tap * cs8;
          if (fs < NS && tap == (taps >> 1) && c < co_n) pb[c] = tot;
        }
      }
    }
  }
}

#define W3_DISPATCH(p, a, b, c)                                  \
  switch ((p)->variant) {                                         \
    case 8 * 1 + 1: wg3_body<1, 1>(p, a, b, c); break;            \
    case 8 * 1 + 2: wg3_body<1, 2>(p, a, b, c); break;            \
    case 8 * 1 + 3: wg3_body<1, 3>(p, a, b, c); break;            \
    case 8 * 1 + 4: wg3_body<1, 4>(p, a, b, c); break;            \
    case 8 * 2 + 1: wg3_body<2, 1>(p, a, b, c); break;            \
    case 8 * 2 + 2: wg3_body<2, 2>(p, a, b, c); break;            \
    case 8 * 2 + 3: wg3_body<2, 3>(p, a, b, c); break;            \
    case 8 * 2 + 4: wg3_body<2, 4>(p, a, b, c); break;            \
    case 8 * 3 + 1: wg3_body<3, 1>(p, a, b, c); break;            \
    case 8 * 3 + 2: wg3_body<3, 2>(p, a, b, c); break;            \
    case 8 * 3 + 3: wg3_body<3, 3>(p, a, b, c); break;            \
    case 8 * 4 + 1: wg3_body<4, 1>(p, a, b, c); break;            \
    default: wg3_body<4, 2>(p, a, b, c); break;                   \
  }

// ONE kernel for both launch forms (the 13 fragment blocks x 7 tile-loop instances take hipcc ~4 minutes; two kernels took ten):
//   packed: all problems of a flush in one launch (as wgrad_tile_mega_kernel): a resident set of workgroups walks the block list
//     (longest first), blocks[b] = {problem, split, P window, S window}; problems are read through a uniform pointer;
//   single (single_slot >= 0): one problem, one workgroup per (S window, P window, split).  Its record reaches the kernel through
//     device memory as well -- wg3_stash_kernel copies it from ITS kernel argument into g_w3single[slot] on the same stream just
//     before -- because a record pointer that is either global memory or the kernarg segment becomes a flat pointer, every field
//     read a vector load, and the body spills a thousand registers (measured).  Both launches are plain kernel launches with
//     by-value arguments: capturable, and a replay re-stashes the same record.
#define W3_SINGLE_SLOTS 512
__device__ Wg3P g_w3single[W3_SINGLE_SLOTS];

__global__ __launch_bounds__(64) void wg3_stash_kernel(const Wg3P p_, const int slot) {
  (void)p_;
  const uint32_t* src = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t* dst = (uint32_t*)(g_w3single + slot);
  for (int i = threadIdx.x; i < (int)(sizeof(Wg3P) / 4); i += 64) dst[i] = src[i];
}

__global__ __launch_bounds__(256, 2) void wg3_mega_kernel(const Wg3P* __restrict__ probs, const int4* __restrict__ blocks, const int nblocks, const int single_slot) {
  const Wg3P* __restrict__ const base = single_slot >= 0 ? g_w3single + single_slot : probs;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    int4 bi;
    if (single_slot >= 0) {
      const int ns = base->nsplit, npw = base->n_pwin;
      const int w = b / ns;
      bi.x = 0; bi.y = b - w * ns; bi.w = w / npw; bi.z = w - bi.w * npw;
    } else {
      bi = blocks[b];
    }
    const Wg3P* __restrict__ gp = base + __builtin_amdgcn_readfirstlane(bi.x);
    const int sp_i = __builtin_amdgcn_readfirstlane(bi.y), pwin = __builtin_amdgcn_readfirstlane(bi.z), swin = __builtin_amdgcn_readfirstlane(bi.w);
    // optional block log (CGEN_WG3_BLOCKLOG=<device address of u64[1 + 4 n]>, tools/wg3_blocklog.py): [0] = entries so far, then per
    // block {problem << 32 | H << 16 | ks << 8 | variant, ci << 48 | co << 32 | split << 16 | P window << 8 | S window, start, end}
    // on the 100 MHz wall clock
    unsigned long long* const blog = gp->blocklog;
    unsigned long long t0 = 0;
    if (blog != nullptr) t0 = wall_clock64();
    W3_DISPATCH(gp, sp_i, pwin, swin)
    __syncthreads();
    if (blog != nullptr && threadIdx.x == 0) {
      const unsigned long long t1 = wall_clock64();
      const unsigned long long e = atomicAdd(blog, 1ull);
      if (e < gp->blocklog_cap) {
        unsigned long long* r = blog + 1 + 4 * e;
        r[0] = ((unsigned long long)(unsigned)bi.x << 32) | ((unsigned)gp->H << 16) | ((unsigned)gp->ks << 8) | (unsigned)gp->variant;
        r[1] = ((unsigned long long)(unsigned)gp->ci_total << 48) | ((unsigned long long)(unsigned)gp->co << 32) | ((unsigned)sp_i << 16) | ((unsigned)pwin << 8) | (unsigned)swin;
        r[2] = t0; r[3] = t1;
      }
    }
  }
}

// ============================================================================= host: plan
static inline int w3_pad(int v, int m) { return (v + m - 1) / m * m; }
static int w3_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

static void w3_mk_op(W3Op& o, int c8_staged, int sp, int rows, int rowpx, int halo, int c8_total) {
  o.c8 = c8_total;
  o.sp = sp; o.gpp = sp / 16; o.ppi = 64 / o.gpp;
  o.rows = rows; o.rowpx = rowpx; o.npx = rows * rowpx; o.halo = halo;
  o.ninstr = (o.npx + o.ppi - 1) / o.ppi;
  o.bytes = o.ninstr * o.ppi * sp;
  o.d_row = mk_w3div(rowpx); o.d_gpp = mk_w3div(o.gpp);
  (void)c8_staged;
}

bool wg3_plan(const cgen_wgrad_args* a, Wg3Plan& g) {
  static const int on = w3_env("CGEN_WG3", 1);
  if (!on || a->dtype != CGEN_F16 || !(a->ks == 1 || a->ks == 3 || a->ks == 7)) return false;
  if (a->nseg < 1 || a->nseg > CGEN_MAX_SEG || !a->gout.p || a->gout.c <= 0) return false;
  static const int min_hw = w3_env("CGEN_WG3_MINHW", 1);
  if (a->h < min_hw || a->w < min_hw) return false;
  if (!dma_clean(a->gout, 2)) return false;
  for (int s = 0; s < a->nseg; ++s)
    if (!dma_clean(a->seg[s], 2)) return false;
  auto fits = [&](const cgen_view& v) {  // 32-bit byte offsets, tile overhang included
    const int64_t ext = (int64_t)a->n * v.sn + (int64_t)(a->h + 34) * v.sh + (int64_t)(a->w + 34) * v.sw + v.c + 8;
    return ext * 2 < ((int64_t)1 << 31) && v.sn >= 0 && v.sh >= 0 && v.sw >= 0;
  };
  if (!fits(a->gout)) return false;
  for (int s = 0; s < a->nseg; ++s)
    if (!fits(a->seg[s])) return false;

  Wg3P& q = g.q;
  memset(&q, 0, sizeof(q));
  q.N = a->n; q.H = a->h; q.W = a->w; q.ks = a->ks; q.taps = a->ks * a->ks; q.act = a->act;
  q.co = a->gout.c;
  int k8 = 0, off = 0;
  q.nsegx = a->nseg;
  for (int s = 0; s < a->nseg; ++s) {
    q.segx[s] = mk(a->seg[s]); q.x_k8[s] = k8; q.x_off[s] = off;
    k8 += w3_pad(a->seg[s].c, 8); off += a->seg[s].c;
  }
  for (int s = a->nseg; s <= CGEN_MAX_SEG; ++s) q.x_k8[s] = k8;
  q.ci_total = off;
  q.cx8 = k8;
  q.g = mk(a->gout);
  const int cx8 = k8, cg8 = w3_pad(q.co, 8);
  if ((int64_t)q.co * q.taps * q.ci_total >= ((int64_t)1 << 29)) return false;
  // the narrower operand is the shifted, tap-packed one (ties: X, the standard partial layout)
  q.x_is_s = cx8 <= cg8 ? 1 : 0;
  const int cs8 = q.x_is_s ? cx8 : cg8, cp8 = q.x_is_s ? cg8 : cx8;
  q.layout = q.x_is_s ? 0 : 1;
  const int NS = (q.taps * cs8 + 31) / 32;
  const int MPT = (cp8 + 31) / 32;
  q.NS = NS;

  // ---- wave grid and per-wave fragment block: fewest P windows (each re-reads S), then fewest MFMA slots, then fewest LDS reads
  static const int grids[6][3] = {{1, 4, 1}, {2, 2, 1}, {4, 1, 1}, {1, 2, 2}, {2, 1, 2}, {1, 1, 4}};
  long best_key = -1;
  int bWM = 1, bWN = 4, bWK = 1, bMPW = 1, bNSW = 1;
  for (int gi = 0; gi < 6; ++gi) {
    const int WM = grids[gi][0], WN = grids[gi][1], WK = grids[gi][2];
    for (int NSW = 1; NSW <= 4; ++NSW) {
      if (NSW > 1 && WN * (NSW - 1) >= NS) break;  // (a narrower block already covers every S fragment)
      const int mpw_max = NSW == 4 ? 2 : (NSW == 3 ? 3 : 4);
      const int nswin = (NS + WN * NSW - 1) / (WN * NSW);
      const int cap = WM * mpw_max;
      const int nwin = (MPT + cap - 1) / cap;
      const int mp_win = (MPT + nwin - 1) / nwin;  // balanced windows
      const int MPW = (mp_win + WM - 1) / WM;
      if (WK > 1 && (4 / WK) * (MPW * NSW * 16 + std::max(MPW, NSW)) * 256 > 48 * 1024) continue;  // (LDS of the K-split reduction)
      // bytes a pixel position costs: every S window re-reads the P window, every P window re-reads the S tile
      const long bytes = (long)nswin * cp8 + (long)nwin * nswin * cs8 * 2;
      const long mfma = (long)nwin * nswin * MPW * NSW * 12 / WK;  // MFMA slots per K16-step and wave (x 12: exact for WK in {1, 2, 4})
      const long reads = (long)nwin * nswin * (MPW + NSW) * 12 / WK;
      const long key = (bytes << 36) + (mfma << 18) + reads;
      if (best_key < 0 || key < best_key) { best_key = key; bWM = WM; bWN = WN; bWK = WK; bMPW = MPW; bNSW = NSW; }
    }
  }
  if (best_key < 0) return false;
  q.WM = bWM; q.WN = bWN; q.WK = bWK;
  q.MP = bMPW * bWM;
  q.pwin_c = q.MP * 32;
  q.n_pwin = (cp8 + q.pwin_c - 1) / q.pwin_c;
  q.n_swin = (NS + bWN * bNSW - 1) / (bWN * bNSW);
  q.variant = bMPW * 8 + bNSW;

  // ---- LDS pixel strides: P fragments read 64 contiguous bytes of 4 pixels -> stride = 64 x odd is conflict free; S (narrow,
  // tap-packed) 16 x odd
  const int wP = std::min(q.pwin_c, cp8);
  int spP = w3_pad(wP * 2, 64);
  if (((spP / 64) & 1) == 0) spP += 64;
  int spS = w3_pad(cs8 * 2, 16);
  if (((spS / 16) & 1) == 0) spS += 16;
  static const int sp_plain = w3_env("CGEN_WG3_PLAINSTRIDE", 1);  // no padding: tools/probe_tr16_bw.hip measured the transpose reads at the same rate for every pixel stride but 256 / 272 B, and the LDS is 7 % busy (PMC)
  if (sp_plain) {
    spP = w3_pad(wP * 2, 16); spS = w3_pad(cs8 * 2, 16);
    if (spP == 256 || spP == 272) spP = 288;
  }
  if (spP > 1024 || spS > 1024) return false;

  // ---- tile: tw = 16 (8 for narrow images), th from the LDS budget; ring of nslot tiles
  static const int lds_cap = w3_env("CGEN_WG3_LDS", 72) * 1024;
  static const int want_slots = w3_env("CGEN_WG3_SLOTS", 3);
  static const int force_tpx = w3_env("CGEN_WG3_TPX", 0);
  const int halo = a->ks / 2;
  // 16 pixels of one row per K16-step, or 8 + 8 of two rows where that wastes fewer columns (24-pixel rows: 16 + 16 would carry a
  // quarter of zeros through the DMA, the LDS and the MFMAs -- the 24^2 layers are a quarter of the final flush's time, LABNOTES 10.5)
  {
    const int pad16 = w3_pad(a->w, 16), pad8 = w3_pad(a->w, 8);
    // (a tie -- 12-pixel rows -- goes to 8 as well: every 16-wide tile of such an image is ragged and takes the slow request form)
    q.tw = (a->w < 12 || ((pad8 < pad16 || (pad8 == pad16 && a->w % 16 != 0 && w3_env("CGEN_WG3_TW8", 2) > 1)) && w3_env("CGEN_WG3_TW8", 2))) ? 8 : 16;
  }
  q.kst_rows = 16 / q.tw;
  const int hpad = w3_pad(a->h, q.kst_rows);
  // the LARGEST tile (pixels) whose ring fits the LDS budget and whose requests fit the register-offset tables: the per-tile costs
  // (barrier, counted wait, request set-up) are fixed, so the narrow 192^2 / 96^2 layers -- 60 % of the bytes of a ukbb192 step --
  // want 256-pixel tiles where the 96-channel layers of 48^2 take 64 (same ~18 KB per ring slot).  Pass 0: three slots and only
  // register-offset requests; pass 1: two slots; pass 2: anything that fits.
  // Candidates also with 3 / 6 / 12 / 24 rows: a 12- or 24-row image cut into 8-row tiles carries a third of padding rows.  Among the
  // candidates a pass admits, the lowest estimated cost per valid pixel wins: (padded rows / rows) x (K16-steps + 2.5) / K16-steps,
  // the 2.5 being the per-tile fixed cost in K16-steps (tools/bench_wgrad3.py batch with CGEN_WG3_TPX: LABNOTES 10.5).
  static const int tpx_order[9] = {512, 384, 256, 192, 128, 96, 64, 48, 32};
  int pick = -1, pick_slots = 0;
  for (int pass = 0; pass < 3 && pick < 0; ++pass) {
    double best = 1e30;
    for (int k = 0; k < 9; ++k) {
      const int tpx = force_tpx ? force_tpx : tpx_order[k];
      if (tpx % q.tw) continue;
      int th = tpx / q.tw;
      if (th < q.kst_rows || th % q.kst_rows || th + 2 * halo > 60) continue;
      if (th > hpad && k != 8 && !force_tpx) continue;  // (a tile taller than the image: only as the last resort)
      W3Op P, S;
      w3_mk_op(P, wP, spP, th, q.tw, 0, cp8);
      w3_mk_op(S, cs8, spS, th + 2 * halo, q.tw + 2 * halo, halo, cs8);
      const int slot = w3_pad(P.bytes + S.bytes, 16);
      const int ns = std::min(want_slots, lds_cap / slot);
      const bool fast = ceil_div(P.ninstr, 4) <= W3_PN && ceil_div(S.ninstr, 4) <= W3_SN;
      const bool counted = (ceil_div(P.ninstr, 4) + ceil_div(S.ninstr, 4)) * std::max(0, ns - 2) <= 15;
      const bool ok = pass == 0 ? (ns >= want_slots && fast && counted) : (pass == 1 ? (ns >= 2 && fast && counted) : ns >= 2);
      if (ok) {
        const int kst = th * q.tw / 16;
        const double cost = (double)(ceil_div(a->h, th) * th) / a->h * (kst + 2.5) / kst;
        if (cost < best - 1e-9) { best = cost; pick = tpx; pick_slots = ns; }
      }
      if (force_tpx) break;
    }
  }
  if (pick < 0) return false;
  q.th = pick / q.tw;
  w3_mk_op(q.P, wP, spP, q.th, q.tw, 0, cp8);
  w3_mk_op(q.S, cs8, spS, q.th + 2 * halo, q.tw + 2 * halo, halo, cs8);
  q.slot_bytes = w3_pad(q.P.bytes + q.S.bytes, 16);
  q.nslot = pick_slots;
  q.ksteps = q.th * q.tw / 16;
  q.tiles_x = (a->w + q.tw - 1) / q.tw; q.tiles_y = (a->h + q.th - 1) / q.th;
  q.ntiles = a->n * q.tiles_x * q.tiles_y;
  q.d_tx = mk_w3div(q.tiles_x); q.d_ty = mk_w3div(q.tiles_y);
  if (ceil_div(q.P.ninstr, 4) + ceil_div(q.S.ninstr, 4) > 48 / std::max(1, q.nslot - 2)) {
    // (more requests per wave in flight than the counted-wait table holds: w3_vmwait would fall back to vmcnt(0) -- correct, slower)
  }

  // ---- split-K: a workgroup's partial slab (written + re-read by the reduce) must stay a few % of the bytes it streams
  const long dbytes = 4L * q.co * q.taps * q.ci_total;
  const long tile_bytes = 2L * ((long)q.n_swin * q.th * q.tw * (long)cp8 + (long)q.n_pwin * q.n_swin * q.S.npx * cs8);
  static const int part_pct = w3_env("CGEN_WG3_PARTIAL_PCT", 30);
  static const int max_kb = w3_env("CGEN_WG3_BLOCK_KB", 1536);  // HBM bytes per workgroup: ~60 us at a CU's share of the bandwidth; short blocks balance the packed launch
  static const int min_tps = w3_env("CGEN_WG3_MINTPS", 4);
  const long blk_tile_bytes = 2L * (q.th * q.tw * (long)wP + (long)q.S.npx * cs8);  // one tile of one workgroup
  long tps = std::max<long>(1, (long)max_kb * 1024 / blk_tile_bytes);
  const long tps_lo = (2 * dbytes * 100 + part_pct * tile_bytes - 1) / (part_pct * tile_bytes);
  tps = std::max<long>(tps, tps_lo);
  tps = std::max<long>(tps, min_tps);
  tps = std::min<long>(tps, q.ntiles);
  q.tps = (int)tps;
  q.nsplit = ceil_div(q.ntiles, q.tps);
  g.nsplit_total = q.nsplit;
  g.nblocks = q.nsplit * q.n_pwin * q.n_swin;
  g.lds = (size_t)q.nslot * q.slot_bytes;
  if (q.WK > 1) g.lds = std::max(g.lds, (size_t)(4 / q.WK) * ((q.variant / 8) * (q.variant % 8) * 16 + std::max(q.variant / 8, q.variant % 8)) * 256);
  g.block_bytes = (long)q.tps * (2L * (q.th * q.tw * (long)wP + (long)q.S.npx * cs8));
  q.pw = a->partial_w; q.pb = a->partial_b;
  static unsigned long long* const stamps = [] { const char* e = getenv("CGEN_WG3_STAMPS"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }();
  q.stamps = stamps;
  static unsigned long long* const blog = [] { const char* e = getenv("CGEN_WG3_BLOCKLOG"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }();
  q.blocklog = blog;
  q.blocklog_cap = (unsigned long long)w3_env("CGEN_WG3_BLOCKLOG_CAP", 0);
  static const int dbg = w3_env("CGEN_WG3_DBG", 0);
  q.dbg = dbg;
  if (getenv("CGEN_WG3_PLAN_DEBUG"))
    fprintf(stderr, "wg3 plan: %dx%dx%d ks%d cx8 %3d co %3d act %d | S=%s cs8 %3d NS %2d | cp8 %3d pwin %3d x%d swin x%d | grid %dx%dx%d frag %dx%d | tile %2dx%2d sp %3d/%3d "
            "instr %2d+%2d slot %5d x%d | tiles %5d tps %4d nsplit %3d (x%d) blocks %4d\n",
            a->n, a->h, a->w, a->ks, cx8, q.co, a->act, q.x_is_s ? "X" : "G", cs8, NS, cp8, q.pwin_c, q.n_pwin, q.n_swin, q.WM, q.WN, q.WK, q.variant / 8, q.variant % 8,
            q.th, q.tw, spP, spS, q.P.ninstr, q.S.ninstr, q.slot_bytes, q.nslot, q.ntiles, q.tps, q.nsplit, q.WK, g.nblocks);
  return true;
}

static void w3_attr_once() {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)wg3_mega_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
}

void wg3_launch_single(const Wg3Plan& g, hipStream_t st) {
  w3_attr_once();
  // The record travels through a rotating stash slot.  A CAPTURED launch bakes its slot into the graph and replays it for ever, so
  // captured launches draw from the upper half of the slots and eager ones from the lower half: an eager launch on another stream can
  // never overwrite a record between a replaying graph's stash and its kernel.  Graphs keep distinct slots among
  // themselves until W3_SINGLE_SLOTS / 2 captured single launches are alive (the train step packs its weight gradients into one
  // batched launch: single launches are the stand-alone C-ABI calls); past that the wrap is reported once.
  static std::atomic<unsigned> next_eager{0}, next_cap{0};
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  constexpr unsigned HALF = W3_SINGLE_SLOTS / 2;
  int slot;
  if (cs == hipStreamCaptureStatusActive) {
    const unsigned k = next_cap.fetch_add(1);
    if (k == HALF) fprintf(stderr, "cgen: more than %u captured single weight-gradient launches: stash slots of captured graphs are being reused\n", HALF);
    slot = (int)(HALF + k % HALF);
  } else {
    slot = (int)(next_eager.fetch_add(1) % HALF);
  }
  hipLaunchKernelGGL(wg3_stash_kernel, dim3(1), dim3(64), 0, st, g.q, slot);
  hipLaunchKernelGGL(wg3_mega_kernel, dim3(g.nblocks), dim3(256), g.lds, st, (const Wg3P*)nullptr, (const int4*)nullptr, g.nblocks, slot);
}

void wg3_launch_mega(const Wg3P* probs_dev, const int4* blocks_dev, int nblocks, int grid, size_t lds, hipStream_t st) {
  w3_attr_once();
  hipLaunchKernelGGL(wg3_mega_kernel, dim3(grid), dim3(256), lds, st, probs_dev, blocks_dev, nblocks, -1);
}

}  // namespace cgen

#ifdef W3_VARIANT_KERNELS  // register audit: one kernel per fragment block (tools/b3_regs.sh style: hipcc -DW3_VARIANT_KERNELS -Rpass-analysis=kernel-resource-usage)
namespace cgen {
#define W3_VK(M, N) __global__ __launch_bounds__(256, 2) void wg3_variant_##M##x##N(const Wg3P* __restrict__ probs, const int4* __restrict__ blocks, const int nblocks) { \
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) { const int4 bi = blocks[b]; wg3_body<M, N>(probs + __builtin_amdgcn_readfirstlane(bi.x), __builtin_amdgcn_readfirstlane(bi.y), __builtin_amdgcn_readfirstlane(bi.z), __builtin_amdgcn_readfirstlane(bi.w)); __syncthreads(); } }
W3_VK(1, 1) W3_VK(1, 2) W3_VK(1, 3) W3_VK(1, 4) W3_VK(2, 1) W3_VK(2, 2) W3_VK(2, 3) W3_VK(2, 4) W3_VK(3, 1) W3_VK(3, 2) W3_VK(3, 3) W3_VK(4, 1) W3_VK(4, 2)
}
#endif
