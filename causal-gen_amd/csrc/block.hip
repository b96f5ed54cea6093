// Fused "light" Block for gfx950 (MI355X): two 3x3 convolutions per launch, the bottleneck tensor resident in LDS.
//   vae.py:49-56,60-71,73-84 (version == "light"):  out = conv2(relu(conv1(relu(cat(segs))))) (+ residual), bottleneck b = in/4
// and the same kernel as the Block's data gradient (aten::convolution_backward x2 + threshold_backward x2, input part):
//   g_t = conv(g_out; dgrad image of conv2) * relu'(t),   g_x = conv(g_t; dgrad image of conv1) * relu'(x) (+ accumulated gradient).
//
// Structure (DESIGN.md section 3 has the measurements that led here):
//   * one workgroup (4 waves) per 8x16 tile of the Block's OUTPUT, persistent over tiles; <= 53 KB of LDS and <= 168 VGPRs for the
//     narrow instances, so three workgroups share a CU (today every instance takes 256 VGPRs: two per CU);
//   * phase A (long K, <= 32 outputs) STREAMS the 12x20 halo tile through a two-slot LDS ring in 32-channel chunks by LDS-DMA
//     (LDS use is independent of the input width: any number of virtual-cat segments); the chunk's 18 weight fragments come straight
//     from L2 into registers, in a fragment-ordered image (one contiguous KiB per wave load), one chunk ahead of their use.
//     The K16-steps of a chunk are dealt to two wave pairs (K split 2 x pixel split 2: every wave owns 3 pixel groups of 32 and half
//     the taps), the two partial sums meet in LDS in a fixed order => deterministic;
//   * the bottleneck tile (10x18 pixels) lives in LDS only; its interior is written once for the weight gradients / backward mask;
//   * phase B (short K, wide output): waves split the output channels in 32-wide pairs (and the tile rows when there are fewer than
//     four pairs); weight fragments straight from L2, B operands from the LDS-resident bottleneck; epilogue from the accumulators with
//     32 contiguous bytes per lane (the weight rows are permuted so that a lane's 16 accumulator rows are 16 consecutive channels).
// MFMA: v_mfma_f32_32x32x16_f16.  Its B operand puts 32 PIXELS of one 8-channel group in lanes 0-31, so with an odd pixel stride
// (in 16-byte groups) the ds_read_b128 service groups {0-3,12-15,20-27} ... of MI355X_MICROARCH.md hit 16 distinct bank groups:
// conflict-free fragment reads, which no pixel-major layout gives the 16x16x32 form (its lanes 0-15 / 16-31 read different channel
// groups of 16 pixels: two of every 16 lanes collide).
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace cgen {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x2v __attribute__((ext_vector_type(2)));

__device__ uint4 g_b3zero[4];  // 64 bytes of zeros: source of out-of-image / padding loads

// Tile: TH x 16 output pixels, TH = 8 (any instance) or 12 (the wide-output instances: 576 eight-row tiles of a 48x48 batch of 32
// are two rounds on the 512 resident workgroups, 384 twelve-row tiles are one)
#define B3_TW 16
#define B3_HW 20                    // halo tile: (TH + 4) x 20 pixels
#define B3_MW 18                    // bottleneck tile: (TH + 2) x 18 pixels
#define B3_XS 80                    // bytes per halo pixel in LDS: 4 channel groups of 16 B + one pad group (odd stride)
constexpr int b3_ndi(int th) { return ((th + 4) * B3_HW + 47) / 48; }            // LDS-DMA instructions per wave and chunk (12 halo pixels each): 5 / 7
constexpr int b3_xbytes(int th) { return 4 * b3_ndi(th) * 12 * B3_XS + 64; }     // one ring slot (whole instructions + the overhang of the last one): 19 264 / 26 944
constexpr int b3_nmp(int th) { return (th + 2) * B3_MW; }                         // bottleneck pixels: 180 / 252

struct B3Div { uint32_t mul, shift; };
static inline B3Div b3_mkdiv(uint32_t d) {
  B3Div f;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t sh = 0;
  while ((1u << sh) < d) ++sh;
  f.shift = sh;
  f.mul = (uint32_t)((((uint64_t)1 << (32 + sh)) + d - 1) / d - ((uint64_t)1 << 32));
  return f;
}
__device__ __forceinline__ int b3_div(int n, const B3Div& f) { return (int)(((uint64_t)__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shift); }

struct BV3 { const char* p; int sn, sh, sw; };  // 32-bit BYTE strides (the host checks every view spans < 2^31 bytes)
struct B3Out {
  const char* w;      // phase-B fragment image [pair][K16-step][lane][8]
  const float* bias;  // [Co] or null
  BV3 out, aux, res;  // aux: mask (v = aux > 0 ? v : 0); res: added
  int Co, npb;
  int out_rem, res_rem, pad0, pad1;  // remainder planes of a residual trunk (byte offsets from out.p / res.p, 0 = none): value = hi + rem
};
struct B3P {
  int N, H, W, nseg, nch, b, nout, nksB;
  int tiles_x, tiles_y, ntiles, ctot8;
  int ns, wb_persist, scratch_off, bias_off;
  int tm_off, tm_bytes, pad1, pad2;  // ring slots; phase-B weights persistent in registers; LDS offset of the second reduction scratch half (0: none)
  B3Div d_tx, d_ty;
  BV3 seg[3];
  int seg_koff[4];  // first CHUNK of segment s (segments are padded to whole 32-channel chunks); [nseg] = nch
  int seg_c8[3];
  // small-image instance (blk3s, images up to 14 pixels wide): a workgroup owns `s_rows` output rows of one image
  int small, s_rows, s_strips, s_nmb;  // s_nmb: 32-row blocks of the bottleneck (1 or 2)
  int s_mid_off, s_kt_off, s_red_off, s_dbg, s_segk0[3], s_segg[3];  // LDS offsets; first chunk / 16-byte groups per pixel of segment s
  B3Div s_dg1, s_dxw, s_dw, s_dstrips, s_dnb;
  // row-streaming instance (blk3r): strips of 32 columns x r_rs rows
  int rows, r_rs, r_sx, r_sy, r_res_tile, r_pad;
  const char* wA16;  // 16-row image of conv1 ([tap][chunk][lane][8]; cgen_weight_prep modes 6 / 7)
  const char* wA;   // phase-A fragment image [32-row block][chunk][channel half 0..1][tap 0..8][lane][8]
  const float* biasA;
  BV3 mid, mid_aux;  // mid: written (interior pixels): forward t (pre-activation), backward g_t; mid_aux: backward mask source t
  B3Out o[2];
  unsigned long long* stamps;  // optional (CGEN_BLK3_STAMPS=<device address>): shader-clock stamps of workgroup 0, 8 per tile after 2 launch stamps
};

__device__ __forceinline__ void b3_pin(h16x8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ h16x8 b3_relu8(h16x8 v) {  // ReLU on the raw bits: one v_pk_max_i16 per channel pair (-0 -> +0)
  union { h16x8 h; s16x2v s[4]; } c;
  c.h = v;
#pragma unroll
  for (int e = 0; e < 4; ++e) c.s[e] = __builtin_elementwise_max(c.s[e], (s16x2v){0, 0});
  return c.h;
}
__device__ __forceinline__ f32x16 b3_mfma(h16x8 a, h16x8 b, f32x16 c) {
#ifdef CGEN_H16_BF16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ uint4 b3_pack8(const float* v) {
  uint4 o;
  o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]); o.z = f2h_pk(v[4], v[5]); o.w = f2h_pk(v[6], v[7]);
  return o;
}
#define B3_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define B3_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define B3_VMWAIT_N(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// a 16-byte global load the compiler does NOT track (no automatic s_waitcnt in front of its uses): the caller waits by count
// (B3_VMWAIT_N) and launders the register (b3_pin) before the first use.  The "memory" clobber keeps it in program order with the
// LDS-DMA requests around it -- the counted waits depend on that order.  s_nop 4: the base may have been written by a VALU
// instruction (v_readlane of a spilled SGPR) right in front of the statement, and the hazard recogniser does not look inside it
// (VALU writes SGPR -> VMEM reads it: 5 wait states; without them the load uses the stale register: a memory fault).
// one LDS-DMA instruction (16 B per lane -> LDS at `lds` + 16 lane), also invisible to the compiler's wait-count pass: no automatic
// vmcnt(0) in front of LDS accesses it cannot prove disjoint, no limit on how many requests may be in flight; ordering is by the
// counted waits + barriers of the tile loop alone.  (Nothing else in these kernels uses M0; s_nop: M0 write -> LDS-DMA hazard.)
__device__ __forceinline__ void b3_dma16(const char* src, const uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds) : "memory");
}
template <int N>
__device__ __forceinline__ void b3_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [LO, HI]: the count is an immediate, so a decision tree of them
template <int LO, int HI>
__device__ __forceinline__ void b3_vmwait_rt(const int n) {
  if constexpr (LO == HI) {
    b3_vmwait<LO>();
  } else {
    constexpr int M = (LO + HI) / 2;
    if (n <= M) b3_vmwait_rt<LO, M>(n); else b3_vmwait_rt<M + 1, HI>(n);
  }
}
// (untracked, like b3_gload, with a per-lane 64-bit address: the small-image instance's weight fragments)
__device__ __forceinline__ void b3_gload_v(h16x8& d, const char* vaddr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(vaddr) : "memory");
}
typedef unsigned int b3_u32x4 __attribute__((ext_vector_type(4)));  // (native vectors: an asm operand cannot be a HIP_vector_type struct)
typedef unsigned int b3_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void b3_gload_v16(b3_u32x4& d, const char* vaddr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(vaddr) : "memory"); }
__device__ __forceinline__ void b3_gload_v8(b3_u32x2& d, const char* vaddr) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(vaddr) : "memory"); }
__device__ __forceinline__ void b3_pin4(b3_u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void b3_pin2(b3_u32x2& v) { asm volatile("" : "+v"(v)); }
template <int OFF>
__device__ __forceinline__ void b3_gload(h16x8& d, const char* sbase, const int voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

// the 9 K16-steps (one per tap) of one 32-channel chunk that this wave owns: K half kh = channels 16 kh .. + 16 of the chunk (folded
// into pbA), fragment reads issued one step ahead of the MFMAs that consume them
template <bool PRE, int NG>
__device__ __forceinline__ void b3_chunk(const char* __restrict__ Xc, const h16x8 (&Ac)[9], const int (&pbA)[NG], f32x16 (&acc)[NG]) {
  h16x8 bq[2][NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) bq[0][g] = *(const h16x8*)(Xc + pbA[g]);
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    if (s + 1 < 9) {
      const int imm = (((s + 1) / 3) * B3_HW + (s + 1) % 3) * B3_XS;
#pragma unroll
      for (int g = 0; g < NG; ++g) bq[(s + 1) & 1][g] = *(const h16x8*)(Xc + pbA[g] + imm);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = b3_mfma(Ac[s], PRE ? b3_relu8(bq[s & 1][g]) : bq[s & 1][g], acc[g]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// PRE: ReLU on the phase-A input (forward); NB = bottleneck width / 8 (1..4); NPG = output pixel groups (of 32) a wave owns in
// phase B: 1 (one 32-channel pair, four waves split the tile), 2 (two pairs x two halves), 4 (>= 3 pairs: a wave owns pairs w, w+4, ..)
//
// Memory pipeline.  The halo chunks travel through a ring of p.ns LDS slots, requested in BURSTS: all chunks of a tile at once (as
// many as there are slots), and -- when the ring has a slot more than a tile needs -- the whole NEXT tile at the start of this
// tile's last chunk, so that in the streaming regime (192x192 / 96x96: one or two chunks per tile, many tiles per workgroup) a
// tile's input is already in LDS when its turn comes.  Loads return in order and hipcc drains every LDS-DMA in flight at the
// first use of an ordinary load's result, so ordinary loads are kept out of the spans a burst should survive: weights are
// PERSISTENT in registers where a tile needs <= 2 chunks / the wave's output pair never changes (loaded once per launch), the
// epilogue operands are requested BEFORE the next tile's burst and consumed at the very end of the tile.
// (the kernel body: `bid` of `nb` workgroups walk the tiles of problem p -- the whole grid, or one half of a pair launch)
template <bool PRE, int NB, int NPG, int SM, int TH, bool REM>
__device__ __forceinline__ void blk3_body(const B3P& p, const int bid, const int nb, const int koff) {  // koff: byte offset of p in the kernarg segment
  constexpr int NMP = b3_nmp(TH), XB = b3_xbytes(TH), NDI = b3_ndi(TH);
  constexpr int NG = (NMP + 63) / 64;  // bottleneck pixel groups (of 32) per wave in phase A: 3 / 4
  constexpr int PGW = (TH / 2) / (NPG == 4 ? 1 : (NPG == 2 ? 2 : 4));  // output pixel groups a wave owns in phase B
  static_assert((TH / 2) % (NPG == 4 ? 1 : (NPG == 2 ? 2 : 4)) == 0, "tile height and phase-B wave split");
  __builtin_amdgcn_s_setprio(3);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  constexpr int NBS = (NB & 1) ? NB : NB + 1;  // bottleneck pixel stride in 16-byte groups (odd)
  constexpr int MS = NBS * 16;
  constexpr int MAXKB = (9 * NB + 1) / 2;      // K16-steps of phase B (= ceil(9 b / 16))
  constexpr int GP = PGW == 1 ? 1 : (PGW == 3 ? 3 : 2);  // pixel groups per pass of phase B (four groups = two passes over the pair's weights: half the registers)
  constexpr int RD = MAXKB;  // phase-B weights of a wave's pair: all K16-steps in registers (the phase-A fragments are dead by then)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = p.ns;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kg = lane >> 5;
  const int kh = wave & 1, gp = wave >> 1;
  const int H = p.H, W = p.W, nch = p.nch, bch = p.b;
  constexpr bool bwd = !PRE;  // (the host pairs them: forward = ReLU on the input + bias, backward = mask from mid_aux)
  const char* const zero = (const char*)g_b3zero;
  unsigned long long* const stamp = (p.stamps != nullptr && bid == 0 && lane == 0) ? p.stamps + wave * 256 : nullptr;  // (every wave's first lane: 256 slots each)
  int nstamp = 2;
#define B3_STAMP(k) do { if (stamp && nstamp + (k) < 250) stamp[nstamp + (k)] = __builtin_readcyclecounter(); } while (0)
  if (stamp) stamp[0] = __builtin_readcyclecounter();
  // SM (streaming mode): 0 = nothing persistent (any number of chunks); 1 = a tile is ONE chunk (192x192: 32 channels): the phase-A
  // weights stay in registers for the whole launch, and so do the phase-B weights (the host picks it only when the wave's pair never
  // changes).  (A two-chunk streaming mode existed until the any-chunk path got its two-ahead ring: 96x96 then ran 49 us against 52.)
  constexpr bool persistA = SM > 0, persistB = SM > 0;

  // ---- phase-A lane constants: this lane's bottleneck pixel of group g (3 groups of 32 per wave; 6 x 32 = 192 >= 180)
  int pbA[NG], mpx[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int m = 32 * (gp * NG + g) + px;
    const int mc = min(m, NMP - 1);
    const int my = (mc * 57) >> 10, mx = mc - my * B3_MW;  // (mc / 18, exact for mc < 252)
    pbA[g] = (my * B3_HW + mx) * B3_XS + kh * 32 + kg * 16;
    mpx[g] = m < NMP ? (my << 8 | mx) : -1;
  }
  // ---- DMA lane constants: a wave instruction fills 12 halo pixels x 5 groups; wave w issues instructions w, w + 4, ... of the 20
  // per chunk.  Every lane is active (no exec branches): group 4 is the padding slot (zeros), lanes 60-63 re-write the first four
  // groups of the NEXT instruction's first pixel with the same bytes (the last instruction's overhang lands behind the slot)
  const int dpl = lane / 5, dq = lane - 5 * dpl;
  // halo pixel (hy, hx) of this lane in the wave's i-th instruction, 10 bits each (hy << 5 | hx), three per register;
  // lmask bit i: a real pixel and a real channel group
  uint32_t hq[(NDI + 2) / 3];
  int lmask = 0;
#pragma unroll
  for (int i = 0; i < (NDI + 2) / 3; ++i) hq[i] = 0;
#pragma unroll
  for (int i = 0; i < NDI; ++i) {
    const int pi = 12 * (wave + 4 * i) + dpl;
    const int hy = (pi * 3277) >> 16, hx = pi - hy * B3_HW;  // (pi / 20, exact for pi < 340)
    hq[i / 3] |= (uint32_t)(hy << 5 | hx) << (10 * (i % 3));
    if (dq < 4 && hy < TH + 4) lmask |= 1 << i;
  }
  auto hy_of = [&](const int i) { return (int)((hq[i / 3] >> (10 * (i % 3) + 5)) & 31); };
  auto hx_of = [&](const int i) { return (int)((hq[i / 3] >> (10 * (i % 3))) & 31); };
  // ---- phase-B lane constants
  const int kgmask = kg ? -1 : 0;
  const int pbB = ((px >> 4) * B3_MW + (px & 15)) * MS;  // out pixel (2 pg + (px >> 4), px & 15) -> bottleneck tile offset (pg part is an immediate)
  const int pg0 = NPG == 4 ? 0 : (NPG == 2 ? PGW * (wave >> 1) : PGW * wave);

  auto tile_of = [&](const int tile, int& n, int& y0, int& x0) {
    const int b1 = b3_div(tile, p.d_tx), tx = tile - b1 * p.tiles_x;
    n = b3_div(b1, p.d_ty);
    y0 = (b1 - n * p.tiles_y) * TH; x0 = tx * B3_TW;
  };
  // which of this lane's five halo pixels of tile (y0, x0) lie inside the image (once per tile, not per chunk)
  auto tile_valid = [&](const int y0, const int x0) {
    int vb = 0;
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const int iy = y0 - 2 + hy_of(i), ix = x0 - 2 + hx_of(i);
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) vb |= 1 << i;
    }
    return vb & lmask;
  };
  // halo pixels of chunk j of tile (n, y0, x0) -> ring slot Xn, by LDS-DMA: five instructions per wave, ~10 VALU each (the first
  // version walked them in a rolled loop with the divisions and bounds checks inside: ~2000 cycles per chunk and wave, more than
  // the chunk's MFMAs)
  auto dma_chunk = [&](char* Xn, const int n, const int y0, const int x0, const int vb, const int j) {
    // chunk j belongs to ONE segment (the concatenated axis pads every segment to whole 32-channel chunks: the parents' few
    // channels cost a fractional chunk either way), so everything about the source but the lane's pixel is wave-uniform
    int sg = 0;
    if (p.nseg > 1 && j >= p.seg_koff[1]) sg = 1;
    if (p.nseg > 2 && j >= p.seg_koff[2]) sg = 2;
    // (scalar loads straight from the kernarg segment, indexed by the wave-uniform segment number: written as a chain of selects
    //  over p.seg[...], hipcc builds a table in SCRATCH and the lookups land on vmcnt, in the middle of the DMA requests)
    typedef const char __attribute__((address_space(4)))* karg_ptr;
    const karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + koff;
    typedef const BV3 __attribute__((address_space(4)))* kseg_ptr;
    typedef const int __attribute__((address_space(4)))* kint_ptr;
    const kseg_ptr ks = (kseg_ptr)(ka + offsetof(B3P, seg)) + sg;
    const char* const sp = ks->p;
    const int sn = ks->sn, sh = ks->sh, sw = ks->sw;
    const int c8 = ((kint_ptr)(ka + offsetof(B3P, seg_c8)))[sg], k0 = ((kint_ptr)(ka + offsetof(B3P, seg_koff)))[sg];
    const int cs = 32 * (j - k0) + 8 * dq;  // this lane's channel group inside the segment
    const int vbc = cs < c8 ? vb : 0;
    const char* base = sp + (n * sn + (y0 - 2) * sh + (x0 - 2) * sw);
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr)Xn) + wave * (12 * B3_XS);
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
      const uint32_t off = __umul24(hy_of(i), sh) + __umul24(hx_of(i), sw) + cs * 2;
      const char* src = ((vbc >> i) & 1) ? base + off : zero;
      b3_dma16(src, la + i * (4 * 12 * B3_XS));
    }
  };
  // (SM > 0, backward) the bottleneck mask tile t[10 x 18 pixels][b] of tile (n, y0, x0) -> LDS, dense, by DMA: instruction i of
  // the 3 NB covers 64 consecutive 16-byte groups; wave w issues i = w, w + 4
  auto dma_mask = [&](char* dst, const int n, const int y0, const int x0) {
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr)dst);
    constexpr int NMI = (NMP * NB + 63) / 64;  // whole instructions over the dense [pixel][NB groups] tile
#pragma unroll
    for (int ii = 0; ii < (NMI + 3) / 4; ++ii) {
      const int i = wave + 4 * ii;
      if (i < NMI) {
        const int slot = 64 * i + lane, m = slot / NB, gq = slot - m * NB;
        const int my = (min(m, NMP - 1) * 57) >> 10, mx = min(m, NMP - 1) - my * B3_MW;
        const int iy = y0 - 1 + my, ix = x0 - 1 + mx;
        const bool ok = m < NMP && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const char* src = ok ? p.mid_aux.p + (n * p.mid_aux.sn + iy * p.mid_aux.sh + ix * p.mid_aux.sw) + gq * 16 : zero;
        b3_dma16(src, la + i * 1024);
      }
    }
  };
  auto load_A = [&](h16x8 (&An)[9], const int j) {  // the 9 weight fragments (this wave's K half) of chunk j
    const char* wa = p.wA + (size_t)(j * 18 + kh * 9) * 1024 + lane * 16;
#pragma unroll
    for (int s = 0; s < 9; ++s) An[s] = *(const h16x8*)(wa + s * 1024);
  };
  // (SM == 0) the same fragments by untracked loads: a chunk's DMA stays in flight under the wait for the previous chunk's weights
  // (immediate offsets up to 3 KiB: three scalar bases per call instead of nine)
  auto load_A_counted = [&](h16x8 (&An)[9], const int j) {
    const char* wa = p.wA + (size_t)(j * 18 + kh * 9) * 1024;
    const int vo = lane * 16;
    b3_gload<0>(An[0], wa, vo); b3_gload<1024>(An[1], wa, vo); b3_gload<2048>(An[2], wa, vo); b3_gload<3072>(An[3], wa, vo);
    b3_gload<0>(An[4], wa + 4096, vo); b3_gload<1024>(An[5], wa + 4096, vo); b3_gload<2048>(An[6], wa + 4096, vo); b3_gload<3072>(An[7], wa + 4096, vo);
    b3_gload<0>(An[8], wa + 8192, vo);
  };
  h16x8 A0[9], A1[SM > 0 ? 1 : 9], wbp[SM > 0 ? RD : 1];  // (wbp: the persistent phase-B weights of SM > 0)
  if constexpr (SM > 0) {
    load_A(A0, 0);
    if constexpr (persistB) {
      const char* wsrc = p.o[0].w + (size_t)(NPG == 2 ? (wave & 1) : 0) * p.nksB * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < RD; ++i) wbp[i] = *(const h16x8*)(wsrc + i * 1024);
    }
    B3_VMWAIT();  // landed, and laundered: the compiler must not carry "load pending" into the tile loop (it would drain the DMA
#pragma unroll   //  bursts at every first use)
    for (int s = 0; s < 9; ++s) b3_pin(A0[s]);
    if constexpr (persistB) {
#pragma unroll
      for (int i = 0; i < RD; ++i) b3_pin(wbp[i]);
    }
  }

  // biases -> LDS, once, by ONE LDS-DMA instruction of wave 0 (lanes 0-7: the bottleneck's 32 floats, lanes 8..: output 0's, zeros
  // elsewhere): the oldest request of the launch, covered by the first chunk's wait + barrier -- no register round trip, no
  // __syncthreads in the prologue
  if (wave == 0) {
    const int i4 = 4 * (lane - 8);
    const char* src = zero;
    if (lane < 8) { if (p.biasA != nullptr && 4 * lane < bch) src = (const char*)(p.biasA + 4 * lane); }
    else if (p.o[0].bias != nullptr && i4 < p.o[0].Co) src = (const char*)(p.o[0].bias + i4);
    b3_dma16(src, (uint32_t)(uintptr_t)(lds_ptr)(smem + p.bias_off));
  }
  // ---- (SM == 0) the halo requests of a tile's first two chunks: those of the FIRST tile go out here, before anything else of the
  // launch (the rest of the prologue -- a few hundred scalar instructions -- runs under their latency; they hold no registers),
  // those of a later tile at the end of the tile before it.  The register loads follow at the top of the tile.  Request order per
  // wave: dma 0, dma 1 (5 instructions each) | [mask loads (backward)] A 0, A 1 (9 loads each) | dma j+2, A j+2 at chunk j | ...
  // => in front of chunk 0 only A 1 may still be in flight (vmcnt(9)), in front of chunk j >= 1 dma j+1 and A j+1 (vmcnt(14))
  constexpr bool QSPLIT = NB >= 2;
  constexpr int GH = (NG + 1) / 2;  // (8-channel bottleneck: K-half 0 finishes groups 0 .. GH - 1, K-half 1 the rest)
  int tn = 0, ty0 = 0, tx0 = 0, tvb = 0;  // the tile whose first chunks are in flight
  auto issue_tile = [&](char* ring, const int tile) {
    tile_of(tile, tn, ty0, tx0);
    tvb = tile_valid(ty0, tx0);
    dma_chunk(ring, tn, ty0, tx0, tvb, 0);
    if (nch > 1) dma_chunk(ring + XB, tn, ty0, tx0, tvb, 1);
  };
  if constexpr (SM == 0) {
    if (bid < p.ntiles) issue_tile(smem, bid);
  }
  if (stamp) stamp[1] = __builtin_readcyclecounter();
  // The tile loop sees LDS through TWO __restrict__ views of the same memory: `lw` is only ever the destination of LDS-DMA, `lr` is
  // what every ds_read / ds_write goes through.  hipcc drains all LDS-DMA in flight (s_waitcnt vmcnt(0)) in front of any LDS access
  // it cannot prove disjoint from them -- with one view, the next tile's burst would be waited for at the first
  // fragment read after its request.  The ordering that matters is explicit: B3_VMWAIT + B3_BARRIER before a slot is read.
  auto run = [&](char* __restrict__ lr, char* __restrict__ lw) {
  char* const MID = lr + NS * XB;
  const float* const BIA = (const float*)(lr + p.bias_off);
  char* const TMB = lr + p.tm_off;  // (SM > 0, backward) two buffers of the bottleneck mask tile, filled by DMA with the bursts
  int sbase = 0;            // ring slot of the current tile's chunk 0
  bool prefetched = false;  // this tile's chunks were requested during the previous tile
  int tmsel = 0;            // which mask buffer holds the current tile's (SM > 0, backward)
  for (int tile = bid; tile < p.ntiles; tile += nb) {
    int n, y0, x0, vb;
    if constexpr (SM == 0) {
      n = tn; y0 = ty0; x0 = tx0; vb = tvb;  // (requested by issue_tile: in the prologue, or at the end of the previous tile)
    } else {
      tile_of(tile, n, y0, x0);
      vb = tile_valid(y0, x0);
    }
    int issued = nch;
    // (finalisation is shared by the two K halves: with a bottleneck of >= 16 channels wave kh finishes the 8-channel half q8 = kh of
    //  all three groups; with 8 channels -- only q8 = 0 exists -- K-half 0 finishes groups 0 and 1, K-half 1 group 2)
    int mo[NG];  // 1: this lane's bottleneck pixel is an interior pixel of the tile inside the image (stored to `mid`)
    bool min_img[NG];
    uint4 tm[(PRE || SM > 0) ? 1 : NG];  // mask source of the bottleneck gradient (backward, SM == 0): ordinary loads, consumed after phase A
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int my = mpx[g] >> 8, mx = mpx[g] & 255;
      const int iy = y0 - 1 + my, ix = x0 - 1 + mx;
      min_img[g] = mpx[g] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
      mo[g] = (min_img[g] && my >= 1 && my <= TH && mx >= 1 && mx <= B3_TW) ? 1 : 0;
      if constexpr (!PRE && SM == 0) {
        const int ch = 16 * kg + (QSPLIT ? 8 * kh : 0);
        const bool mine = QSPLIT || (kh == 0 ? g < GH : g >= GH);
        const bool ok = min_img[g] && mine && ch < bch;
        const char* src = ok ? p.mid_aux.p + (n * p.mid_aux.sn + iy * p.mid_aux.sh + ix * p.mid_aux.sw) + ch * 2 : zero;
        tm[g] = *(const uint4*)src;
      }
    }
    if constexpr (SM == 0) {
      load_A_counted(A0, 0);
      if (nch > 1) load_A_counted(A1, 1);
    }
    if constexpr (SM > 0) {
      if (!prefetched) {
        issued = min(nch, NS);
        for (int k = 0; k < issued; ++k) dma_chunk(lw + ((sbase + k) % NS) * XB, n, y0, x0, vb, k);
        if constexpr (!PRE) dma_mask(lw + p.tm_off + tmsel * p.tm_bytes, n, y0, x0);
      }
    }
    B3_STAMP(0);
    // ------------------------------------------------------------------ phase A
    f32x16 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;  // (the bias joins at the finalisation, from the LDS copy)
    }
    // phase-B epilogue operands of the (single) pair this wave owns when NPG <= 2: requested at the start of the tile's last chunk
    constexpr int NEPI = SM > 0 ? PGW : 1;
    uint4 ea0[NEPI][2], er0[NEPI][2];
    constexpr bool early_epi = SM > 0;  // (SM > 0: one output, the wave's pair is fixed -- checked by the host)
    const int next_tile = tile + nb;
    const bool burst_next = SM > 0 && next_tile < p.ntiles && NS >= nch + 1;
    // One chunk step.  Xc (the slot being read) and the ring (DMA destinations: always OTHER slots) are __restrict__ parameters
    // of ONE body: that is what lets hipcc keep a DMA in flight under the fragment reads.
    auto step = [&](const char* __restrict__ Xc, char* __restrict__ ring, h16x8 (&Ac)[9], auto& An, const int j, const int slot2) {
      if constexpr (SM == 0) {
        // three slots: chunk j + 2 into the slot of chunk j - 1 (everyone is through the barrier behind it); two slots (the
        // twelve-row tiles): chunk j + 1 into the slot of chunk j - 1, one chunk ahead
        if (NS == 3) { if (j + 2 < nch) dma_chunk(ring + slot2 * XB, n, y0, x0, vb, j + 2); }
        else if (j >= 1 && j + 1 < nch) dma_chunk(ring + slot2 * XB, n, y0, x0, vb, j + 1);
      } else if (j == nch - 1) {
        if (early_epi) {
          const B3Out& O = p.o[0];
          const int pair = NPG == 2 ? (wave & 1) : 0, ch0 = pair * 32 + 16 * kg;
#pragma unroll
          for (int g = 0; g < NEPI; ++g) {
            const int oy = y0 + 2 * (pg0 + g) + (px >> 4), ox = x0 + (px & 15);
            const int oa = n * O.aux.sn + oy * O.aux.sh + ox * O.aux.sw + ch0 * 2;
            const int orr = n * O.res.sn + oy * O.res.sh + ox * O.res.sw + ch0 * 2;
#pragma unroll
            for (int q8 = 0; q8 < 2; ++q8) {
              const bool ok = oy < H && ox < W && ch0 + 8 * q8 < O.Co;
              ea0[g][q8] = er0[g][q8] = make_uint4(0, 0, 0, 0);
              if (O.aux.p != nullptr) ea0[g][q8] = *(const uint4*)(ok ? O.aux.p + oa + 16 * q8 : zero);
              if (O.res.p != nullptr) er0[g][q8] = *(const uint4*)(ok ? O.res.p + orr + 16 * q8 : zero);
            }
          }
        }
        if (burst_next) {  // the whole next tile, into the slots behind this tile's
          int n2, y2, x2;
          tile_of(next_tile, n2, y2, x2);
          const int vb2 = tile_valid(y2, x2);
          for (int k = 0; k < nch; ++k) dma_chunk(ring + ((sbase + nch + k) % NS) * XB, n2, y2, x2, vb2, k);
          if constexpr (!PRE) dma_mask(ring + p.tm_off + (tmsel ^ 1) * p.tm_bytes, n2, y2, x2);
        }
      }
      if (j == nch - 1) B3_STAMP(7);
      b3_chunk<PRE, NG>(Xc, Ac, pbA, acc);
      if constexpr (SM == 0) {
        if (j + 2 < nch) load_A_counted(Ac, j + 2);  // into the registers this chunk has just finished with
      }
    };
    // chunk boundary: everything requested so far has landed (own loads + own DMA pieces), fragments laundered, everyone through
    auto boundary = [&](const int jj, auto& Acur) {
      if (SM > 0 && jj == issued) {  // ring exhausted (more chunks than slots): the next burst, once everyone has left the slots
        B3_BARRIER();
        const int cnt = min(nch - jj, NS);
        for (int k = 0; k < cnt; ++k) dma_chunk(lw + ((sbase + jj + k) % NS) * XB, n, y0, x0, vb, jj + k);
        issued += cnt;
      }
      // a prefetched tile's chunks landed before the previous tile ended (the wait in front of its epilogue); the backward pass
      // still waits for its mask loads at the last boundary
      if constexpr (SM == 0) {
        if (jj + 1 >= nch) B3_VMWAIT(); else if (jj == 0 || NS == 2) B3_VMWAIT_N(9); else if (NDI == 5) B3_VMWAIT_N(14); else B3_VMWAIT_N(16);
      } else if (!prefetched) {
        B3_VMWAIT();
      }
      if constexpr (!persistA) {
#pragma unroll
        for (int s = 0; s < 9; ++s) b3_pin(Acur[s]);
      }
      if constexpr (!PRE && SM == 0) {
        if (jj == nch - 1) {
#pragma unroll
          for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(tm[g].x), "+v"(tm[g].y), "+v"(tm[g].z), "+v"(tm[g].w));
        }
      }
      B3_BARRIER();
      if (jj == 0) B3_STAMP(1);
    };
    int sl = 0;  // (SM == 0) ring slot of chunk j
    auto nxt = [&](const int v) { return v + 1 == NS ? 0 : v + 1; };
    for (int j = 0; j < nch; j += 2) {
      boundary(j, A0);
      step(lr + (SM == 0 ? sl : (sbase + j) % NS) * XB, lw, A0, A1, j, NS == 3 ? nxt(nxt(sl)) : nxt(sl));
      sl = nxt(sl);
      if constexpr (SM == 0) {
        if (j + 1 < nch) {
          boundary(j + 1, A1);
          step(lr + (SM == 0 ? sl : (sbase + j + 1) % NS) * XB, lw, A1, A0, j + 1, NS == 3 ? nxt(nxt(sl)) : nxt(sl));
          sl = nxt(sl);
        }
      }
    }
    B3_STAMP(2);
    // ---- the K-half-1 waves hand their partial sums over (12 KiB each) through ring slots this tile is done with: the slot of
    // its last chunk and the one before it (everyone is past the last chunk; the slots BEHIND belong to the next tile's burst).
    // A one-chunk tile has a single slot: the second wave's half sits in a region of its own behind the bottleneck tile.
    B3_BARRIER();
    // (SM == 0: sl is one past the last chunk's slot; nothing is in flight here, the last boundary drained the queue)
    const int sl_last = SM == 0 ? (sl == 0 ? NS - 1 : sl - 1) : 0, sl_prev = SM == 0 ? (sl_last == 0 ? NS - 1 : sl_last - 1) : 0;
    char* const scr = lr + (SM == 0 ? sl_last : (sbase + nch - 1) % NS) * XB;
    // (streaming modes: the slot before the last chunk's already belongs to the next tile's burst -- a region of its own as well)
    char* const scr2 = (SM == 0 && nch >= 2) ? lr + sl_prev * XB : lr + p.scratch_off;
    char* const sx = gp == 0 ? scr : scr2;  // this wave pair's 12 KiB
    auto send = [&](const int g, auto Q) {  // the 8 accumulator rows of half Q of group g -> the partner
      constexpr int q8 = decltype(Q)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *(float4*)(sx + ((kh * NG + g) * 2 + h) * 1024 + lane * 16) =
            make_float4(acc[g][8 * q8 + 4 * h], acc[g][8 * q8 + 4 * h + 1], acc[g][8 * q8 + 4 * h + 2], acc[g][8 * q8 + 4 * h + 3]);
    };
    const std::integral_constant<int, 0> q0;
    const std::integral_constant<int, 1> q1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (QSPLIT) { if (kh == 0) send(g, q1); else send(g, q0); }
      else if ((kh == 0) == (g >= GH)) send(g, q0);  // (the half that does NOT finish group g hands it over)
    }
    B3_BARRIER();
    B3_STAMP(3);
    auto finish = [&](const int g, auto Q) {  // bottleneck pixel of group g, channels 16 kg + 8 Q .. + 8: sum, store, LDS
      constexpr int q8 = decltype(Q)::value;
      const int ch = 16 * kg + 8 * q8;
      if (mpx[g] < 0 || ch >= bch) return;
      float v[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 o = *(const float4*)(sx + (((1 - kh) * NG + g) * 2 + h) * 1024 + lane * 16);
        const float4 bb = *(const float4*)(BIA + ch + 4 * h);
        v[4 * h] = (acc[g][8 * q8 + 4 * h] + o.x) + bb.x; v[4 * h + 1] = (acc[g][8 * q8 + 4 * h + 1] + o.y) + bb.y;
        v[4 * h + 2] = (acc[g][8 * q8 + 4 * h + 2] + o.z) + bb.z; v[4 * h + 3] = (acc[g][8 * q8 + 4 * h + 3] + o.w) + bb.w;
      }
      const int my = mpx[g] >> 8, mx = mpx[g] & 255;
      const int iy = y0 - 1 + my, ix = x0 - 1 + mx;
      char* gdst = (char*)p.mid.p + (n * p.mid.sn + iy * p.mid.sh + ix * p.mid.sw) + ch * 2;
      char* ldst = MID + (my * B3_MW + mx) * MS + ch * 2;
      if (bwd) {  // g_t = acc * relu'(t); zero outside the image because t was read as zero there
        uint4 tv = tm[(PRE || SM > 0) ? 0 : g];
        if constexpr (SM > 0) tv = *(const uint4*)(TMB + tmsel * p.tm_bytes + (my * B3_MW + mx) * (NB * 16) + ch * 2);
        const uint32_t w[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = h_lo(w[e]) > 0.f ? v[2 * e] : 0.f;
          v[2 * e + 1] = h_hi(w[e]) > 0.f ? v[2 * e + 1] : 0.f;
        }
        const uint4 o = b3_pack8(v);
        if (mo[g]) *(uint4*)gdst = o;
        *(uint4*)ldst = o;
      } else {  // t = acc (+ bias, already in); LDS gets relu(t), zero outside the image (conv2 pads ITS input with zeros)
        const uint4 o = b3_pack8(v);
        if (mo[g]) *(uint4*)gdst = o;
        union { uint4 q; h16x8 h; } r;
        r.q = o;
        r.h = b3_relu8(r.h);
        *(uint4*)ldst = min_img[g] ? r.q : make_uint4(0, 0, 0, 0);
      }
    };
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if constexpr (QSPLIT) { if (kh == 0) finish(g, q0); else finish(g, q1); }
      else if ((kh == 0) == (g < GH)) finish(g, q0);
    }
    B3_BARRIER();
    B3_STAMP(4);
    // ------------------------------------------------------------------ phase B
#pragma unroll 1
    for (int oi = 0; oi < p.nout; ++oi) {
      // (read through the kernarg segment: `p.o[oi]` with a run-time index takes the address of the by-value struct, and hipcc
      //  then keeps a 472-byte copy of it in scratch -- every later field read becomes a scratch load counted on vmcnt)
      typedef const B3Out __attribute__((address_space(4)))* kout_ptr;
      const kout_ptr Ok = (kout_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + koff + offsetof(B3P, o)) + oi;
      const char* const Ow = Ok->w;
      const char* const Oout = Ok->out.p; const int Oout_sn = Ok->out.sn, Oout_sh = Ok->out.sh, Oout_sw = Ok->out.sw;
      const char* const Oaux = Ok->aux.p; const int Oaux_sn = Ok->aux.sn, Oaux_sh = Ok->aux.sh, Oaux_sw = Ok->aux.sw;
      const char* const Ores = Ok->res.p; const int Ores_sn = Ok->res.sn, Ores_sh = Ok->res.sh, Ores_sw = Ok->res.sw;
      const int npb = Ok->npb, Co = Ok->Co, nks = p.nksB;
      const int Oout_rem = Ok->out_rem, Ores_rem = Ok->res_rem;  // (forward trunk Blocks of an inference pass; 0 otherwise)
#pragma unroll 1
      for (int r = 0;; ++r) {
        const int pair = NPG == 4 ? wave + 4 * r : (NPG == 2 ? (wave & 1) + 2 * r : r);
        if (pair >= npb) break;
        // weights of the pair: all K16-steps in registers, for all passes over it (a ring refilled per pass read them once per
        // pass, each refill one L2 round trip in front of its MFMA; the phase-A fragments are dead by now); the epilogue
        // operands are requested behind them (loads return in order: a weight fragment queued behind an HBM-cold residual
        // would wait for it) and are in flight under the MFMAs
        const char* wsrc = Ow + (size_t)pair * nks * 1024 + lane * 16;
        h16x8 wfull[SM == 0 ? MAXKB : 1];
        if constexpr (SM == 0) {
#pragma unroll
          for (int i = 0; i < MAXKB; ++i) wfull[i] = *(const h16x8*)(wsrc + i * 1024);
        }
        const int ch0 = pair * 32 + 16 * kg;
        const bool has_aux = Oaux != nullptr, has_res = Ores != nullptr;
#pragma unroll 1
        for (int pass = 0; pass < PGW / GP; ++pass) {
          const int pgb = pg0 + pass * GP;
          auto& wb = *[&]() { if constexpr (SM > 0) return &wbp; else return &wfull; }();
          uint4 ea[GP][2], er[GP][2], el[REM ? GP : 1][2];  // (el: the residual's remainder plane; REM instances only, so that the plain ones keep their registers)
          int eoff_o[GP];
          bool ev[GP];
#pragma unroll
          for (int g = 0; g < GP; ++g) {
            const int oy = y0 + 2 * (pgb + g) + (px >> 4), ox = x0 + (px & 15);
            ev[g] = oy < H && ox < W;
            eoff_o[g] = n * Oout_sn + oy * Oout_sh + ox * Oout_sw + ch0 * 2;
          }
          auto epi_request = [&]() {
#pragma unroll
            for (int g = 0; g < GP; ++g) {
              const int oy = y0 + 2 * (pgb + g) + (px >> 4), ox = x0 + (px & 15);
              const int oa = n * Oaux_sn + oy * Oaux_sh + ox * Oaux_sw + ch0 * 2;
              const int orr = n * Ores_sn + oy * Ores_sh + ox * Ores_sw + ch0 * 2;
#pragma unroll
              for (int q8 = 0; q8 < 2; ++q8) {
                const bool ok = ev[g] && ch0 + 8 * q8 < Co;
                ea[g][q8] = er[g][q8] = make_uint4(0, 0, 0, 0);
                if (has_aux) ea[g][q8] = *(const uint4*)(ok ? Oaux + oa + 16 * q8 : zero);
                if (has_res) er[g][q8] = *(const uint4*)(ok ? Ores + orr + 16 * q8 : zero);
                if constexpr (REM) {
                  el[g][q8] = make_uint4(0, 0, 0, 0);
                  if (has_res && Ores_rem != 0) el[g][q8] = *(const uint4*)(ok ? Ores + Ores_rem + orr + 16 * q8 : zero);
                }
              }
            }
          };
          if (early_epi) {
#pragma unroll
            for (int g = 0; g < NEPI; ++g) { ea[g][0] = ea0[g][0]; ea[g][1] = ea0[g][1]; er[g][0] = er0[g][0]; er[g][1] = er0[g][1]; }
          } else {
            epi_request();
          }
          f32x16 ac[GP];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            float4 bb = *(const float4*)(BIA + 32 + ch0 + 4 * q4);  // (the LDS copy holds output 0's bias, zero padded)
            if (oi != 0) bb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int g = 0; g < GP; ++g) { ac[g][4 * q4] = bb.x; ac[g][4 * q4 + 1] = bb.y; ac[g][4 * q4 + 2] = bb.z; ac[g][4 * q4 + 3] = bb.w; }
          }
          const char* mbase = MID + pbB + pgb * (2 * B3_MW * MS);
          // K16-step i: lane half kg reads the 8-channel group u = 2 i + kg of the flattened (tap, channel group) axis; fragment
          // reads one step ahead of the MFMAs
          auto kaddr = [&](const int i) {
            const int uE = 2 * i, uO = 2 * i + 1;
            const int tE = uE / NB < 9 ? uE / NB : 8, tO = uO / NB < 9 ? uO / NB : 8;
            const int offE = ((tE / 3) * B3_MW + tE % 3) * MS + (uE % NB) * 16;
            const int offO = ((tO / 3) * B3_MW + tO % 3) * MS + (uO % NB) * 16;
            return mbase + offE + (kgmask & (offO - offE));
          };
          h16x8 bq[2][GP];
          {
            const char* rp = kaddr(0);
#pragma unroll
            for (int g = 0; g < GP; ++g) bq[0][g] = *(const h16x8*)(rp + g * (2 * B3_MW * MS));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MAXKB; ++i) {
            if (i + 1 < MAXKB) {
              const char* rp = kaddr(i + 1);
#pragma unroll
              for (int g = 0; g < GP; ++g) bq[(i + 1) & 1][g] = *(const h16x8*)(rp + g * (2 * B3_MW * MS));
            }
#pragma unroll
            for (int g = 0; g < GP; ++g) ac[g] = b3_mfma(wb[i], bq[i & 1][g], ac[g]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (oi == 0 && r == 0 && pass == 0) B3_STAMP(5);
          if constexpr (SM > 0) B3_VMWAIT();  // the early epilogue operands AND the next tile's burst (requested a whole tile ago): the only wait of a streaming tile
          // epilogue straight from the accumulators: 16 consecutive channels of one pixel per lane
#pragma unroll
          for (int g = 0; g < GP; ++g) {
#pragma unroll
            for (int q8 = 0; q8 < 2; ++q8) {
              if (!(ev[g] && ch0 + 8 * q8 < Co)) continue;
              float u[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) u[e] = ac[g][8 * q8 + e];
              if (has_aux) {
                const uint32_t w[4] = {ea[g][q8].x, ea[g][q8].y, ea[g][q8].z, ea[g][q8].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  u[2 * e] = h_lo(w[e]) > 0.f ? u[2 * e] : 0.f;
                  u[2 * e + 1] = h_hi(w[e]) > 0.f ? u[2 * e + 1] : 0.f;
                }
              }
              if (has_res) {
                const uint32_t w[4] = {er[g][q8].x, er[g][q8].y, er[g][q8].z, er[g][q8].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { u[2 * e] += h_lo(w[e]); u[2 * e + 1] += h_hi(w[e]); }
                if constexpr (REM) {
                  if (Ores_rem != 0) {
                    const uint32_t wl[4] = {el[g][q8].x, el[g][q8].y, el[g][q8].z, el[g][q8].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { u[2 * e] += h_lo(wl[e]); u[2 * e + 1] += h_hi(wl[e]); }
                  }
                }
              }
              const uint4 o16 = b3_pack8(u);
              *(uint4*)((char*)Oout + eoff_o[g] + 16 * q8) = o16;
              if constexpr (REM) {
                if (Oout_rem != 0) {  // what the rounding just dropped goes to the remainder plane: out_rem = rn16(v - out)
                  const uint32_t wo[4] = {o16.x, o16.y, o16.z, o16.w};
                  float d[8];
#pragma unroll
                  for (int e = 0; e < 4; ++e) { d[2 * e] = u[2 * e] - h_lo(wo[e]); d[2 * e + 1] = u[2 * e + 1] - h_hi(wo[e]); }
                  *(uint4*)((char*)Oout + Oout_rem + eoff_o[g] + 16 * q8) = b3_pack8(d);
                }
              }
            }
          }
        }
      }
    }
    // (the next tile's first barrier separates these reads of the bottleneck tile from its next writes)
    if constexpr (SM == 0) {
      if (tile + nb < p.ntiles) issue_tile(lw, tile + nb);  // (every ring slot is free: the exchange through them ended before phase B)
    }
    B3_STAMP(6);
    nstamp += 8;
    if (SM > 0 && burst_next) { sbase = (sbase + nch) % NS; prefetched = true; tmsel ^= 1; } else { prefetched = false; }
  }
  };
  // (the two views differ by opaque zero offsets: handed the same CONSTANT address twice, interprocedural constant propagation
  //  substitutes it for both parameters and the restrict information is gone before the lambda is inlined)
  int off_r = 0, off_w = 0;
  asm volatile("" : "+s"(off_r));
  asm volatile("" : "+s"(off_w));
  run(smem + off_r, smem + off_w);
}

template <bool PRE, int NB, int NPG, int SM, int TH, bool REM>
__global__ __launch_bounds__(256, 2) void blk3_kernel(B3P p) {
  blk3_body<PRE, NB, NPG, SM, TH, REM>(p, (int)blockIdx.x, (int)gridDim.x, 0);
}

// Two independent problems of the SAME instance in one launch: workgroups 0 .. na - 1 take the first, the rest the second (the data
// gradients of a decoder layer's posterior and prior Blocks: two kernels of 100-400 workgroups each on 512 resident slots, which a
// second queue can only overlap at ~6 us per cross-queue edge on the main chain, LABNOTES 9.9 / 10.4).  The record is picked by
// address inside the kernarg segment, so its fields stay scalar loads.
template <bool PRE, int NB, int NPG, int SM, int TH, bool REM>
__global__ __launch_bounds__(256, 2) void blk3_pair_kernel(B3P pa, B3P pb, const int na) {
  (void)pa; (void)pb;
  const bool second = (int)blockIdx.x >= na;
  constexpr size_t off_b = (sizeof(B3P) + alignof(B3P) - 1) / alignof(B3P) * alignof(B3P);
  const int koff = second ? (int)off_b : 0;
  typedef const B3P __attribute__((address_space(4)))* kp_ptr;
  const B3P& p = *(const B3P*)(kp_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + koff);
  blk3_body<PRE, NB, NPG, SM, TH, REM>(p, second ? (int)blockIdx.x - na : (int)blockIdx.x, second ? (int)gridDim.x - na : na, koff);
}

// ============================================================================= small images (<= 14 pixels wide: 12x12, 6x6, ...)
// The top of both hierarchies: ~1 % of a step's FLOPs, 22 % of its time as two latency-bound conv launches per Block (bottlenecks of
// 40 / 48 channels, which the tile kernel above does not take).  Here ONE workgroup owns R output rows of ONE image -- R = the most
// rows whose (R + 2) x W bottleneck pixels fit two 32-pixel MFMA groups: 3 of 12, all 6 of 6 -- and holds everything it needs in LDS:
//   load   the (R + 4) x (W + 2) input rows of every segment (virtual cat, zero padded, ReLU applied on the way in),
//   phase A  bottleneck[b <= 64][<= 64 pixels] = W1 * X: wave = (32-row block, pixel group), the whole K axis, weight fragments
//            straight from L2 in fragment order, nine (one 16-channel half chunk) ahead; no partial-sum exchange;
//   finalise bias / ReLU (forward) or relu'(t) mask (data gradient) -> `mid` in HBM (rows the strip owns) and the LDS tile;
//   phase B  out[Co][R x W pixels] = W2 * mid: jobs (32-channel pair, pixel group) dealt round-robin to the waves, epilogue from the
//            accumulators exactly as the tile kernel's (bias, mask, residual / accumulate, remainder planes).
// Same problem record, same fragment images, same results contract as blk3_body; bottlenecks above 32 channels take the second
// 32-row block of the phase-A image.
#define B3S_NW 8    // waves per workgroup: two per SIMD (one wave alone issues an instruction every ~9 cycles of a dependent chain: measured)
#define B3S_MW 16   // pixels per row of the bottleneck tile in LDS (W + 2 <= 16): tap offsets do not depend on the image width

// One K loop of the small-image instance: `cnt` K16-steps.  Step j multiplies the weight fragment at wbase + 1024 j (a wave-uniform
// address: one s_add per step; lane part `voff`) with the B operands at bp0 / bp1 + T[j], T an int table in LDS read TWO steps ahead
// through `tp` (TSTEP bytes per step: immediate offsets inside the unrolled body), the operands themselves ONE step ahead.  The
// fragments are untracked loads, D steps in flight, ordered by counted waits: requests return in order, so in front of step j only
// the fragments of steps j + 1 .. j + D - 1 may be outstanding.  EVERY request is unconditional -- past the last step the address is
// clamped to the last fragment -- so that the count is the same everywhere and no ring register is ever written under a branch
// (a conditionally written one gets copied at the join, by a v_mov that may run before the load has landed: measured, as garbage);
// the loop ends with a drain of the queue, because the registers of requests still in flight must not be handed back to the
// compiler.  start() issues the first D requests: callers hoist it above whatever they wait for next (a cold fragment takes 1-2 us).
// ~17 instructions per step around two MFMAs (both 32-pixel groups always: the shapes this instance exists for have more than 32
// pixels per strip).  wbase must be readable even when cnt == 0.
template <int TSTEP>
struct B3sK {
  static constexpr int D = 6;  // (a multiple of 2 and 3: the operand / table pipelines rotate inside the unrolled body)
  h16x8 wr[D];
  const char* wl;
  int jl, cnt, voff;
  __device__ __forceinline__ void issue(h16x8& dst) {
    b3_gload<0>(dst, wl, voff);
    wl += jl < cnt - 1 ? 1024 : 0;
    ++jl;
  }
  __device__ __forceinline__ void start(const char* wbase, const int voff_, const int cnt_) {
    wl = wbase; jl = 0; cnt = cnt_; voff = voff_;
#pragma unroll
    for (int d = 0; d < D; ++d) issue(wr[d]);
  }
  template <bool TWO>
  __device__ __forceinline__ void run(const char* tp, const char* bp0, const char* bp1, f32x16& c0, f32x16& c1) {
    int tv[3];
    tv[0] = *(const int*)tp;
    tv[1] = *(const int*)(tp + TSTEP);
    h16x8 bq[2][TWO ? 2 : 1];
    bq[0][0] = *(const h16x8*)(bp0 + tv[0]);
    if constexpr (TWO) bq[0][1] = *(const h16x8*)(bp1 + tv[0]);
#pragma unroll 1
    for (int i = 0; i < cnt; i += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        tv[(d + 2) % 3] = *(const int*)(tp + (d + 2) * TSTEP);
        bq[(d + 1) & 1][0] = *(const h16x8*)(bp0 + tv[(d + 1) % 3]);
        if constexpr (TWO) bq[(d + 1) & 1][1] = *(const h16x8*)(bp1 + tv[(d + 1) % 3]);
        b3_vmwait<D - 1>();
        b3_pin(wr[d]);
        if (i + d < cnt) {
          c0 = b3_mfma(wr[d], bq[d & 1][0], c0);
          if constexpr (TWO) c1 = b3_mfma(wr[d], bq[d & 1][1], c1);
        }
        issue(wr[d]);
      }
      tp += D * TSTEP;
    }
    B3_VMWAIT();  // (the clamped requests past the end: their registers are free again)
#pragma unroll
    for (int d = 0; d < D; ++d) b3_pin(wr[d]);
  }
};

template <bool PRE>
__device__ __forceinline__ void blk3s_body(const B3P& p, const int bid, const int nb, const int koff) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  typedef const char __attribute__((address_space(4)))* karg_ptr;
  const karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + koff;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, kg = lane >> 5;
  const int H = p.H, W = p.W, nch = p.nch, bch = p.b, R = p.s_rows;
  const int XW = W + 2, G1 = nch * 4 + 1, XS = G1 * 16;  // tile columns (one zero column either side); 16-byte slots / bytes per input pixel (one pad slot: odd stride)
  const int NB = bch >> 3, NBS = (NB & 1) ? NB : NB + 1, MS = NBS * 16;  // bottleneck pixel stride: an odd number of 16-byte groups
  char* const X = smem;
  char* const MID = smem + p.s_mid_off;   // [R + 2 rows][B3S_MW pixels][MS]
  char* const RED = smem + p.s_red_off;   // partial sums: [wave][pixel group 0..1][quad 0..3][64 lanes][4 floats] (in the input tile's memory)
  int* const KT = (int*)(smem + p.s_kt_off);  // phase-B K axis: 8-channel group u = (tap, bottleneck group) -> byte offset in the bottleneck tile
  int* const TA = KT + 128;                   // phase-A K axis: step s = (half chunk q, tap) in image order -> byte offset in the input tile
  const char* const zero = (const char*)g_b3zero;
  const int nks = p.nksB, nmb = p.s_nmb, nst = 18 * nch;
  if (p.s_dbg & 8) return;
  unsigned long long* const stamp = (p.stamps != nullptr && bid == 0 && lane == 0) ? p.stamps + wave * 32 : nullptr;  // (debug: shader-clock stamps of workgroup 0, tools/blk3s_stamps.py)
#define B3S_STAMP(k) do { if (stamp) stamp[k] = __builtin_readcyclecounter(); } while (0)
  B3S_STAMP(0);
  typedef const BV3 __attribute__((address_space(4)))* kseg_ptr;
  const kseg_ptr ks = (kseg_ptr)(ka + offsetof(B3P, seg));
  const int NMP = (R + 2) * W;  // bottleneck pixels a strip computes (<= 64)
  // phase-A roles: wave = (32-row block mb, K part kq of KQ); finalisers: wave f < 4 nmb = (mb, pixel group, 8-channel half)
  const int KQ = B3S_NW / nmb, mbA = wave / KQ, kqA = wave - mbA * KQ;
  const int SA = (nst + KQ - 1) / KQ, sA0 = min(kqA * SA, nst), mineA = min(sA0 + SA, nst) - sA0;
  const int fmb = wave >> 2, fng = (wave >> 1) & 1, fhalf = wave & 1;
  // ONE item per workgroup (the host launches p.ntiles of them).  Not a loop: everything below is invariant in everything but the
  // item, and hipcc hoisted all of it out of an item loop and kept it live across -- 90 spilled registers, their reloads (scratch
  // loads the compiler waits for with vmcnt(0)) in the middle of the request bursts.
  (void)nb;
  {
    const int item = bid;
    B3sK<4> KA;
    B3sK<8> KB;
    const int n = b3_div(item, p.s_dstrips), y0 = (item - n * p.s_strips) * R;
    const int Rr = min(R, H - y0), NOP = Rr * W;
    // ---- input rows y0 - 2 .. y0 + R + 1, columns -1 .. W, every segment (virtual cat), -> LDS by DMA: one 16-byte slot per lane,
    // zeros outside the image / past a segment's channels / in the pad slot.  The first thing a workgroup does: everything below
    // up to the barrier runs under the requests' latency.
    const int nslots = (R + 4) * XW * G1, ninst = (nslots + 63) >> 6;
    {
      const char* const p0 = ks[0].p + n * ks[0].sn;
      const char* const p1 = p.nseg > 1 ? ks[1].p + n * ks[1].sn : zero;
      const char* const p2 = p.nseg > 2 ? ks[2].p + n * ks[2].sn : zero;
      const int g1 = p.nseg > 1 ? 4 * p.s_segk0[1] : (1 << 20), g2 = p.nseg > 2 ? 4 * p.s_segk0[2] : (1 << 20);
      for (int ii = wave; ii < ((p.s_dbg & 1) ? 0 : ninst); ii += B3S_NW) {
        const int iu = __builtin_amdgcn_readfirstlane(ii);
        const int sl = min(iu * 64 + lane, nslots - 1);
        const int pxl = b3_div(sl, p.s_dg1), g = sl - pxl * G1;
        const int ry = b3_div(pxl, p.s_dxw), cx = pxl - ry * XW;
        const int iy = y0 - 2 + ry, ix = cx - 1;
        const bool s1 = g >= g1, s2 = g >= g2;
        const char* sp = s2 ? p2 : (s1 ? p1 : p0);
        const int sh = s2 ? ks[2].sh : (s1 ? ks[1].sh : ks[0].sh), sw = s2 ? ks[2].sw : (s1 ? ks[1].sw : ks[0].sw);
        const int c8 = s2 ? p.seg_c8[2] : (s1 ? p.seg_c8[1] : p.seg_c8[0]);
        const int gl = g - (s2 ? g2 : (s1 ? g1 : 0));  // 8-channel group inside the segment
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && 8 * gl < c8 && g < G1 - 1;
        const char* src = ok ? sp + iy * sh + ix * sw + 16 * gl : zero;
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(X + (size_t)iu * 1024), 16, 0, 0);
      }
    }
    B3S_STAMP(1);
    // ---- the first fragments of phase A (cold: 1-2 us), behind the tile's requests
    KA.start(p.wA + ((size_t)mbA * nst + min(sA0, nst - 1)) * 1024, lane * 16, (p.s_dbg & 2) ? 0 : mineA);
    // ---- lane constants of phase A: bottleneck pixel m = 32 g + px -> tile row / column; the finaliser's operands
    int mr[2], mx[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int mc = min(32 * g + px, NMP - 1);
      mr[g] = b3_div(mc, p.s_dw); mx[g] = mc - mr[g] * W;
    }
    const bool fjob = wave < 4 * nmb && (fng == 0 || NMP > 32);
    const int fiy = y0 - 1 + mr[fng];
    const bool f_img = fjob && 32 * fng + px < NMP && fiy >= 0 && fiy < H;
    const int fch = 32 * fmb + 16 * kg + 8 * fhalf;  // the 8 bottleneck channels this lane finalises
    uint4 tmk = make_uint4(0, 0, 0, 0);  // backward: the forward bottleneck t there (the mask of the bottleneck gradient); forward: the bias
    float4 fb0 = make_float4(0.f, 0.f, 0.f, 0.f), fb1 = fb0;
    if constexpr (!PRE) {
      const bool ok = f_img && fch < bch;
      tmk = *(const uint4*)(ok ? p.mid_aux.p + (n * p.mid_aux.sn + fiy * p.mid_aux.sh + mx[fng] * p.mid_aux.sw) + fch * 2 : zero);
    } else if (p.biasA != nullptr && fjob && fch < bch) {
      fb0 = *(const float4*)(p.biasA + fch); fb1 = *(const float4*)(p.biasA + fch + 4);
    }
    // ---- tables, zeros of the bottleneck tile (the border columns and the rows outside the image stay zero: conv2's padding)
    {
      for (int u = tid; u < 128; u += 64 * B3S_NW) {
        const int uu = min(u, 2 * nks - 1), tq = b3_div(uu, p.s_dnb), tp = min(tq, 8), gq = uu - tq * NB;
        KT[u] = ((tp / 3) * B3S_MW + tp % 3) * MS + gq * 16;
      }
      for (int sidx = tid; sidx < nst + 16; sidx += 64 * B3S_NW) {
        const int ss = min(sidx, nst - 1), q = ss / 9, tp = ss - 9 * q;
        TA[sidx] = ((tp / 3) * XW + tp % 3) * XS + q * 32;
      }
    }
    for (int i = tid; i < ((R + 2) * B3S_MW * MS) >> 4; i += 64 * B3S_NW) *(uint4*)(MID + i * 16) = make_uint4(0, 0, 0, 0);
    B3S_STAMP(2);
    B3_BARRIER();  // (the tables; hipcc drains vmcnt -- the DMA it tracks -- in front of the first read of the tile below)
    if constexpr (PRE) {  // ReLU once per element, in place (each is read by nine taps)
      for (int i = tid; i < ninst * 64; i += 64 * B3S_NW) {
        union { uint4 q; h16x8 h; } r;
        r.q = *(const uint4*)(X + i * 16);
        r.h = b3_relu8(r.h);
        *(uint4*)(X + i * 16) = r.q;
      }
      B3_BARRIER();
    }
    B3S_STAMP(3);
    // ---- phase A: this wave's K part of one 32-row block, both pixel groups
    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    KA.run<true>((const char*)(TA + sA0), X + (mr[0] * XW + mx[0]) * XS + kg * 16, X + (mr[1] * XW + mx[1]) * XS + kg * 16, acc[0], acc[1]);
    B3S_STAMP(4);
    // ---- phase B, output 0: roles, and the first fragments (under the exchange and the finalisation).  A job = (32-channel pair,
    // 32-pixel group, K part kp of KP); KP > 1 where there are fewer (pair, group) blocks than waves (the posterior Block's 32
    // channels: 2 blocks): the K parts meet in LDS, the block's two 8-channel halves go to its first two K-part waves.
    const int npg = NOP > 32 ? 2 : 1;
    typedef const B3Out __attribute__((address_space(4)))* kout_ptr;
    const kout_ptr Ok0 = (kout_ptr)(ka + offsetof(B3P, o));
    auto kp_of = [&](const int npb) { int KP = 1; while (2 * KP * npb * npg <= B3S_NW && nks >= 3 * 2 * KP) KP *= 2; return KP; };
    auto b_start = [&](const kout_ptr Ok, const int job) {  // (job past the last: a started ring that run() drains without using it)
      const int npb = Ok->npb, KP = kp_of(npb), njobs = npb * npg * KP;
      const int jc = min(job, njobs - 1), blk = jc / KP, kp = jc - blk * KP, pair = npg == 2 ? blk >> 1 : blk;
      const int SK = (nks + KP - 1) / KP, k0 = min(kp * SK, nks - 1);
      const int cntB = job < njobs ? min(kp * SK + SK, nks) - min(kp * SK, nks) : 0;
      KB.start(Ok->w + ((size_t)pair * nks + k0) * 1024, lane * 16, cntB);
    };
    if (!(p.s_dbg & 4)) b_start(Ok0, wave);
    B3_BARRIER();  // (everyone is done with the input tile: the exchange buffer lies in its memory)
    // ---- exchange: every wave's partial blocks -> LDS; the finalisers add the K parts in wave order (deterministic)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      char* dst = RED + ((wave * 2 + g) * 4) * 1024 + lane * 16;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) *(float4*)(dst + q4 * 1024) = make_float4(acc[g][4 * q4], acc[g][4 * q4 + 1], acc[g][4 * q4 + 2], acc[g][4 * q4 + 3]);
    }
    B3_BARRIER();
    B3S_STAMP(5);
    if (fjob) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll 1
      for (int kq = 0; kq < KQ; ++kq) {
        const char* src = RED + (((fmb * KQ + kq) * 2 + fng) * 4 + 2 * fhalf) * 1024 + lane * 16;
        const float4 a = *(const float4*)src, b = *(const float4*)(src + 1024);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
      if (f_img && fch < bch) {
        const bool own = mr[fng] >= 1 && mr[fng] <= R;  // the strip that owns the row writes it to HBM
        char* gdst = (char*)p.mid.p + (n * p.mid.sn + fiy * p.mid.sh + mx[fng] * p.mid.sw) + fch * 2;
        char* ldst = MID + (mr[fng] * B3S_MW + mx[fng] + 1) * MS + fch * 2;
        if constexpr (PRE) {
          v[0] += fb0.x; v[1] += fb0.y; v[2] += fb0.z; v[3] += fb0.w; v[4] += fb1.x; v[5] += fb1.y; v[6] += fb1.z; v[7] += fb1.w;
          const uint4 o = b3_pack8(v);
          if (own) *(uint4*)gdst = o;
          union { uint4 q; h16x8 h; } r;
          r.q = o;
          r.h = b3_relu8(r.h);
          *(uint4*)ldst = r.q;
        } else {
          const uint32_t w[4] = {tmk.x, tmk.y, tmk.z, tmk.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = h_lo(w[e]) > 0.f ? v[2 * e] : 0.f;
            v[2 * e + 1] = h_hi(w[e]) > 0.f ? v[2 * e + 1] : 0.f;
          }
          const uint4 o = b3_pack8(v);
          if (own) *(uint4*)gdst = o;
          *(uint4*)ldst = o;
        }
      }
    }
    B3S_STAMP(6);
    B3_BARRIER();
    B3S_STAMP(7);
    // ---- phase B
#pragma unroll 1
    for (int oi = 0; oi < ((p.s_dbg & 4) ? 0 : p.nout); ++oi) {
      const kout_ptr Ok = Ok0 + oi;
      if (oi > 0) b_start(Ok, wave);
      const float* const Obias = oi == 0 ? Ok->bias : nullptr;
      const char* const Oout = Ok->out.p; const int Oout_sn = Ok->out.sn, Oout_sh = Ok->out.sh, Oout_sw = Ok->out.sw;
      const char* const Oaux = Ok->aux.p; const int Oaux_sn = Ok->aux.sn, Oaux_sh = Ok->aux.sh, Oaux_sw = Ok->aux.sw;
      const char* const Ores = Ok->res.p; const int Ores_sn = Ok->res.sn, Ores_sh = Ok->res.sh, Ores_sw = Ok->res.sw;
      const int npb = Ok->npb, Co = Ok->Co;
      const int Oout_rem = Ok->out_rem, Ores_rem = Ok->res_rem;
      const bool has_aux = Oaux != nullptr, has_res = Ores != nullptr, has_rl = has_res && Ores_rem != 0;
      const int KP = kp_of(npb), njobs = npb * npg * KP;
      // epilogue of 8 channels of one pixel: requests first (in flight under the K loop), arithmetic + store later
      // (two operands per unit: forward = residual + its remainder plane, data gradient = mask source + accumulated gradient)
      struct Epi { uint4 a, b; float4 b0, b1; };
      auto epi_request = [&](const bool ok_, const int gy, const int gx, const int ch, Epi& E) {
        const bool ok = ok_ && ch < Co;
        E.a = E.b = make_uint4(0, 0, 0, 0);
        E.b0 = E.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (Obias != nullptr && ok) { E.b0 = *(const float4*)(Obias + ch); E.b1 = *(const float4*)(Obias + ch + 4); }
        if constexpr (PRE) {
          if (has_res) E.a = *(const uint4*)(ok ? Ores + (n * Ores_sn + gy * Ores_sh + gx * Ores_sw) + ch * 2 : zero);
          if (has_rl) E.b = *(const uint4*)(ok ? Ores + Ores_rem + (n * Ores_sn + gy * Ores_sh + gx * Ores_sw) + ch * 2 : zero);
        } else {
          if (has_aux) E.a = *(const uint4*)(ok ? Oaux + (n * Oaux_sn + gy * Oaux_sh + gx * Oaux_sw) + ch * 2 : zero);
          if (has_res) E.b = *(const uint4*)(ok ? Ores + (n * Ores_sn + gy * Ores_sh + gx * Ores_sw) + ch * 2 : zero);
        }
      };
      auto epi_finish = [&](const bool ok_, const int gy, const int gx, const int ch, float (&u)[8], const Epi& E) {
        if (!(ok_ && ch < Co)) return;
        u[0] += E.b0.x; u[1] += E.b0.y; u[2] += E.b0.z; u[3] += E.b0.w; u[4] += E.b1.x; u[5] += E.b1.y; u[6] += E.b1.z; u[7] += E.b1.w;
        const uint32_t wa[4] = {E.a.x, E.a.y, E.a.z, E.a.w}, wb[4] = {E.b.x, E.b.y, E.b.z, E.b.w};
        if constexpr (PRE) {  // (zeros were loaded where an operand is absent)
#pragma unroll
          for (int e = 0; e < 4; ++e) { u[2 * e] = (u[2 * e] + h_lo(wa[e])) + h_lo(wb[e]); u[2 * e + 1] = (u[2 * e + 1] + h_hi(wa[e])) + h_hi(wb[e]); }  // (residual, then its remainder: the tile kernel's order)
        } else {
          if (has_aux) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              u[2 * e] = h_lo(wa[e]) > 0.f ? u[2 * e] : 0.f;
              u[2 * e + 1] = h_hi(wa[e]) > 0.f ? u[2 * e + 1] : 0.f;
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { u[2 * e] += h_lo(wb[e]); u[2 * e + 1] += h_hi(wb[e]); }
        }
        const uint4 o16 = b3_pack8(u);
        char* dst = (char*)Oout + (n * Oout_sn + gy * Oout_sh + gx * Oout_sw) + ch * 2;
        *(uint4*)dst = o16;
        if (Oout_rem != 0) {  // what the rounding just dropped goes to the remainder plane: out_rem = rn16(v - out)
          const uint32_t wo[4] = {o16.x, o16.y, o16.z, o16.w};
          float dd[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { dd[2 * e] = u[2 * e] - h_lo(wo[e]); dd[2 * e + 1] = u[2 * e + 1] - h_hi(wo[e]); }
          *(uint4*)(dst + Oout_rem) = b3_pack8(dd);
        }
      };
      const int rounds = KP > 1 ? 1 : (njobs + B3S_NW - 1) / B3S_NW;  // (K parts: one job per wave, every wave meets the barrier)
#pragma unroll 1
      for (int r = 0; r < rounds; ++r) {
        const int job = wave + B3S_NW * r;
        const bool work = job < njobs;
        const int jc = min(job, njobs - 1), blk = jc / KP, kp = jc - blk * KP;
        const int pair = npg == 2 ? blk >> 1 : blk, pg = npg == 2 ? blk & 1 : 0;
        const int SK = (nks + KP - 1) / KP, k0 = min(kp * SK, nks - 1);
        const int o = 32 * pg + px, oc = min(o, NOP - 1);
        const int oy = b3_div(oc, p.s_dw), ox = oc - oy * W;
        const bool ev = work && o < NOP;
        const int ch0 = pair * 32 + 16 * kg, gy = y0 + oy;
        Epi E[2];
#pragma unroll
        for (int q8 = 0; q8 < 2; ++q8)
          if (KP == 1 || kp == q8) epi_request(ev, gy, ox, ch0 + 8 * q8, E[q8]);
        f32x16 c, cx;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = 0.f;
        const char* const mb = MID + (oy * B3S_MW + ox) * MS;
        KB.run<false>((const char*)(KT + 2 * k0 + kg), mb, mb, c, cx);
        if (r + 1 < rounds) b_start(Ok, job + B3S_NW);  // the next job's first fragments, under this one's epilogue (only where a run() follows: it drains them)
        if (KP == 1) {
#pragma unroll
          for (int q8 = 0; q8 < 2; ++q8) {
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = c[8 * q8 + e];
            epi_finish(ev, gy, ox, ch0 + 8 * q8, u, E[q8]);
          }
        } else {
          char* dst = RED + (wave * 4) * 1024 + lane * 16;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) *(float4*)(dst + q4 * 1024) = make_float4(c[4 * q4], c[4 * q4 + 1], c[4 * q4 + 2], c[4 * q4 + 3]);
          B3_BARRIER();
#pragma unroll
          for (int q8 = 0; q8 < 2; ++q8) {
            if (kp != q8) continue;
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.f;
#pragma unroll 1
            for (int k = 0; k < KP; ++k) {
              const char* src = RED + ((blk * KP + k) * 4 + 2 * q8) * 1024 + lane * 16;
              const float4 a4 = *(const float4*)src, b4 = *(const float4*)(src + 1024);
              u[0] += a4.x; u[1] += a4.y; u[2] += a4.z; u[3] += a4.w; u[4] += b4.x; u[5] += b4.y; u[6] += b4.z; u[7] += b4.w;
            }
            epi_finish(ev, gy, ox, ch0 + 8 * q8, u, E[q8]);
          }
          if (oi + 1 < p.nout) B3_BARRIER();  // (the next output's partial sums overwrite the exchange buffer)
        }
      }
    }
    B3S_STAMP(8);
  }
}

template <bool PRE>
// (128 registers per wave: two of its waves and one wave of the background weight-gradient kernel -- 256 registers -- share a SIMD.
//  With 256 the launch needs whole SIMDs and waits for the background kernel to END: measured, 1.5 ms of stall in a step)
__global__ __launch_bounds__(64 * B3S_NW, 4) void blk3s_kernel(B3P p) {
  blk3s_body<PRE>(p, (int)blockIdx.x, (int)gridDim.x, 0);
}
// two independent data-gradient problems in one launch (as blk3_pair_kernel): 128 + 128 workgroups of a decoder layer's posterior and
// prior Blocks fill the chip that either alone leaves half empty
__global__ __launch_bounds__(64 * B3S_NW, 4) void blk3s_pair_kernel(B3P pa, B3P pb, const int na) {
  (void)pa; (void)pb;
  const bool second = (int)blockIdx.x >= na;
  constexpr size_t off_b = (sizeof(B3P) + alignof(B3P) - 1) / alignof(B3P) * alignof(B3P);
  const int koff = second ? (int)off_b : 0;
  typedef const B3P __attribute__((address_space(4)))* kp_ptr;
  const B3P& p = *(const B3P*)(kp_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + koff);
  blk3s_body<false>(p, second ? (int)blockIdx.x - na : (int)blockIdx.x, second ? (int)gridDim.x - na : na, koff);
}

// ============================================================================= wide images, narrow channels (96x96, 192x192)
// The Blocks that carry 42 % of a step's FLOPs are HBM-bound: 32 / 64 channels in, a bottleneck of 8 / 16, 32 / 64 out, one input
// segment.  As two launches a 192x192 Block moves 264 MB (x, the bottleneck out and back in, x again for the residual, out) in
// 90 us; the floor of one launch is x + out + the bottleneck once = 170 MB = 27 us.  This instance STREAMS rows:
//   * a workgroup (8 waves) owns a column strip of 32 output pixels and walks down `rs` rows in bands of four; the input rows it
//     has seen stay in an LDS ring (four groups of four rows, 36 pixels wide), so only the strip's own border is fetched twice
//     (36 / 32 columns, (rs + 4) / rs rows), and the residual is read from that ring, not from HBM;
//   * the rows of band k + 3 are requested (LDS-DMA, untracked) at the top of band k: two to three bands of latency hiding; the one
//     counted wait of a band sits behind its phase-A MFMAs and in FRONT of its first store, so that no store is ever younger than
//     a request it has to outlive (a wait that must let stores pass would wait for them);
//   * phase A: the four NEW bottleneck rows of the band (136 pixels = 9 groups of 16) with v_mfma_f32_16x16x32 -- 16 rows are
//     exactly the bottleneck -- every wave keeps ALL of conv1's weights in registers for the whole launch (36 / 72 registers), a
//     B operand is one ds_read_b128 with an immediate offset; the bottleneck rows go into a 16-row LDS ring (+ HBM once);
//   * phase B: a wave owns one output row of 32 pixels for one 32-channel pair (32x32x16, weights in registers), epilogue from the
//     accumulators: bias, residual from the input ring / mask and accumulated gradient from HBM, two 16-byte stores per lane.
// Two barriers per band.  Forward and data gradient (one output) share the body like the tile instance.
#define B3R_NW 8
#define B3R_XC 36
#define B3R_MC 34
#define B3R_NG 4
#define B3R_MR 16
constexpr int b3r_grpb(int nch) { return ((4 * B3R_XC * (nch * 4 + 1) * 16 + 1023) / 1024) * 1024; }  // one ring group: four input rows, whole DMA instructions
constexpr int b3r_mrowb(int nb) { return B3R_MC * ((nb & 1) ? nb : nb + 1) * 16; }
constexpr size_t b3r_lds(int nch, int nb) { return (size_t)B3R_NG * b3r_grpb(nch) + (size_t)B3R_MR * b3r_mrowb(nb) + 1024; }

// RES: where the epilogue's added operand comes from -- 0 none, 1 the input ring (forward: the residual IS the input), 2 HBM
// (forward: another tensor; data gradient: the accumulated gradient)
template <bool PRE, int NCH, int NB, int RES>
__global__ __launch_bounds__(64 * B3R_NW, 2) void blk3r_kernel(B3P p) {
  __builtin_amdgcn_s_setprio(3);
  constexpr int G1 = NCH * 4 + 1, XS = G1 * 16, ROWB = B3R_XC * XS, GRPB = b3r_grpb(NCH), NINST = GRPB / 1024, IPW = (NINST + B3R_NW - 1) / B3R_NW;
  constexpr int NBS = (NB & 1) ? NB : NB + 1, MS = NBS * 16, MROWB = B3R_MC * MS;
  constexpr int KA = 9 * NCH, NKB = (9 * NB + 1) / 2;
  constexpr bool AUX = !PRE;  // (the data gradient masks its output with the forward input; the forward pass has no mask)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const X = smem;
  char* const MID = smem + B3R_NG * GRPB;
  char* const SCR = MID + B3R_MR * MROWB;  // destination of the DMA instructions a wave issues beyond the group's last (uniform count per wave)
  const char* const zero = (const char*)g_b3zero;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W, bch = p.b;
  // ---- work item: (image, row strip, column strip)
  const int item = blockIdx.x;
  const int sx = item % p.r_sx, t1 = item / p.r_sx, sy = t1 % p.r_sy, n = t1 / p.r_sy;
  const int x0 = sx * 32, Y0 = sy * p.r_rs, Yend = min(Y0 + p.r_rs, H), nbands = (Yend - Y0 + 3) >> 2;
  const BV3 sv = p.seg[0];
  const char* const xin = sv.p + n * sv.sn;
  // ---- the four input rows of ring group gi (rows Y0 - 2 + 4 gi ..), every wave IPW instructions
  auto dma_group = [&](const int gi) {
    const uint32_t gbase = (uint32_t)(uintptr_t)(lds_ptr)(X + (gi & (B3R_NG - 1)) * GRPB);
    const uint32_t sbase = (uint32_t)(uintptr_t)(lds_ptr)SCR;
    const int row0 = Y0 - 2 + 4 * gi;
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
      const int ii = wave + B3R_NW * t;
      const int sl = ii * 64 + lane;
      const int r = sl / (B3R_XC * G1), rem = sl - r * (B3R_XC * G1), px = rem / G1, g = rem - px * G1;
      const int iy = row0 + r, ix = x0 - 2 + px;
      const bool ok = ii < NINST && r < 4 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && g < G1 - 1;
      const char* src = ok ? xin + iy * sv.sh + ix * sv.sw + g * 16 : zero;
      b3_dma16(src, ii < NINST ? gbase + ii * 1024 : sbase);
    }
  };
  dma_group(0);
  dma_group(1);
  dma_group(2);  // (a strip of one band never reads it: harmless rows of the image, or zeros)
  // ---- weights: all of conv1 (16-row fragments, [tap][chunk]) and this wave's pair of conv2, in registers for the launch
  h16x8 Aw[KA], Bw[NKB];
  {
    const char* wa = p.wA16 + lane * 16;
#pragma unroll
    for (int s_ = 0; s_ < KA; ++s_) Aw[s_] = *(const h16x8*)(wa + s_ * 1024);
    const int pair_w = min(wave >> 2, p.o[0].npb - 1);
    const char* wb = p.o[0].w + (size_t)pair_w * NKB * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < NKB; ++i) Bw[i] = *(const h16x8*)(wb + i * 1024);
  }
  // ---- phase-A lane constants: bottleneck pixel q = 16 g + (lane & 15) of a band's new rows (34 wide), groups wave and wave + 8
  const int l16 = lane & 15, l4 = lane >> 4;
  int pr[2], pc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = min(16 * (wave + B3R_NW * j) + l16, 4 * B3R_MC - 1);
    pr[j] = q / B3R_MC; pc[j] = q - pr[j] * B3R_MC;
  }
  const int chA = 4 * l4;  // the four bottleneck channels this lane finalises
  float biasA[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (PRE) {
    if (p.biasA != nullptr && chA < bch) { const float4 b4 = *(const float4*)(p.biasA + chA); biasA[0] = b4.x; biasA[1] = b4.y; biasA[2] = b4.z; biasA[3] = b4.w; }
  }
  // ---- phase-B roles: wave = (pair, output row of the band); lane = (pixel of the row, 16-channel half)
  const B3Out& O = p.o[0];
  const int npb = O.npb, Co = O.Co;
  const int pairB = wave >> 2, rB = wave & 3;
  const bool jobB = pairB < npb;
  const int pxB = lane & 31, kgB = lane >> 5;
  const int ch0 = pairB * 32 + 16 * kgB;
  // K16-step i: this lane half reads 8-channel group u = 2 i + kg of (tap, bottleneck group): static per lane
  int koff[NKB], kty[NKB];
#pragma unroll
  for (int i = 0; i < NKB; ++i) {
    const int u = 2 * i + kgB, tp = min(u / NB, 8), gq = u - (u / NB) * NB;
    kty[i] = tp / 3;
    koff[i] = (pxB + tp % 3) * MS + gq * 16;
  }
  float biasB[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) biasB[e] = 0.f;
  if (jobB && O.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 16; e += 4)
      if (ch0 + e < Co) { const float4 b4 = *(const float4*)(O.bias + ch0 + e); biasB[e] = b4.x; biasB[e + 1] = b4.y; biasB[e + 2] = b4.z; biasB[e + 3] = b4.w; }
  }
  // Everything loaded so far is in its registers BEFORE the band loop, as far as the compiler is concerned too: a load it still
  // considers pending inside the loop would be waited for there with a small vmcnt -- a drain of the requests in flight, per band
  B3_VMWAIT();
#pragma unroll
  for (int s_ = 0; s_ < KA; ++s_) b3_pin(Aw[s_]);
#pragma unroll
  for (int i = 0; i < NKB; ++i) b3_pin(Bw[i]);
#pragma unroll
  for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(biasA[e]));
#pragma unroll
  for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(biasB[e]));
  B3_BARRIER();

  // The loads of a band that are not DMA (data gradient: the forward bottleneck under this lane's two bottleneck pixels, the forward
  // input under its output pixel, the accumulated gradient) are UNTRACKED too, requested at the top of the band in front of the
  // DMA request: the band's one counted wait -- "only the DMA request just issued may be in flight" -- covers them, and the compiler
  // never puts a wait of its own into the loop.
  b3_u32x2 tm[2];
  b3_u32x4 ea[2], er[2];
  tm[0] = tm[1] = (b3_u32x2){0u, 0u};
  ea[0] = ea[1] = er[0] = er[1] = (b3_u32x4){0u, 0u, 0u, 0u};
  auto band_loads = [&](const int m0, const int nrow, const int y) {
    if constexpr (!PRE) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + pr[j], col = x0 - 1 + pc[j];
        const bool ok = 16 * (wave + B3R_NW * j) + l16 < nrow * B3R_MC && (unsigned)m < (unsigned)H && (unsigned)col < (unsigned)W && chA < bch;
        b3_gload_v8(tm[j], ok ? p.mid_aux.p + (n * p.mid_aux.sn + m * p.mid_aux.sh + col * p.mid_aux.sw) + chA * 2 : zero);
      }
    }
    if constexpr (AUX || RES == 2) {
      const int ox = x0 + pxB;
#pragma unroll
      for (int q8 = 0; q8 < 2; ++q8) {
        const bool ok = jobB && y >= 0 && y < Yend && ox < W && ch0 + 8 * q8 < Co;
        if constexpr (AUX) b3_gload_v16(ea[q8], ok ? O.aux.p + (n * O.aux.sn + y * O.aux.sh + ox * O.aux.sw) + ch0 * 2 + 16 * q8 : zero);
        if constexpr (RES == 2) b3_gload_v16(er[q8], ok ? O.res.p + (n * O.res.sn + y * O.res.sh + ox * O.res.sw) + ch0 * 2 + 16 * q8 : zero);
      }
    }
  };
  // bottleneck rows m0 .. m0 + nrow - 1 (nrow = 2: the strip's first two; 4: a band's); WAIT: the counted wait of the band, behind
  // the MFMAs and in front of the first store (0: everything; 1: all but the DMA request just issued)
  auto phaseA = [&](const int m0, const int nrow, const int WAIT) {
    const int npix = nrow * B3R_MC;
    f32x4_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const int g = wave + B3R_NW * j;
      if (16 * g >= npix) continue;
      const int m = m0 + pr[j];
      // input row (m - 1 + ty) -> ring group / row in group
      int rowoff[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int rel = m - 1 + ty - (Y0 - 2);
        rowoff[ty] = ((rel >> 2) & (B3R_NG - 1)) * GRPB + (rel & 3) * ROWB;
      }
      const char* const xb = X + pc[j] * XS + l4 * 16;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int q = 0; q < NCH; ++q) {
            h16x8 bq = *(const h16x8*)(xb + rowoff[ty] + tx * XS + q * 64);
            if constexpr (PRE) bq = b3_relu8(bq);
            acc[j] = mfma_h16(Aw[(ty * 3 + tx) * NCH + q], bq, acc[j], 0, 0, 0);
          }
    }
    if (WAIT) b3_vmwait<IPW>(); else b3_vmwait<0>();
    if constexpr (!PRE) { b3_pin2(tm[0]); b3_pin2(tm[1]); }
    if constexpr (AUX) { b3_pin4(ea[0]); b3_pin4(ea[1]); }
    if constexpr (RES == 2) { b3_pin4(er[0]); b3_pin4(er[1]); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = wave + B3R_NW * j;
      if (16 * g + l16 >= npix || chA >= bch) continue;
      const int m = m0 + pr[j], col = x0 - 1 + pc[j];
      const bool in_img = (unsigned)m < (unsigned)H && (unsigned)col < (unsigned)W;
      const bool own = in_img && m >= Y0 && m < Yend && pc[j] >= 1 && pc[j] <= 32;
      float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
      uint2 o, ol;
      if constexpr (PRE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += biasA[e];
        o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]);
        union { uint2 u; s16x2v s[2]; } r;
        r.u = o;
        r.s[0] = __builtin_elementwise_max(r.s[0], (s16x2v){0, 0});
        r.s[1] = __builtin_elementwise_max(r.s[1], (s16x2v){0, 0});
        ol = r.u;
      } else {
        v[0] = h_lo(tm[j].x) > 0.f ? v[0] : 0.f; v[1] = h_hi(tm[j].x) > 0.f ? v[1] : 0.f;
        v[2] = h_lo(tm[j].y) > 0.f ? v[2] : 0.f; v[3] = h_hi(tm[j].y) > 0.f ? v[3] : 0.f;
        o.x = f2h_pk(v[0], v[1]); o.y = f2h_pk(v[2], v[3]);
        ol = o;
      }
      if (own) *(uint2*)((char*)p.mid.p + (n * p.mid.sn + m * p.mid.sh + col * p.mid.sw) + chA * 2) = o;
      *(uint2*)(MID + (m & (B3R_MR - 1)) * MROWB + pc[j] * MS + chA * 2) = in_img ? ol : make_uint2(0, 0);
    }
  };
  // ---- prologue: the strip's first two bottleneck rows
  band_loads(Y0 - 1, 2, -1);
  phaseA(Y0 - 1, 2, 0);
  for (int k = 0; k < nbands; ++k) {
    const int Y = Y0 + 4 * k, y = Y + rB;
    B3_BARRIER();            // everyone is through band k - 1: the group the next request overwrites is free, the two new rows above are visible
    band_loads(Y + 1, 4, y);
    dma_group(k + 3);        // (past the strip's last group: rows nobody reads -- the count of requests stays the same)
    phaseA(Y + 1, 4, 1);     // ... group k + 2 has landed (this wave's pieces): only the request just issued may still be in flight
    B3_BARRIER();            // the band's bottleneck rows, and everyone's pieces of group k + 2
    if (jobB) {
      const int ox = x0 + pxB;
      const bool ev = y < Yend && ox < W;
      int rowb[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) rowb[ty] = ((y - 1 + ty) & (B3R_MR - 1)) * MROWB;
      f32x16 c;
#pragma unroll
      for (int e = 0; e < 16; ++e) c[e] = biasB[e];
#pragma unroll
      for (int i = 0; i < NKB; ++i) {
        const int rb = kty[i] == 0 ? rowb[0] : (kty[i] == 1 ? rowb[1] : rowb[2]);
        const h16x8 bq = *(const h16x8*)(MID + rb + koff[i]);
        c = b3_mfma(Bw[i], bq, c);
      }
      if constexpr (RES == 1) {  // the residual is the input: pixel (y, ox) of the ring, raw (the ReLU was applied to the fragments)
        const int rel = y - (Y0 - 2);
        const char* rp = X + ((rel >> 2) & (B3R_NG - 1)) * GRPB + (rel & 3) * ROWB + (pxB + 2) * XS + (ch0 >> 3) * 16;
        er[0] = *(const b3_u32x4*)rp;
        er[1] = *(const b3_u32x4*)(rp + 16);
      }
#pragma unroll
      for (int q8 = 0; q8 < 2; ++q8) {
        if (!(ev && ch0 + 8 * q8 < Co)) continue;
        float u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = c[8 * q8 + e];
        if constexpr (AUX) {
          const uint32_t w[4] = {ea[q8].x, ea[q8].y, ea[q8].z, ea[q8].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            u[2 * e] = h_lo(w[e]) > 0.f ? u[2 * e] : 0.f;
            u[2 * e + 1] = h_hi(w[e]) > 0.f ? u[2 * e + 1] : 0.f;
          }
        }
        if constexpr (RES != 0) {
          const uint32_t w[4] = {er[q8].x, er[q8].y, er[q8].z, er[q8].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { u[2 * e] += h_lo(w[e]); u[2 * e + 1] += h_hi(w[e]); }
        }
        *(uint4*)((char*)O.out.p + (n * O.out.sn + y * O.out.sh + ox * O.out.sw) + ch0 * 2 + 16 * q8) = b3_pack8(u);
      }
    }
  }
  B3_VMWAIT();  // (requests past the strip's last band: nothing may land after the wave has ended)
}

// ----------------------------------------------------------------------------- host side
static bool b3_view(const cgen_view& v, int n, int h, int w, BV3& o) {
  o.p = (const char*)v.p; o.sn = o.sh = o.sw = 0;
  if (!v.p) return true;
  const int64_t ext = ((int64_t)n * v.sn + (int64_t)(h + 12 + 4) * v.sh + (int64_t)(w + B3_TW + 4) * v.sw + v.c + 64) * 2;
  if (ext >= ((int64_t)1 << 31) || v.sn < 0 || v.sh < 0 || v.sw < 0) return false;
  if (((uintptr_t)v.p % 16) || (v.sn * 2) % 16 || (v.sh * 2) % 16 || (v.sw * 2) % 16) return false;
  o.sn = (int)(v.sn * 2); o.sh = (int)(v.sh * 2); o.sw = (int)(v.sw * 2);
  return true;
}

static void b3_tiles(B3P& p, int th) {
  p.tiles_x = ceil_div(p.W, B3_TW); p.tiles_y = ceil_div(p.H, th);
  p.ntiles = p.N * p.tiles_x * p.tiles_y;
  p.d_tx = b3_mkdiv(p.tiles_x); p.d_ty = b3_mkdiv(p.tiles_y);
}

static size_t b3s_lds(const B3P& p) { return (size_t)p.s_kt_off + (128 + 18 * p.nch + 16) * 4; }  // the two K-axis tables end the image

static int b3_fill(const cgen_block3_args* a, B3P& p) {
  if (!a || a->dtype != CGEN_F16 || a->nseg < 1 || a->nseg > 3 || a->n <= 0 || a->h < 4 || a->w < 4) return 0;
  if (a->nout < 1 || a->nout > 2 || !a->w_a || !a->mid.p) return 0;
  memset(&p, 0, sizeof(p));
  p.N = a->n; p.H = a->h; p.W = a->w; p.nseg = a->nseg; p.nout = a->nout;
  int koff = 0;
  for (int s = 0; s < a->nseg; ++s) {
    if (!a->seg[s].p || a->seg[s].c <= 0 || !dma_clean(a->seg[s], 2)) return 0;
    if (!b3_view(a->seg[s], a->n, a->h, a->w, p.seg[s])) return 0;
    p.seg_koff[s] = koff;
    p.seg_c8[s] = (a->seg[s].c + 7) & ~7;
    koff += (a->seg[s].c + 31) / 32;
  }
  for (int s = a->nseg; s < 4; ++s) p.seg_koff[s] = koff;
  p.ctot8 = koff * 32;
  p.nch = koff;
  p.b = a->mid.c;
  static const int small_on = [] { const char* e = getenv("CGEN_BLK3S"); return e ? atoi(e) : 1; }();
  p.small = (small_on && a->w <= 14 && a->h <= 64) ? 1 : 0;
  if (p.b % 8 != 0 || p.b < 8 || p.b > (p.small ? 64 : 32)) return 0;
  p.nksB = (9 * p.b + 15) / 16;
  if (((uintptr_t)a->w_a % 16) || (a->bias_a && (uintptr_t)a->bias_a % 16)) return 0;
  p.wA = (const char*)a->w_a; p.biasA = a->bias_a;
  if (!b3_view(a->mid, a->n, a->h, a->w, p.mid) || !b3_view(a->mid_aux, a->n, a->h, a->w, p.mid_aux)) return 0;
  if (a->mid_aux.p && a->mid_aux.c != a->mid.c) return 0;
  for (int o = 0; o < a->nout; ++o) {
    const cgen_block3_out& s = a->o[o];
    B3Out& d = p.o[o];
    if (!s.out.p || !s.w || s.out.c % 8 != 0 || s.out.c < 8) return 0;
    if (((uintptr_t)s.w % 16) || (s.bias && (uintptr_t)s.bias % 16)) return 0;
    d.w = (const char*)s.w; d.bias = s.bias; d.Co = s.out.c; d.npb = (s.out.c + 31) / 32;
    if (s.out_rem < 0 || s.res1_rem < 0 || s.out_rem >= ((int64_t)1 << 31) || s.res1_rem >= ((int64_t)1 << 31) || (s.out_rem % 16) || (s.res1_rem % 16)) return 0;
    if ((s.out_rem || s.res1_rem) && (!a->pre_act || a->nout != 1)) return 0;  // (remainder planes: forward, single output)
    if (s.res1_rem && !s.res1.p) return 0;
    d.out_rem = (int)s.out_rem; d.res_rem = (int)s.res1_rem;
    if (!b3_view(s.out, a->n, a->h, a->w, d.out) || !b3_view(s.aux, a->n, a->h, a->w, d.aux) || !b3_view(s.res1, a->n, a->h, a->w, d.res)) return 0;
    if ((s.aux.p && s.aux.c != s.out.c) || (s.res1.p && s.res1.c != s.out.c)) return 0;
    if (o == 0 && s.out.c > 224 && !p.small) return 0;  // (lanes 8 .. 63 of the bias DMA instruction: 56 x 4 channels)
    if (o > 0 && s.bias) return 0;  // (only the first output's bias has an LDS copy: the forward pass has one output)
  }
  {  // row-streaming instance (blk3r)
    static const int rows_on = [] { const char* e = getenv("CGEN_BLK3R"); return e ? atoi(e) : 1; }();
    const int c0 = a->seg[0].c, co0 = a->o[0].out.c;
    p.rows = (rows_on && !p.small && a->w_a16 && a->nseg == 1 && a->nout == 1 && (c0 == 32 || c0 == 64) && (p.b == 8 || p.b == 16)
              && (co0 == 32 || co0 == 64) && a->w >= 48 && a->h >= 16 && !a->o[0].out_rem && !a->o[0].res1_rem
              && ((uintptr_t)a->w_a16 % 16) == 0) ? 1 : 0;
    if (p.rows) {
      p.wA16 = (const char*)a->w_a16;
      p.r_sx = ceil_div(p.W, 32);
      // strip height: whole bands of four rows; the cheapest (rounds of resident workgroups) x (rows + the prologue's worth)
      const int slots = 256;  // (one workgroup of eight waves per CU: 160-215 registers)
      long best = -1;
      for (int rs = 8; rs < p.H + 4; rs += 4) {
        const int rsc = rs > p.H ? ((p.H + 3) & ~3) : rs;
        const long wgs = (long)p.N * p.r_sx * ceil_div(p.H, rsc), cost = (long)ceil_div(wgs, slots) * (rsc + 16);
        if (best < 0 || cost < best) { best = cost; p.r_rs = rsc; }
      }
      { static const int rs_env = [] { const char* e = getenv("CGEN_BLK3R_RS"); return e ? atoi(e) : 0; }(); if (rs_env >= 4) p.r_rs = std::min((rs_env + 3) & ~3, (p.H + 3) & ~3); }
      p.r_sy = ceil_div(p.H, p.r_rs);
      p.ntiles = p.N * p.r_sx * p.r_sy;
      const cgen_view& rv = a->o[0].res1;
      p.r_res_tile = (a->pre_act && rv.p && rv.p == a->seg[0].p && rv.sn == a->seg[0].sn && rv.sh == a->seg[0].sh && rv.sw == a->seg[0].sw && co0 == c0) ? 1 : 0;
      p.stamps = nullptr;
      { static const int dbg = [] { const char* e = getenv("CGEN_BLK3R_DBG"); return e ? atoi(e) : 0; }(); p.s_dbg = dbg; }  // (unused)
      return 2;
    }
  }
  if (p.small) {
    int R = 64 / p.W - 2;
    if (R < 1) return 0;
    if (R > p.H) R = p.H;
    p.s_rows = R; p.s_strips = ceil_div(p.H, R); p.s_nmb = (p.b + 31) / 32;
    p.ntiles = p.N * p.s_strips;
    p.s_dxw = b3_mkdiv(p.W + 2); p.s_dw = b3_mkdiv(p.W); p.s_dstrips = b3_mkdiv(p.s_strips);
    for (int s = 0; s < a->nseg; ++s) {
      p.s_segk0[s] = p.seg_koff[s];
      p.s_segg[s] = 4 * ((a->seg[s].c + 31) / 32);
    }
    p.s_dg1 = b3_mkdiv(p.nch * 4 + 1); p.s_dnb = b3_mkdiv(p.b / 8);
    const int nb8 = p.b / 8, nbs = (nb8 & 1) ? nb8 : nb8 + 1;
    const int xbytes = (((R + 4) * (p.W + 2) * (p.nch * 4 + 1) + 63) / 64) * 1024;  // whole DMA instructions
    p.s_red_off = 0;  // the exchange buffer [wave][pixel group][16 floats][64 lanes] (64 KiB) shares the input tile's memory: the tile is dead after phase A
    p.s_mid_off = xbytes > 8 * 2 * 4096 ? xbytes : 8 * 2 * 4096;
    p.s_kt_off = p.s_mid_off + (((R + 2) * 16 * nbs * 16 + 1023) & ~1023);  // (B3S_MW = 16 pixels per row)
    if (2 * p.nksB + 30 > 128) return 0;      // (the phase-B table: 128 entries, read up to 13 steps past the end)
    if (b3s_lds(p) > 158 * 1024) return 0;
    for (int o = 0; o < a->nout; ++o)
      if (p.o[o].bias && ((uintptr_t)p.o[o].bias % 16)) return 0;
    { static unsigned long long* const stamps_env = [] { const char* e = getenv("CGEN_BLK3_STAMPS"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : (unsigned long long*)nullptr; }(); p.stamps = stamps_env; }
    { static const int dbg = [] { const char* e = getenv("CGEN_BLK3S_DBG"); return e ? atoi(e) : 0; }(); p.s_dbg = dbg; }  // (ablation: 1 no input DMA, 2 no phase-A K loop, 4 no phase B)
    return 1;
  }
  b3_tiles(p, 8);
  // (debug hook, read ONCE per process: the address of a device buffer for cycle stamps, tools/blk_stamps.py)
  static unsigned long long* const stamps_env = [] { const char* e = getenv("CGEN_BLK3_STAMPS"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : (unsigned long long*)nullptr; }();
  p.stamps = stamps_env;
  return 1;
}

struct B3Launch { int npg, grid, sm, th; size_t lds; };
// Ring depth, persistence and LDS size.  Two workgroups per CU (78 KB each) unless the launch has at most one tile per CU, which
// may take a whole CU's LDS; a ring one slot deeper than a tile needs lets the next tile's burst travel under this tile's work.
static B3Launch b3_plan(B3P& p, const int co_tiles = 0) {  // co_tiles: tiles of the problem sharing the launch (cgen_block3_pair)
  B3Launch L;
  int npb = 0;
  for (int o = 0; o < p.nout; ++o) npb = p.o[o].npb > npb ? p.o[o].npb : npb;
  L.npg = npb == 1 ? 1 : (npb == 2 ? 2 : 4);  // the wave split of phase B follows the WIDEST output
  const int nb = p.b / 8, nbs = (nb & 1) ? nb : nb + 1;
  static const int per_cu = [] { const char* e = getenv("CGEN_BLK3_PER_CU"); return e ? atoi(e) : 2; }();
  const int slots_wg = 256 * per_cu;
  p.wb_persist = (p.nout == 1 && nb <= 2 && p.nch <= 2 && ((L.npg == 1 && npb == 1) || (L.npg == 2 && npb == 2))) ? 1 : 0;
  static const int no_sm = [] { const char* e = getenv("CGEN_BLK3_NOSM"); return e ? atoi(e) : 0; }();
  L.sm = (p.wb_persist && !no_sm && p.nch == 1 && !p.o[0].out_rem && !p.o[0].res_rem) ? 1 : 0;  // (remainder planes: the any-chunk path's epilogue)
  // Tile height.  A launch is a whole number of rounds over the resident workgroups: twelve-row tiles where they save a round (48x48
  // at batch 32: 576 tiles = two rounds of eight rows, 384 = one of twelve); they exist for the wide-output, any-chunk instances
  L.th = 8;
  static const int th_env = [] { const char* e = getenv("CGEN_BLK3_TH"); return e ? atoi(e) : 0; }();
  if (L.npg == 4 && L.sm == 0 && nb <= 3 && th_env != 8) {  // (a 32-wide bottleneck with twelve rows spills registers: not built)
    const int t8 = p.ntiles, t12 = p.N * p.tiles_x * ceil_div(p.H, 12);
    const int r8 = ceil_div(t8, slots_wg), r12 = ceil_div(t12, slots_wg);
    if (th_env == 12 || (r12 * 3 < r8 * 2 + (r8 > 1 ? 1 : 0) && t12 > 128)) L.th = 12;  // (1.5x the work per round: worth it when it removes a round)
  }
  if (L.th != 8) b3_tiles(p, L.th);
  const int xb = b3_xbytes(L.th), nmp = b3_nmp(L.th);
  const int mid_bytes = nmp * nbs * 16;
  const int nmi = (nmp * nb + 63) / 64;  // DMA instructions of a mask tile
  const int extra = p.nch <= 2 ? ((nmp + 63) / 64) * 4096 : 0;  // second half of the partial-sum exchange (one-chunk tiles, and every streaming launch): 4 KiB per pixel group
  L.grid = p.ntiles < slots_wg ? p.ntiles : slots_wg;
  const int cap = (p.ntiles + co_tiles <= 256 ? 150 : (per_cu >= 3 ? 52 : 78)) * 1024;
  int ns;
  if (L.sm == 0) {  // two chunks ahead where three slots fit next to a second workgroup (<= 73 KB with eight rows and a 32-wide bottleneck)
    ns = (p.nch >= 3 && 3 * xb + mid_bytes + extra + 1024 <= cap) ? 3 : 2;
  } else {
    const int budget = cap - mid_bytes - extra - 2560 - (p.mid_aux.p ? 2 * nmi * 1024 : 0);
    const int want = p.nch + (p.ntiles > L.grid ? 1 : 0);
    static const int max_ns = [] { const char* e = getenv("CGEN_BLK3_MAXNS"); return e ? atoi(e) : 8; }();
    ns = budget / xb;
    if (ns > want) ns = want;
    if (ns > max_ns) ns = max_ns;
    if (ns < 2) ns = 2;
  }
  p.ns = ns;
  p.scratch_off = extra ? ns * xb + mid_bytes : 0;
  p.bias_off = ns * xb + mid_bytes + extra;
  p.tm_off = p.bias_off + 1024;  // (the bias copy is one whole DMA instruction: 1 KiB)
  p.tm_bytes = (L.sm > 0 && p.mid_aux.p) ? (nmi * 1024) : 0;  // whole DMA instructions (>= 180 pixels x b x 2 bytes)
  L.lds = (size_t)p.tm_off + 2 * (size_t)p.tm_bytes;
  return L;
}

template <bool PRE, int NB, int NPG, int SM, int TH, bool REM>
static void b3_launch_rem(const B3P& p, const B3Launch& L, hipStream_t st) {
  // (per call: the attribute is per device and per function, the call is a few hundred ns and thread-safe; a `static bool once` was neither)
  (void)hipFuncSetAttribute((const void*)blk3_kernel<PRE, NB, NPG, SM, TH, REM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((blk3_kernel<PRE, NB, NPG, SM, TH, REM>), dim3(L.grid), dim3(256), L.lds, st, p);
}
template <bool PRE, int NB, int NPG, int SM, int TH = 8>
static void b3_launch_inst(const B3P& p, const B3Launch& L, hipStream_t st) {
  if constexpr (PRE && SM == 0) {  // remainder planes of a residual trunk: forward, any-chunk instances (b3_fill / b3_plan see to that)
    if (p.o[0].out_rem || p.o[0].res_rem) return b3_launch_rem<PRE, NB, NPG, SM, TH, true>(p, L, st);
  }
  b3_launch_rem<PRE, NB, NPG, SM, TH, false>(p, L, st);
}
template <bool PRE, int NB>
static void b3_launch_nb(const B3P& p, const B3Launch& L, hipStream_t st) {
  if constexpr (NB <= 2) {  // streaming instances: a tile is one or two chunks, one output, the wave's pair is fixed
    if (L.sm == 1 && L.npg == 1) return b3_launch_inst<PRE, NB, 1, 1>(p, L, st);
    if (L.sm == 1 && L.npg == 2) return b3_launch_inst<PRE, NB, 2, 1>(p, L, st);
  }
  if (L.npg == 1) b3_launch_inst<PRE, NB, 1, 0>(p, L, st);
  else if (L.npg == 2) b3_launch_inst<PRE, NB, 2, 0>(p, L, st);
  else if (NB <= 3 && L.th == 12) { if constexpr (NB <= 3) b3_launch_inst<PRE, NB, 4, 0, 12>(p, L, st); }
  else b3_launch_inst<PRE, NB, 4, 0>(p, L, st);
}
template <bool PRE>
static void b3_launch_pre(const B3P& p, const B3Launch& L, hipStream_t st) {
  switch (p.b / 8) {
    case 1: b3_launch_nb<PRE, 1>(p, L, st); break;
    case 2: b3_launch_nb<PRE, 2>(p, L, st); break;
    case 3: b3_launch_nb<PRE, 3>(p, L, st); break;
    default: b3_launch_nb<PRE, 4>(p, L, st); break;
  }
}

// pair launch: the data-gradient instances without the streaming mode (both problems must plan to the SAME one)
template <int NB, int NPG, int TH = 8>
static void b3_pair_inst(const B3P& pa, const B3P& pb, const B3Launch& La, const B3Launch& Lb, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)blk3_pair_kernel<false, NB, NPG, 0, TH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((blk3_pair_kernel<false, NB, NPG, 0, TH, false>), dim3(La.grid + Lb.grid), dim3(256), std::max(La.lds, Lb.lds), st, pa, pb, La.grid);
}
template <int NB>
static void b3_pair_nb(const B3P& pa, const B3P& pb, const B3Launch& La, const B3Launch& Lb, hipStream_t st) {
  if (La.npg == 1) b3_pair_inst<NB, 1>(pa, pb, La, Lb, st);
  else if (La.npg == 2) b3_pair_inst<NB, 2>(pa, pb, La, Lb, st);
  else if (NB <= 3 && La.th == 12) { if constexpr (NB <= 3) b3_pair_inst<NB, 4, 12>(pa, pb, La, Lb, st); }
  else b3_pair_inst<NB, 4>(pa, pb, La, Lb, st);
}
// 0: the two problems cannot share a launch
static int b3_pair_plan(const cgen_block3_args* a, const cgen_block3_args* b, B3P& pa, B3P& pb, B3Launch& La, B3Launch& Lb) {
  if (!a || !b || a->pre_act || b->pre_act || !a->mid_aux.p || !b->mid_aux.p) return 0;
  if (!b3_fill(a, pa) || !b3_fill(b, pb)) return 0;
  if (pa.rows || pb.rows) return 0;
  if (pa.small || pb.small) {
    La.grid = pa.ntiles; Lb.grid = pb.ntiles;
    La.lds = b3s_lds(pa); Lb.lds = b3s_lds(pb);
    La.npg = Lb.npg = 0; La.sm = Lb.sm = 0; La.th = Lb.th = 0;
    return (pa.small && pb.small && !pa.o[0].out_rem && !pb.o[0].out_rem) ? 1 : 0;
  }
  const int ta = pa.ntiles, tb = pb.ntiles;  // (eight-row tiles: the cap decision below only needs "more than one tile per CU or not")
  La = b3_plan(pa, tb);
  Lb = b3_plan(pb, ta);
  if (La.sm || Lb.sm || La.npg != Lb.npg || La.th != Lb.th || pa.b != pb.b) return 0;
  if (pa.o[0].out_rem || pa.o[0].res_rem || pb.o[0].out_rem || pb.o[0].res_rem) return 0;
  return 1;
}

}  // namespace cgen

using namespace cgen;

extern "C" int cgen_block3_pair_supported(const cgen_block3_args* a, const cgen_block3_args* b) {
  B3P pa, pb;
  B3Launch La, Lb;
  return b3_pair_plan(a, b, pa, pb, La, Lb);
}

extern "C" int cgen_block3_pair(const cgen_block3_args* a, const cgen_block3_args* b, cgen_stream_t stream) {
  B3P pa, pb;
  B3Launch La, Lb;
  CGEN_REQUIRE(b3_pair_plan(a, b, pa, pb, La, Lb), "cgen_block3_pair: the two problems do not plan to the same data-gradient instance (ask cgen_block3_pair_supported first)");
  if (pa.small) {
    (void)hipFuncSetAttribute((const void*)blk3s_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(blk3s_pair_kernel, dim3(La.grid + Lb.grid), dim3(64 * B3S_NW), std::max(La.lds, Lb.lds), (hipStream_t)stream, pa, pb, La.grid);
    return check_launch("cgen_block3_pair(small)");
  }
  switch (pa.b / 8) {
    case 1: b3_pair_nb<1>(pa, pb, La, Lb, (hipStream_t)stream); break;
    case 2: b3_pair_nb<2>(pa, pb, La, Lb, (hipStream_t)stream); break;
    case 3: b3_pair_nb<3>(pa, pb, La, Lb, (hipStream_t)stream); break;
    default: b3_pair_nb<4>(pa, pb, La, Lb, (hipStream_t)stream); break;
  }
  static const bool trace = getenv("CGEN_CONV_TRACE") != nullptr;
  if (trace) fprintf(stderr, "blk3 pair[bwd] %dx%dx%d b %d | grids %d + %d, lds %zu, tile rows %d\n", a->n, a->h, a->w, pa.b, La.grid, Lb.grid, std::max(La.lds, Lb.lds), La.th);
  return check_launch("cgen_block3_pair");
}

extern "C" int cgen_block3_supported(const cgen_block3_args* a) {
  B3P p;
  return b3_fill(a, p);
}

extern "C" int cgen_block3(const cgen_block3_args* a, cgen_stream_t stream) {
  B3P p;
  CGEN_REQUIRE(b3_fill(a, p), "cgen_block3: shape / layout not served by the fused Block kernel (ask cgen_block3_supported first)");
  CGEN_REQUIRE((a->pre_act != 0) == (a->mid_aux.p == nullptr), "cgen_block3: pre_act = 1 is the forward pass (no mid_aux), pre_act = 0 the data gradient (mid_aux = the forward mid)");
  if (p.rows) {
    const int nch = a->seg[0].c / 32, nb = p.b / 8;
    const size_t lds = b3r_lds(nch, nb);
#define B3R_LAUNCH(PRE_, NCH_, NB_, RES_) do { (void)hipFuncSetAttribute((const void*)blk3r_kernel<PRE_, NCH_, NB_, RES_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((blk3r_kernel<PRE_, NCH_, NB_, RES_>), dim3(p.ntiles), dim3(64 * B3R_NW), lds, (hipStream_t)stream, p); } while (0)
#define B3R_RES(PRE_, NCH_, NB_) do { if (res_mode == 0) B3R_LAUNCH(PRE_, NCH_, NB_, 0); else if (res_mode == 1) B3R_LAUNCH(PRE_, NCH_, NB_, 1); else B3R_LAUNCH(PRE_, NCH_, NB_, 2); } while (0)
#define B3R_RES_BWD(NCH_, NB_) do { if (res_mode == 0) B3R_LAUNCH(false, NCH_, NB_, 0); else B3R_LAUNCH(false, NCH_, NB_, 2); } while (0)
    const int res_mode = !a->o[0].res1.p ? 0 : (p.r_res_tile ? 1 : 2);
    if (a->pre_act) { if (nch == 1) { if (nb == 1) B3R_RES(true, 1, 1); else B3R_RES(true, 1, 2); } else { if (nb == 1) B3R_RES(true, 2, 1); else B3R_RES(true, 2, 2); } }
    else { if (nch == 1) { if (nb == 1) B3R_RES_BWD(1, 1); else B3R_RES_BWD(1, 2); } else { if (nb == 1) B3R_RES_BWD(2, 1); else B3R_RES_BWD(2, 2); } }
#undef B3R_RES
#undef B3R_RES_BWD
#undef B3R_LAUNCH
    static const bool trace_r = getenv("CGEN_CONV_TRACE") != nullptr;
    if (trace_r) fprintf(stderr, "blk3r[%s] %dx%dx%d c %d b %d Co %d | strips %d x %d of %d rows, grid %d, lds %zu, residual from the tile %d\n", a->mid_aux.p ? "bwd" : "fwd", a->n, a->h, a->w,
                         a->seg[0].c, p.b, p.o[0].Co, p.r_sx, p.r_sy, p.r_rs, p.ntiles, lds, p.r_res_tile);
    return check_launch("cgen_block3(rows)");
  }
  if (p.small) {
    const size_t lds = b3s_lds(p);
    if (a->pre_act) {
      (void)hipFuncSetAttribute((const void*)blk3s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((blk3s_kernel<true>), dim3(p.ntiles), dim3(64 * B3S_NW), lds, (hipStream_t)stream, p);
    } else {
      (void)hipFuncSetAttribute((const void*)blk3s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL((blk3s_kernel<false>), dim3(p.ntiles), dim3(64 * B3S_NW), lds, (hipStream_t)stream, p);
    }
    static const bool trace_s = getenv("CGEN_CONV_TRACE") != nullptr;
    if (trace_s) fprintf(stderr, "blk3s[%s] %dx%dx%d ctot8 %d b %d Co %d nseg %d nout %d | %d rows per strip, grid %d, lds %zu\n", a->mid_aux.p ? "bwd" : "fwd", a->n, a->h, a->w, p.ctot8, p.b, p.o[0].Co, a->nseg, a->nout, p.s_rows, p.ntiles, lds);
    return check_launch("cgen_block3(small)");
  }
  const B3Launch L = b3_plan(p);
  if (a->pre_act) b3_launch_pre<true>(p, L, (hipStream_t)stream);
  else b3_launch_pre<false>(p, L, (hipStream_t)stream);
  static const bool trace = getenv("CGEN_CONV_TRACE") != nullptr;
  if (trace) fprintf(stderr, "blk3[%s] %dx%dx%d ctot8 %d b %d Co %d nseg %d nout %d | ring %d slots, lds %zu, grid %d, persist wb %d, tile rows %d\n", a->mid_aux.p ? "bwd" : "fwd", a->n, a->h, a->w, p.ctot8, p.b, p.o[0].Co, a->nseg, a->nout, p.ns, L.lds, L.grid, p.wb_persist, L.th);
  return check_launch("cgen_block3");
}
