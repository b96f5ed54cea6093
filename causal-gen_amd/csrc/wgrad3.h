// Streaming weight-gradient kernel, round 5 (csrc/wgrad3.hip): host-side plan + launch interface used by the C-ABI entry points
// in conv.hip (cgen_conv2d_wgrad, cgen_conv2d_wgrad_plan, cgen_conv2d_wgrad_batch_plan / _run).
#pragma once
#include "common.h"

namespace cgen {

struct W3Div { uint32_t mul, shift; };

// One operand tile in LDS (P: unshifted, wide; S: shifted by the taps, narrow).  Pixels are laid out linearly (row-major over
// rows x rowpx) at `sp` bytes per pixel; a DMA instruction fills `ppi` whole pixels (gpp 16-byte groups each).
struct W3Op {
  int c8;      // channels at 8-channel granularity (all segments side by side)
  int sp;      // LDS pixel stride, bytes
  int gpp;     // sp / 16
  int ppi;     // pixels per DMA instruction (64 / gpp)
  int rowpx;   // tile row length, pixels (tw, or tw + 2 with a halo)
  int rows;    // tile rows
  int npx;     // rows * rowpx
  int ninstr;  // DMA instructions per tile
  int halo;    // 0 / 1
  int bytes;   // LDS bytes reserved for the tile (ninstr * ppi * sp)
  W3Div d_row; // division by rowpx
  W3Div d_gpp; // division by gpp
};

struct Wg3P {
  int N, H, W, ks, taps, act;
  int x_is_s;   // 1: S = act(X) (n = tap * cs8 + ci), P = grad_out (rows = co); 0: S = grad_out (n = flipped tap * cs8 + co), P = act(X) (rows = ci)
  int th, tw, tiles_x, tiles_y, ntiles, tps, nsplit;
  int ksteps, kst_rows;  // K16-steps per tile; tile rows per K-step (1: tw = 16, 2: tw = 8)
  W3Div d_tx, d_ty;
  W3Op P, S;
  int pwin_c, n_pwin, MP, NS;  // channels per P window (multiple of 32), windows, P fragments per window, S fragments
  int WM, WN, WK;              // wave grid: P-fragment blocks x S-fragment blocks x K split (WM * WN * WK = 4)
  int variant;                 // MPW * 8 + NSW
  int n_swin, pad1;            // S fragment windows (WN * NSW fragments each): low-resolution layers whose tap-packed axis exceeds 16 fragments
  int nslot, slot_bytes;       // tile ring
  int layout;                  // partial layout: 0 = [split][co][tap][ci], 1 = [split][ci][flipped tap][co]
  int co, ci_total;
  int nsegx;
  View segx[CGEN_MAX_SEG];
  int x_k8[CGEN_MAX_SEG + 1];  // 8-granular start of each X segment in the concatenated axis ([nseg] = total)
  int x_off[CGEN_MAX_SEG];     // real channel offset of each X segment in the OIHW gradient
  View g;
  float* pw;
  float* pb;
  unsigned long long* stamps;
  unsigned long long* blocklog;      // CGEN_WG3_BLOCKLOG: per-block (problem, split, start, end) log of the packed launch, or null
  unsigned long long blocklog_cap;   // entries the log holds
  int dbg, cx8;  // cx8: X channels at 8-channel granularity, all segments
};

struct Wg3Plan {
  Wg3P q;          // everything but pw / pb filled in
  int nsplit_total;  // = nsplit: what the caller sizes the partial buffers with
  int nblocks;       // workgroups: nsplit * n_pwin * n_swin
  size_t lds;
  long block_bytes;  // HBM bytes one workgroup moves (sort key of the packed launch)
};

// false: the problem is not served by this kernel (the caller falls back to the older kernels)
bool wg3_plan(const cgen_wgrad_args* a, Wg3Plan& g);
void wg3_launch_single(const Wg3Plan& g, hipStream_t st);
// packed form: `probs` is a device table of Wg3P, blocks[b] = {problem, split, P window, S window}
void wg3_launch_mega(const Wg3P* probs_dev, const int4* blocks_dev, int nblocks, int grid, size_t lds, hipStream_t st);

}  // namespace cgen
