/* cgen_hip.h -- C ABI of libcgen_hip.so: the MI355X (gfx950) kernels behind the HVAE image-mechanism /
 * counterfactual hot path of biomedia-mira/causal-gen.
 *
 * The reference has no FFI layer: its L0 is stock ATen.  Each entry point below replaces the
 * ATen op class(es) the reference dispatches at the cited src/ file:line; the host-side mirror of the
 * reference's Python surface (causal-gen_amd/vae.py, dscm.py, dmol.py) is the only caller.
 *
 * Conventions
 *   - Activations are NHWC *views*: channel stride 1, arbitrary (n,h,w) strides in ELEMENTS.  A channel slice
 *     or a [:res,:res] crop of a bigger tensor is therefore a view, and torch.cat along C is never materialised
 *     (multi-segment inputs).  dtype: CGEN_F32 (exact path: f32 MFMA 16x16x4, bit-level fmaf chains) or
 *     CGEN_F16 (IEEE binary16 storage + v_mfma_f32_16x16x32_f16 / 32x32x16_f16, f32 accumulate; activation GRADIENTS are carried
 *     times a power-of-two loss scale chosen by the caller, and the two kernels that turn them into f32 parameter gradients --
 *     cgen_wgrad_reduce via cgen_wred_desc.unscale, cgen_batch_reduce via its `unscale` argument -- multiply by its inverse).
 *     Parameters/gradients are always f32.
 *   - Ownership: the caller owns every buffer including workspaces; the library never allocates, frees or
 *     synchronises, and keeps no mutable global state (deepcopy / fork / hipGraph-capture safe).
 *   - Every call only enqueues work on `stream` and returns 0, or a negative cgen_status (message via
 *     cgen_last_error()).  Numerical NaNs are data, not errors (trainer.py:71-85 tests for them).
 *   - Thread-safe and re-entrant; cgen_last_error() is thread-local.
 */
#ifndef CGEN_HIP_H
#define CGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cgen_stream_t; /* hipStream_t */

enum cgen_status { CGEN_OK = 0, CGEN_EINVAL = -1, CGEN_ELAUNCH = -2, CGEN_EUNSUPPORTED = -3 };
enum cgen_dtype { CGEN_F32 = 0, CGEN_F16 = 1,
                  /* cgen_conv2d only (ABI 405): CGEN_F32 tensors and weight image, but the tiled kernel forms every product from
                   * SPLIT binary16 operands -- v = hi + lo, three v_mfma_f32_16x16x32_f16 per K-step (hi*hi + hi*lo + lo*hi), f32
                   * accumulate: ~2^-21 relative per product instead of binary16's 2^-11, at a fifth of the f32 MFMA's cycles.  The
                   * inference passes of the f32 engine use it (counterfactual pixels within 1e-3 of the reference, north_star);
                   * shapes the tiled kernel does not take run the exact f32 kernels.  Operands must lie inside binary16's range. */
                  CGEN_F32S = 2 };
enum cgen_act { CGEN_ACT_NONE = 0, CGEN_ACT_RELU = 1, CGEN_ACT_GELU = 2 };

/* NHWC strided view; strides in elements; p == NULL means "absent".
 * cpad (optional, 0 = none): the caller guarantees that channels [c, cpad) of every pixel are readable and hold
 * finite values (zeros); lets the tiled kernels fetch whole 16-byte groups of a ragged-width tensor by LDS-DMA.
 * On an OUTPUT view of cgen_conv2d, cpad > c asks the kernel to write zeros into channels [c, cpad) as well, so that a
 * ragged-width result (e.g. the 4-channel bottleneck of a 16-wide Block) is itself DMA-clean for its consumers. */
typedef struct cgen_view {
  void* p;
  int64_t sn, sh, sw;
  int32_t c;
  int32_t cpad;
} cgen_view;

#define CGEN_MAX_SEG 4

/* ABI version of this header.  cgen_version() of the loaded library must equal it (causal-gen_amd/_lib.py checks): struct layouts,
 * enum values and signatures are only compatible within one version.  cgen_h16_format(): the 16-bit storage format the library was
 * BUILT for -- 0 = IEEE binary16 (default), 1 = bfloat16 (-DCGEN_H16_BF16, an A/B build); CGEN_F16 tensors must be in that format. */
#define CGEN_ABI_VERSION 407
int cgen_version(void);
int cgen_h16_format(void);
const char* cgen_last_error(void);

/* ------------------------------------------------------------------ convolution (K1/K2/K3/K4/K5/K8/K9)
 * out = (bias + conv_{KSxKS, stride 1, pad KS/2}( act(cat_C(seg[0..nseg))) )) * act'(aux) + res1 + res2
 * Replaces aten::convolution (+ the gelu/relu before it, the torch.cat feeding it, the residual adds after it)
 * at vae.py:53-55,61-67,71 (Block), :104-110 (stem), :165,167 (z_proj, z_feat_proj), :176,188 (cat), :78,292-294
 * (adds), :325-333 (likelihood heads), dmol.py:223.  The SAME entry point is the data-gradient kernel:
 * seg = grad_out, weight = the "dgrad image" from cgen_weight_prep, dact/aux = the forward activation and
 * its input, res1 = out (accumulate).  res1/res2 may alias out.
 * weight: image built by cgen_weight_prep: [ceil16(Co)][krow], column k = tap * C8 + c with C8 = sum_s ceil8(C_s)
 * (segments side by side at 8-channel granularity), krow = ceil32(KS*KS*C8) + 32, in `dtype`, zero padded.
 * Remainder planes (CGEN_F16 only; 0 = absent): the residual trunk of the HVAE (vae.py:78, 253, 292-294: ~100 consecutive
 * `h = h + f(h)` updates) is carried as a PAIR of binary16 tensors, value = hi + rem with |rem| <= ulp(hi)/2, so that the sum
 * keeps ~22 significant bits although every conv still reads 16-bit operands (hi alone).  `res1_rem` / `out_rem` are BYTE
 * offsets from res1.p / out.p to a plane of identical layout: the epilogue adds res1 + res1_rem in f32, stores
 * out = rn16(v) and, when out_rem != 0, out_rem = rn16(v - out). */
typedef struct cgen_conv_args {
  int32_t dtype, n, h, w, ks, nseg, act, dact;
  cgen_view seg[CGEN_MAX_SEG];
  const void* weight;
  const float* bias; /* [Co] or NULL */
  cgen_view out;     /* out.c = Co */
  cgen_view aux, res1, res2;
  int64_t out_rem, res1_rem;
} cgen_conv_args;
int cgen_conv2d(const cgen_conv_args* a, cgen_stream_t stream);
/* Two independent convolutions in ONE launch where both are small-image problems of the same kernel instance (f16, 1x1 or 3x3, the
 * <= ~6000-pixel batches of the <= 12x12 layers: conv_smallp, whose launches are latency-bound whatever their size).  In the reference's
 * graph: the data gradients of the posterior and the prior Block's convs of a DecoderBlock (vae.py:240-301), which autograd runs back to
 * back.  Bit-identical to two cgen_conv2d calls; the caller guarantees that neither reads or accumulates into what the other writes. */
int cgen_conv2d_pair_supported(const cgen_conv_args* a, const cgen_conv_args* b);
int cgen_conv2d_pair(const cgen_conv_args* a, const cgen_conv_args* b, cgen_stream_t stream);

/* ------------------------------------------------------------------ fused "light" Block (csrc/block.hip; binary16)
 * Block.forward with version == "light" (vae.py:49-56, 60-71, 73-84) as ONE launch, the bottleneck tensor resident in LDS:
 *   pre_act = 1 (forward):   mid = bias_a + conv3x3(relu(cat_C(seg)))          (pre-activation bottleneck, interior pixels written once)
 *                            o[k].out = o[k].bias + conv3x3(relu(mid)) + o[k].res1
 *   pre_act = 0 (data gradient: aten::convolution_backward x2 + threshold_backward x2, the input part; mid_aux = the forward `mid`):
 *                            mid = conv3x3(seg[0] = grad_out; w_a = fragment image of conv2's dgrad) * relu'(mid_aux)
 *                            o[k].out = conv3x3(mid; o[k].w = fragment image of conv1's dgrad w.r.t. segment k) * relu'(o[k].aux) + o[k].res1
 * nout = 2 serves a Block with two differentiable input segments (the posterior Block: h and the encoder activation).
 * Weight images are FRAGMENT-ORDERED (one contiguous KiB per wave load), built by cgen_weight_prep modes 2-5:
 *   w_a   [sum_s ceil(seg[s].c / 32) chunks][18 K16-steps kk: channel half kk / 9 (channels 16 (kk / 9) .. + 16), tap = kk % 9][64 lanes][8]
 *         (every segment occupies whole 32-channel chunks of the K axis, zero padded; a wave's nine fragments are contiguous)
 *   o[].w [ceil(Co / 32) pairs][ceil(9 b / 16) K16-steps over k = tap * b + c][64 lanes][8]
 *   lane l of a fragment holds row (l & 31) -> channel 16 ((r >> 2) & 1) + (r & 3) + 4 (r >> 3) of the 32-row block, k = 8 (l >> 5) .. + 8.
 * Served: mid.c in {8, 16, 24, 32}, out.c a multiple of 8, up to 3 input segments (each DMA-clean: channels a multiple of 8 or
 * zero padded, cgen_view.cpad), 16-byte aligned views below 2^31 bytes; cgen_block3_supported() answers without launching. */
typedef struct cgen_block3_out {
  const void* w;
  const float* bias;
  cgen_view out, aux, res1;
  int64_t out_rem, res1_rem; /* remainder planes of a residual trunk, as in cgen_conv_args (byte offsets from out.p / res1.p, 0 = none) */
} cgen_block3_out;
typedef struct cgen_block3_args {
  int32_t dtype, n, h, w, nseg, nout, pre_act, reserved;
  cgen_view seg[CGEN_MAX_SEG];
  const void* w_a;
  const float* bias_a;
  cgen_view mid, mid_aux;
  cgen_block3_out o[2];
  const void* w_a16; /* optional (NULL: absent): the first conv's weights as the 16-row image of cgen_weight_prep modes 6 / 7, which the
                      * row-streaming instance (wide images, one input segment of 32 or 64 channels, bottleneck <= 16: the 96x96 and
                      * 192x192 Blocks) reads; without it those shapes run on the tile instance */
} cgen_block3_args;
int cgen_block3_supported(const cgen_block3_args* a);
int cgen_block3(const cgen_block3_args* a, cgen_stream_t stream);
/* Two independent DATA-GRADIENT problems (pre_act = 0) in ONE launch: the backward of a decoder layer's posterior and prior Blocks
 * (vae.py:240-301: both hang off the layer's reparameterisation, neither reads what the other writes).  Each is 100-400 workgroups on
 * 512 resident slots; one launch runs them side by side without a second queue.  Served when both plan to the same kernel instance
 * (same bottleneck width, same widest-output class, no streaming mode, no remainder planes): ask _supported first.  The caller
 * guarantees that neither problem reads or accumulates into a tensor the other one writes. */
int cgen_block3_pair_supported(const cgen_block3_args* a, const cgen_block3_args* b);
int cgen_block3_pair(const cgen_block3_args* a, const cgen_block3_args* b, cgen_stream_t stream);

/* ------------------------------------------------------------------ fused DEFAULT Block (csrc/block4.hip; binary16)
 * Block.forward of the non-"light" version (vae.py:57-71, 73-84: GELU -> 1x1 -> GELU -> 3x3 -> GELU -> 3x3 -> GELU -> 1x1) as ONE launch,
 * the three bottleneck tensors' activated tiles resident in LDS:
 *   fwd = 1:  mid[0] = bias[0] + conv1x1(gelu(cat_C(seg)); w[0]),  mid[1] = bias[1] + conv3x3(gelu(mid[0]); w[1]),
 *             mid[2] = bias[2] + conv3x3(gelu(mid[1]); w[2]),       o[0].out = o[0].bias + conv1x1(gelu(mid[2]); o[0].w) + o[0].res1
 *             (mid[]: pre-activations, written once -- the weight gradients and the backward pass read them; mid_aux absent)
 *   fwd = 0 (data gradient: aten::convolution_backward x4 + gelu_backward x4, the input part; seg[0] = grad_out, nseg = 1;
 *             mid_aux[0..2] = the forward mid[2], mid[1], mid[0]):
 *             mid[0] = conv1x1(seg[0]; w[0] = image of the last conv's transpose) * gelu'(mid_aux[0])       (gradient w.r.t. forward mid[2])
 *             mid[1] = conv3x3(mid[0]; w[1] = flipped transpose of the second 3x3) * gelu'(mid_aux[1])      (... forward mid[1])
 *             mid[2] = conv3x3(mid[1]; w[2] = flipped transpose of the first 3x3) * gelu'(mid_aux[2])       (... forward mid[0])
 *             o[k].out = conv1x1(mid[2]; o[k].w = transpose of the first conv w.r.t. segment k) * gelu'(o[k].aux) + o[k].res1
 *             for up to three differentiable input segments (the posterior Block: h and the encoder activation).
 * Weight images are FRAGMENT-ORDERED (1 KiB = one v_mfma_f32_32x32x16 A operand: lane l holds fragment row r = l & 31, k-group kg = l >> 5),
 * built by cgen_weight_prep modes 8-13; G = ceil(b / 16).  Row permutation, with kr = (r >> 2) & 1 and j = (r & 3) + 4 (r >> 3):
 * o[].w rows carry channel 16 kr + j of their 32-row block (a lane of the accumulator tile ends up with 16 consecutive channels);
 * w[0..2] rows carry channel HW kr + j for j < HW (zero rows otherwise), HW = half of the block's channels after rounding b up to 8
 * (4, 8, 12 or 16): every lane of the post-operation works on real channels whatever the bottleneck width.
 *   w[0]        [ceil(b / 32) row blocks][sum_s ceil(seg[s].c / 32) chunks][2 steps][64 lanes][8]: lane (row, kg), step s, element e
 *               multiplies input channel 32 chunk + 16 s + 8 kg + e (segments in whole chunks, zero padded)
 *   w[1], w[2]  [ceil(b / 32)][9 taps][G][64][8]: k = 16 g + 8 kg + e
 *   o[].w       [ceil(Co / 32)][G][64][8]
 * Served: b <= 48 or 57..64, up to 3 input segments, output widths multiples of 8 (<= 256 forward), every view 16-byte aligned, below 2^31
 * bytes, ragged channel counts zero padded to 8 (cgen_view.cpad); any image size (8 x 16 output tiles).  mid[k].c = mid_aux[k].c = b. */
typedef struct cgen_block4_args {
  int32_t dtype, n, h, w, nseg, nout, fwd, b;
  cgen_view seg[CGEN_MAX_SEG];
  const void* wimg[3]; /* w[0..2] below */
  const float* bias[3];
  cgen_view mid[3], mid_aux[3];
  cgen_block3_out o[3];
} cgen_block4_args;
int cgen_block4_supported(const cgen_block4_args* a);
int cgen_block4(const cgen_block4_args* a, cgen_stream_t stream);
/* Two independent DATA-GRADIENT problems (fwd = 0) of the same bottleneck class (ceil(b / 8) equal) in ONE launch: the backward of a
 * decoder layer's posterior and prior Blocks (vae.py:240-301).  Bit-identical to two cgen_block4 calls; the caller guarantees that
 * neither problem reads or accumulates into a tensor the other one writes. */
int cgen_block4_pair_supported(const cgen_block4_args* a, const cgen_block4_args* b);
int cgen_block4_pair(const cgen_block4_args* a, const cgen_block4_args* b, cgen_stream_t stream);

/* Weight gradient (aten::convolution_backward, weight/bias part) as split-K partials:
 *   partial_w[split][Co][KS*KS][Ci_total] (f32), partial_b[split][Co] (f32, may be NULL)
 * with Ci_total = sum_s seg[s].c and nsplit = cgen_conv2d_wgrad_plan(args) (call it with the views filled in).  *tiled_out says
 * which kernel will serve the call and in which layout it leaves partial_w: 0 = generic kernel, 1 = tiled 16-bit (CGEN_F16) kernel,
 * 2 = streaming kernel of round 5 (csrc/wgrad3.hip) -- all three the layout above --, 3 = streaming kernel with grad_out as its
 * shifted operand: partial_w[split][Ci_total][KS*KS - 1 - tap][Co] (pass cgen_wred_desc.layout = 1 to cgen_wgrad_reduce).
 * Deterministic: partials are summed in a fixed order by cgen_wgrad_reduce. */
typedef struct cgen_wgrad_args {
  int32_t dtype, n, h, w, ks, nseg, act, nsplit;
  cgen_view seg[CGEN_MAX_SEG];
  cgen_view gout; /* gout.c = Co */
  float* partial_w;
  float* partial_b;
} cgen_wgrad_args;
int cgen_conv2d_wgrad_plan(const cgen_wgrad_args* a, int32_t* tiled_out);
/* Horizontally batched weight gradients.  The weight-gradient launches of a step are independent of each other; most of
 * them fill a fraction of the chip.  `plan` packs every problem the tiled 16-bit (CGEN_F16) kernel serves (eligible[i] = 1; the others
 * go through cgen_conv2d_wgrad) into a host blob: a table of problems followed by one {problem, split, window, co range}
 * record per workgroup, and describes one launch per kernel variant.  Call it with blob_host = NULL to get the sizes.
 * The caller copies the blob to the device ONCE (addresses are stable across steps: the arena is deterministic) and
 * `run` then issues n_launches kernels instead of `count`.  nsplit / partial buffers exactly as cgen_conv2d_wgrad.
 * max_workgroups > 0 caps each launch's grid (a resident set of workgroups walks the records): a batch issued on a side
 * stream in the background of the backward chain must leave CUs free for the chain's kernels; 0 = one workgroup per record. */
typedef struct {
  int32_t ncf, ks, lds_bytes, nblocks;
  int64_t blocks_offset; /* byte offset of this launch's workgroup records inside the blob */
} cgen_wgrad_batch_launch;
int cgen_conv2d_wgrad_batch_plan(const cgen_wgrad_args* args, int32_t count, void* blob_host, int64_t capacity, int64_t* blob_bytes,
                                 cgen_wgrad_batch_launch* launches, int32_t max_launches, int32_t* n_launches, int32_t* eligible);
int cgen_conv2d_wgrad_batch_run(const void* blob_dev, const cgen_wgrad_batch_launch* launches, int32_t n_launches, int32_t max_workgroups,
                               cgen_stream_t);
int cgen_conv2d_wgrad(const cgen_wgrad_args* a, cgen_stream_t stream);

/* Multi-tensor descriptor tables (device memory, built once by the host).  One launch serves every conv site. */
typedef struct cgen_wprep_desc { /* OIHW f32 parameter -> forward image or dgrad image */
  const float* src;
  void* dst;
  int32_t co, ci_total, ks, mode; /* mode 0: fwd image, 1: dgrad image of segment [seg_off, seg_off+seg_c[0]);
                                   * fragment-ordered images of cgen_block3 (src is always the OIHW parameter [co][ci_total][3][3]):
                                   * (1 KiB fragments in v_mfma_f32_32x32x16 A-operand lane order; phase-A images [chunk][channel half][tap])
                                   * 2: w_a of the forward pass (conv1: rows = co, K = the segments' channels),
                                   * 3: w_a of the data gradient (conv2: rows = ci_total, K = co, taps flipped),
                                   * 4: o[].w of the forward pass (conv2: rows = co, K = tap * ci_total + c; k_pad = K16-steps per pair),
                                   * 5: o[].w of the data gradient w.r.t. segment [seg_off, seg_off + seg_c[0]) (conv1: rows = the
                                   *    segment's channels, K = tap * co + c, taps flipped; k_pad = K16-steps per pair)
                                   * 6 / 7: w_a16 of the forward pass / the data gradient -- 1 KiB fragments in v_mfma_f32_16x16x32 A-operand lane
                                   *    order, [tap 0..8][32-channel chunk q][64 lanes][8]: lane l holds row (l & 15) (bottleneck channel, <= 16),
                                   *    k = 32 q + 8 (l >> 4) .. + 8 (6: conv1's input channel; 7: conv2's output channel, taps flipped)
                                   * 8-13: images of cgen_block4 (layouts there): 8 / 9 = w[0] forward (src = the first 1x1 [b][ci_total]; k_pad = 2 x chunks)
                                   *    / data gradient (src = the last 1x1 [co][b]: rows = ci_total, K = co), 10 / 11 = a 3x3 forward / flipped
                                   *    transpose (k_pad = G), 12 = o[].w forward (src = the last 1x1), 13 = o[].w of the data gradient w.r.t. segment
                                   *    [seg_off, seg_off + seg_c[0]) (src = the first 1x1; k_pad = G) */
  int32_t nseg, seg_off;
  int32_t seg_c[CGEN_MAX_SEG];
  int32_t dtype, rows_pad, k_pad, reserved; /* k_pad = krow of the image */
  int64_t numel; /* rows_pad * k_pad */
} cgen_wprep_desc;
int cgen_weight_prep(const cgen_wprep_desc* descs_dev, const int32_t* chunk_site_dev, const int32_t* chunk_index_dev,
                     int32_t nchunks, cgen_stream_t stream);

typedef struct cgen_wred_desc { /* split-K partials -> OIHW f32 gradient (+bias gradient) */
  const float* partial_w;
  const float* partial_b;
  float* grad_w; /* [Co][Ci][KS][KS] */
  float* grad_b; /* [Co] or NULL */
  int32_t co, ci_total, ks, nsplit;
  int32_t accumulate;
  float unscale; /* the sums are multiplied by this (1 / loss scale of the f16 engine); 0 means 1 */
  int32_t layout; /* of partial_w, as reported by cgen_conv2d_wgrad_plan: 0 = [split][Co][tap][Ci], 1 = [split][Ci][KS*KS-1-tap][Co] */
  int32_t reserved;
  int64_t numel; /* co*ci_total*ks*ks + co */
} cgen_wred_desc;
int cgen_wgrad_reduce(const cgen_wred_desc* descs_dev, const int32_t* chunk_site_dev, const int32_t* chunk_index_dev,
                      int32_t nchunks, cgen_stream_t stream);

/* ------------------------------------------------------------------ element-wise / data movement
 * aten::avg_pool2d fwd/bwd (vae.py:79-83), upsample_nearest2d (+ learned per-res bias, vae.py:251-262),
 * F.pad / slicing / clone (vae.py:131-133, 241), repeat of bias[1] (vae.py:233), pa_sto scaling (vae.py:244-247),
 * layout + u8->[-1,1] preprocessing (trainer.py:16-21). */
int cgen_avgpool_fwd(int32_t dtype, int32_t n, int32_t ho, int32_t wo, int32_t d, cgen_view in, cgen_view out, cgen_stream_t);
int cgen_avgpool_bwd(int32_t dtype, int32_t n, int32_t ho, int32_t wo, int32_t d, cgen_view gout, cgen_view gin,
                     int32_t accumulate, cgen_stream_t);
/* aten::adaptive_avg_pool2d fwd/bwd for a Block with a float down-rate (vae.py:79-81): output cell o averages
 * [floor(o*in/out), ceil((o+1)*in/out)) along each axis; the backward pass gathers gout / window area. */
int cgen_adaptive_avgpool_fwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view in, cgen_view out,
                              cgen_stream_t);
int cgen_adaptive_avgpool_bwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view gout, cgen_view gin,
                              int32_t accumulate, cgen_stream_t);
/* out[n,y,x,:] = in[n, floor(y*hi/ho), floor(x*wi/wo), :] + (bias ? bias[y,x,:] : 0); bias is f32 [ho][wo][C] */
int cgen_upsample_fwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view in,
                      const float* bias, cgen_view out, cgen_stream_t);
int cgen_upsample_bwd(int32_t dtype, int32_t n, int32_t hi, int32_t wi, int32_t ho, int32_t wo, cgen_view gout,
                      cgen_view gin, int32_t accumulate, cgen_stream_t);
/* out[y,x,c] (+)= unscale * sum_n in[n,y,x,c]  (gradient of a batch-broadcast parameter); out f32 contiguous [h][w][C];
 * unscale = 1 / loss scale of the engine whose 16-bit gradients `in` holds (1 for f32) */
int cgen_batch_reduce(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view in, float* out, int32_t accumulate,
                      float unscale, cgen_stream_t);
/* out[n,y,x,c] = src[y,x,c] (batch broadcast of an f32 [h][w][C] parameter) */
int cgen_batch_broadcast(int32_t dtype, int32_t n, int32_t h, int32_t w, const float* src, cgen_view out, cgen_stream_t);
/* out = alpha*in (+ out if accumulate); channels >= c_from additionally scaled by beta.  in may be absent (fill alpha). */
int cgen_axpby(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view in, cgen_view out, float alpha, float beta,
               int32_t c_from, int32_t accumulate, cgen_stream_t);
/* NCHW (f32 or u8, contiguous) -> NHWC view in `dtype`: out = (in - sub) * mul */
int cgen_nchw_to_nhwc(int32_t src_is_u8, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, const void* src,
                      cgen_view out, float sub, float mul, cgen_stream_t);
/* Direct 7x7 stem conv, forward (vae.py:104-110,126: Encoder.stem, Cin 1..4, Cout 16 / 32 / 64): halo tile and the whole
 * weight matrix in LDS, one output pixel x all Cout per thread; weight_oihw / bias are the f32 parameters themselves (the 16-bit
 * engine rounds the weights to the 16-bit storage format on load).  No patch tensor on the forward path; cgen_im2col + the 1x1 weight-gradient
 * kernels remain the backward route. */
int cgen_stem_conv_supported(int32_t dtype, int32_t cin, int32_t ks, int32_t co);
int cgen_stem_conv_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t ks, int32_t co, cgen_view in,
                       const float* weight_oihw, const float* bias, cgen_view out, cgen_stream_t);
/* im2col for the thin-K stem conv (vae.py:104-110): out[n,y,x, c*ks*ks + tap] = in[n, y+dy, x+dx, c], zeros outside the
 * image and in out's padding channels (out.c = in.c*ks*ks, out.cpad = ceil8(out.c)).  The 7x7 stem then runs as a 1x1
 * conv over 49*Ci channels on the MFMA kernels, with the OIHW weight used as the [Co][49*Ci] matrix unchanged. */
int cgen_im2col(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t ks, cgen_view in, cgen_view out, cgen_stream_t);
/* General im2col / col2im (config 1, simple_vae.py:38-49: the 5x5/s2/p1 and 3x3/s2/p1 convolutions run as im2col + a
 * 1x1 conv with the OIHW weight used as the [Co][Ci*ks*ks] matrix).  out[n,oy,ox, c*ks*ks + tap] =
 * in[n, oy*stride - pad + dy, ox*stride - pad + dx, c] (zeros outside; channels [out.c, out.cpad) zeroed);
 * col2im is its transpose: gin (+)= scatter of gcol. */
int cgen_im2col_strided(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t ks, int32_t stride, int32_t pad, int32_t ho,
                        int32_t wo, cgen_view in, cgen_view out, cgen_stream_t);
int cgen_col2im_strided(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t ks, int32_t stride, int32_t pad, int32_t ho,
                        int32_t wo, cgen_view gcol, cgen_view gin, int32_t accumulate, cgen_stream_t);
/* Stand-alone unary ops where an activation cannot be fused into a consumer (simple_vae.py:13 LeakyReLU applied to one
 * segment of a concatenation only; .clamp(min=EPS) on log-scales, :68,97): op = CGEN_ACT_* or CGEN_UNARY_*;
 * bwd: gin (+)= gout * f'(x) */
#define CGEN_UNARY_LEAKY_RELU 3 /* param = negative slope */
#define CGEN_UNARY_CLAMP_MIN 4  /* param = minimum; gradient passes where x >= min */
#define CGEN_UNARY_ADD 5        /* x + param (temperature: logscale + log t) */
int cgen_unary_fwd(int32_t dtype, int32_t op, float param, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view in, cgen_view out,
                   cgen_stream_t);
int cgen_unary_bwd(int32_t dtype, int32_t op, float param, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view x, cgen_view gout,
                   cgen_view gin, int32_t accumulate, cgen_stream_t);
/* NHWC view -> contiguous NCHW f32 */
int cgen_nhwc_to_nchw(int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, cgen_view in, float* dst, cgen_stream_t);

/* ------------------------------------------------------------------ latent layer (K10, K16)
 * Fused sample_gaussian + gaussian_kl (vae.py:14-30, 268-269):
 *   z = q_loc + exp(q_ls + logt) * eps ; kl = -.5 + p - q + .5 (e^{2q} + (q_loc-p_loc)^2) e^{-2p}  (p = p_ls+logt, q = q_ls+logt)
 * eps: explicit tensor (parity runs) or NULL => Philox4x32-10 + Box-Muller keyed by rng[0]=seed, rng[1]=offset
 * (device memory, so a captured graph replays fresh noise) and `stream_id`; eps_out (optional) receives it.
 * kl_part[b*kl_stride + chunk] = per-sample partial sums (deterministic two-stage reduction), chunk <
 * cgen_reparam_kl_chunks(h,w,c); the caller lays all layers' chunks of one sample side by side (kl_stride = total). */
int cgen_reparam_kl_chunks(int32_t h, int32_t w, int32_t c);
int cgen_reparam_kl_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                        cgen_view p_loc, cgen_view p_ls, cgen_view eps_in, const uint64_t* rng, uint32_t stream_id,
                        float logt, cgen_view z, cgen_view eps_out, float* kl_part, int32_t kl_stride, cgen_stream_t);
/* Backward: gz = d/dz (view, may be absent), kl_coef_dev[b*coef_stride] = d elbo / d kl_sum[b] (device, f32;
 * coef_stride 0 broadcasts one value).  dz/dq_ls = e^{q_ls} eps is taken as (z - q_loc), so eps is not needed.
 * Writes (or accumulates into) the four gradients. */
int cgen_reparam_kl_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                        cgen_view p_loc, cgen_view p_ls, cgen_view z, float logt, cgen_view gz,
                        const float* kl_coef_dev, int32_t coef_stride, const float* kl_chan_scale, cgen_view g_q_loc,
                        cgen_view g_q_ls, cgen_view g_p_loc, cgen_view g_p_ls, int32_t acc_q, int32_t acc_p, cgen_stream_t);
/* The same launch carrying a RIDER: extra workgroups copy (ride_acc = 0) or accumulate (1) the [n,h,w,ride_src.c] view
 * ride_src into ride_dst.  In the decoder block  h = z_proj(z, pa) + h + p_feat  (vae.py:279-287) the gradient of the
 * residual p_feat -- a channel slice of the prior Block's output -- is the gradient of h verbatim; it has to land in the
 * prior output's gradient buffer next to the g_p_loc / g_p_ls this kernel writes, and used to be a launch of its own per
 * decoder block.  CGEN_F16 (16-bit storage), every view 16-byte aligned with channel counts in multiples of 8 (CGEN_EINVAL otherwise). */
int cgen_reparam_kl_bwd_rider(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                              cgen_view p_loc, cgen_view p_ls, cgen_view z, float logt, cgen_view gz,
                              const float* kl_coef_dev, int32_t coef_stride, const float* kl_chan_scale, cgen_view g_q_loc,
                              cgen_view g_q_ls, cgen_view g_p_loc, cgen_view g_p_ls, int32_t acc_q, int32_t acc_p,
                              cgen_view ride_src, cgen_view ride_dst, int32_t ride_acc, cgen_stream_t);
/* kl_chan_scale (optional, [c]): per-channel multiplier of the KL gradient -- the free-bits mask of this layer.
 *
 * Free bits (kl_free_bits > 0, vae.py:443-449).  S[b*out_stride + ch] = sum_{h,w} KL(q||p)[b,h,w,ch] of one layer
 * (c must divide 256; one deterministic workgroup per sample).  The caller lays the layers' channels side by side. */
int cgen_kl_channel_sums(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view q_loc, cgen_view q_ls,
                         cgen_view p_loc, cgen_view p_ls, float logt, float* out, int32_t out_stride, cgen_stream_t);
/* element-wise KL(q || p) map on flat contiguous f32 arrays, no reduction: the module-level gaussian_kl of vae.py:14-25
 * (-0.5 + p_ls - q_ls + 0.5 * (exp(q_ls)^2 + (q_loc - p_loc)^2) / exp(p_ls)^2; no clamps) */
int cgen_gaussian_kl_map(int64_t count, const float* q_loc, const float* q_ls, const float* p_loc, const float* p_ls,
                         float* out, cgen_stream_t);
/* z = loc + exp(ls + logt) * eps (prior sampling, vae.py:283-286); eps as above */
int cgen_sample_gaussian(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view loc, cgen_view ls,
                         cgen_view eps_in, const uint64_t* rng, uint32_t stream_id, float logt, cgen_view z,
                         cgen_stream_t);
/* Mediator latent z* (vae.py:485-513), q = q_ls+logt, p = p_ls+logt: u=(z-q_loc)/e^{q}; r_loc=a q_loc+(1-a)p_loc;
 * r_var=a^2 e^{2q}+(1-a)^2 e^{2p}; z* = r_loc + sqrt(r_var)*t*u   (t<=0 => no temperature factor).
 * linear_var != 0: r_var = a e^{2q} + (1-a) e^{2p}, the config-1 model's variant (simple_vae.py:383-386) */
int cgen_mediator_mix(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view z, cgen_view q_loc,
                      cgen_view q_ls, cgen_view p_loc, cgen_view p_ls, float alpha, float t, float logt, int32_t linear_var, cgen_view out,
                      cgen_stream_t);

/* ------------------------------------------------------------------ likelihoods (K11-K14) and ELBO
 * Discretised Gaussian (vae.py:352-411).  params view holds [loc(C) | logscale(C) | coeff(3, only C==3)] per pixel
 * (the raw 1x1-conv outputs); x is NHWC.  For C == 3 a params view of >= 9 channels selects vae.py's autoregressive
 * RGB form; a 6-channel view [loc(3) | logscale(3)] selects independent channels (simple_vae.py:103-171, no coeffs).  nll_part[b*nchunk+chunk] = partial sums of -log p over (C,H,W). */
int cgen_like_chunks(int32_t h, int32_t w);
int cgen_dgauss_nll_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x,
                        float* nll_part, cgen_stream_t);
/* g_params = coef_dev[b*coef_stride] * d(sum -log p)/d params */
int cgen_dgauss_nll_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x,
                        const float* coef_dev, int32_t coef_stride, cgen_view g_params, cgen_stream_t);
/* DGaussNet.sample (vae.py:413-422): loc (RGB autoregressive on the clamped predicted channels) and exp(logscale + log t),
 * NCHW f32 out.  rng == NULL: return_loc=True (x = clamp(loc)); rng = device (seed, offset): return_loc=False,
 * x = clamp(loc + scale * N(0,1)) with Philox noise of stream `stream_id` */
int cgen_dgauss_sample(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, float logt,
                       const uint64_t* rng, uint32_t stream_id, float* x_nchw, float* scale_nchw, cgen_stream_t);
/* DGaussNet.forward (vae.py:352-386): (loc, logscale) from the heads' outputs `params` = [loc(c) | logscale(c) | coeffs(3 if c == 3)]:
 * logscale = max(ls, -9) + logt; RGB: autoregressive means with tanh coefficients -- on the TRUE pixels `x` (NHWC view, c channels) when
 * given (vae.py:370-377), on the clamped predicted channels when x.p == NULL (vae.py:360-369).  Outputs NCHW f32. */
int cgen_dgauss_params(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x, float logt,
                       float* loc_nchw, float* logscale_nchw, cgen_stream_t stream);
/* Logit-space Gaussian of the config-1 model (simple_vae.py:173-248, GaussNet; x_like = *_gauss).  params = [loc(C) |
 * logscale(C)].  nll: x in [-1,1] -> (x+1)*127.5 + u, u ~ U[0,1) -> logit(./256) -> -log N(.; loc, exp(max(logscale,-9)))
 * (simple_vae.py:215-229; no log-determinant, as the reference).  u: NHWC view of injected uniforms (u.p != NULL), else
 * Philox uniforms from the device (seed, offset) pair `rng`, stream `stream_id`; bwd must be given the same u / rng state.
 * sample (simple_vae.py:231-238): scale = exp(logscale + logt) in both modes, x = loc (+ scale*N(0,1) when rng != NULL)
 * -> sigmoid*256 -> clamp((.-128)/128, -1, 1); NCHW f32 out. */
int cgen_gauss_nll_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x, cgen_view u,
                       const uint64_t* rng, uint32_t stream_id, float* nll_part, cgen_stream_t);
int cgen_gauss_nll_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, cgen_view x, cgen_view u,
                       const uint64_t* rng, uint32_t stream_id, const float* coef_dev, int32_t coef_stride,
                       cgen_view g_params, cgen_stream_t);
int cgen_gauss_sample(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view params, float logt,
                      const uint64_t* rng, uint32_t stream_id, float* x_nchw, float* scale_nchw, cgen_stream_t);
/* Discretised mixture of logistics, 10 mixtures, 3 channels (dmol.py:24-118, 164-215, 121-161). logits: [.,100].
 * nll_fwd / nll_bwd: dtype may be OR-ed with CGEN_DMOL_LOW_BIT for the reference's low_bit = True branch (5-bit pixels: half-bin
 * 1/31 and the mid-bin fallback's log 15.5 instead of 1/255 and log 127.5; dmol.py:52-60, 88-102). */
#define CGEN_DMOL_LOW_BIT 0x100
int cgen_dmol_nll_fwd(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view logits, cgen_view x, float* nll_part,
                      cgen_stream_t);
int cgen_dmol_nll_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view logits, cgen_view x,
                      const float* coef_dev, int32_t coef_stride, cgen_view g_logits, cgen_stream_t);
/* mode 0 soft mean, 1 hard (argmax) mean, 2 sample (Gumbel argmax + logistic noise from rng, temperature logt),
 * 10 + k (k = 1..9): top-k mean -- mixtures below the k-th largest logit are switched off and the rest renormalised */
int cgen_dmol_decode(int32_t dtype, int32_t n, int32_t h, int32_t w, cgen_view logits, int32_t mode,
                     const uint64_t* rng, uint32_t stream_id, float logt, float* x_nchw, float* scale_nchw, cgen_stream_t);
/* elbo/nll/kl (vae.py:450-457): nll = mean_b( sum(nll_part[b]) / nll_div ), kl = mean_b( sum(kl_part[b]) / kl_div ),
 * out3 = {nll + beta*kl, nll, kl}.  kl_part: [nkl][B] per-sample sums already reduced per layer by the caller's
 * layout: kl_part[b*kl_stride + j], j < kl_count.  beta_dev (optional, device memory) overrides beta, so a captured
 * launch follows trainer.py's beta warm-up without re-capture. */
int cgen_elbo_finalize(int32_t n, const float* nll_part, int32_t nll_count, float nll_div, const float* kl_part,
                       int32_t kl_count, float kl_div, float beta, const float* beta_dev, float* out3, cgen_stream_t);
/* free-bits variant: kl = sum_j max(free_bits, mean_b kl_bc[b*ncol + j]) / kl_div; out3 as above;
 * chan_mask[j] = d max/d mean (1 / 0 / 0.5 on a tie) for cgen_reparam_kl_bwd's kl_chan_scale */
int cgen_elbo_finalize_fb(int32_t n, const float* nll_part, int32_t nll_count, float nll_div, const float* kl_bc, int32_t ncol,
                          float kl_div, float free_bits, float beta, const float* beta_dev, float* out3, float* chan_mask,
                          cgen_stream_t);
/* Backward of the counterfactual pixel step composed with both heads' DGaussNet.sample(h) decodes (dscm.py:52-56 over
 * vae.py:352-385,413-422), for the differentiable counterfactual branch of DSCM.forward (train_cf.py:159-183 fine-tunes the
 * HVAE through it): g_cfx_nchw = d loss / d cf_x (f32 NCHW), gscale = 1 / cf_particles; writes d/d params of the
 * reconstruction head and of the counterfactual head ([loc | logscale | coeffs] NHWC, same layout as the inputs). */
int cgen_cf_dgauss_bwd(int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, cgen_view rec_params, cgen_view cf_params,
                       cgen_view x, const float* g_cfx_nchw, float gscale, cgen_view g_rec_params, cgen_view g_cf_params,
                       cgen_stream_t);
/* Counterfactual pixel step (dscm.py:55-63): u=(x-rec_loc)/max(rec_scale,1e-12); cf=clamp(cf_loc+cf_scale*u,-1,1);
 * optional running sums sum_x += cf, sum_x2 += cf^2.  All NCHW f32 contiguous, `count` elements. */
int cgen_cf_pixels(int64_t count, const float* x, const float* rec_loc, const float* rec_scale, const float* cf_loc,
                   const float* cf_scale, float* cf_x, float* sum_x, float* sum_x2, cgen_stream_t);

/* ------------------------------------------------------------------ step tail (K17; trainer.py:67-87, utils.py:178-225)
 * Flat-buffer fused global-norm -> clip -> skip predicate -> AdamW -> EMA.
 * state_dev: f32[8] = {sum_sq, grad_norm, clip_coef, skip_flag, n_skipped, opt_steps, n_skipped_nonfinite, -}.  LambdaLR warm-up, Adam
 * bias correction and the EMA warm-up decay are all derived ON DEVICE from opt_steps (successful steps so far), so
 * the host never reads the skip decision (the reference syncs three times per step).
 * out3 (optional) = {elbo, nll, kl}: NaN nll/kl forces a skip as trainer.py:71-74 does. */
int cgen_sumsq_partial(const float* g, int64_t count, float* partial /*[nblk]*/, int32_t nblk, cgen_stream_t);
int cgen_clip_decide(const float* partial, int32_t nblk, const float* out3, float max_norm, float skip_norm,
                     float* state_dev, cgen_stream_t);
typedef struct cgen_adamw_args {
  float* p; const float* g; float* m; float* v; float* ema; /* ema may be NULL */
  int64_t count;
  float lr, beta1, beta2, eps, wd, ema_beta;
  int32_t warmup_steps, ema_update_after;
  const float* state_dev; /* reads clip_coef, skip_flag, opt_steps */
} cgen_adamw_args;
int cgen_adamw_ema(const cgen_adamw_args* a, cgen_stream_t);
/* opt_steps += 1 unless the step was skipped (run once after all cgen_adamw_ema launches of a step) */
int cgen_step_commit(float* state_dev, cgen_stream_t);

/* Philox normal fill (f32 contiguous) -- exposed for tests of the in-kernel generator */
int cgen_philox_normal(float* out, int64_t count, const uint64_t* rng, uint32_t stream_id, cgen_stream_t);
/* rng[1] += inc (device-side counter bump so graph replays draw fresh noise) */
int cgen_rng_advance(uint64_t* rng, uint64_t inc, cgen_stream_t);

#ifdef __cplusplus
}
#endif
#endif /* CGEN_HIP_H */
