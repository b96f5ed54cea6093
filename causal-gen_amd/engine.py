"""Execution engine: NHWC views over a bump arena, one Python call per fused HIP launch, and a reverse tape.

The reference leans on torch.autograd over ~10 tiny ATen ops per conv.  Here the forward is a short
list of fused launches (conv with virtual concat + activation prologue + residual epilogue, pool, upsample+bias,
reparam+KL, NLL) and the backward replays the tape with hand-written gradient kernels: dgrad is the same conv kernel
on a flipped weight image with the activation derivative in the epilogue, wgrad is split-K partials reduced by one
multi-tensor launch.  Every buffer comes from an arena that is reset per step, so a step has a fixed launch
sequence with fixed addresses and can be captured in a hipGraph.

PyTorch supplies device memory, streams and the nn.Module parameter containers only.
"""
import ctypes as C
import os
import math

import torch

from . import _lib
from .wgrad_sched import WgradMixin
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, F16, F32, F32S, NULL_VIEW, UNARY_CLAMP_MIN, UNARY_LEAKY_RELU, View

_ALIGN = 256


def _ceil(a, m):
    return (a + m - 1) // m * m


def compact_parents(t):
    """[B,ctx] view of a spatially constant parents tensor given as [B,ctx], [B,ctx,1,1] or an expand()-ed [B,ctx,R,R]
    (strides 0 over H and W); None for a materialised [B,ctx,R,R] (which may vary over space)."""
    if t.dim() == 2:
        return t
    if t.dim() == 4 and ((t.shape[2] == 1 and t.shape[3] == 1) or (t.stride(2) == 0 and t.stride(3) == 0)):
        return t[:, :, 0, 0]
    return None


def expand_parents(t, h, w=None):
    """[B,ctx] -> the reference's [B,ctx,H,W] shape as a stride-0 view (no copy)."""
    return t[:, :, None, None].expand(-1, -1, h, h if w is None else w)


class StaticParents:
    """Stable-address copy of a parents tensor for a captured graph, keeping the stride-0 (virtual) form when it has one."""

    def __init__(self, pa):
        c = compact_parents(pa)
        self.virtual = c is not None
        if self.virtual:
            self.buf = c.clone()
            self.t = self.buf if pa.dim() == 2 else self.buf[:, :, None, None].expand(*pa.shape)
        else:
            self.buf = self.t = pa.clone()

    def load(self, pa):
        src = compact_parents(pa) if self.virtual else pa
        assert src is not None and src.shape == self.buf.shape, "parents changed form (materialised <-> broadcast) between calls"
        self.buf.copy_(src, non_blocking=True)


class _Tape(list):
    """The reverse tape: (backward fn, args, recorded on the side stream?).  The tag lets backward() run the z strand of the
    decoder (prior Blocks, z_feat_proj, the upsampling of z) on the side stream again, next to the h strand."""
    __slots__ = ("eng",)

    def append(self, item):
        if len(item) == 2:
            item = (item[0], item[1], self.eng._bw_tag)
        list.append(self, item)


class NT:
    """NHWC strided view (channel stride 1) -- the Python twin of cgen_view."""
    __slots__ = ("ptr", "n", "h", "w", "c", "sn", "sh", "sw", "base", "coff", "rg", "keep", "_cv", "es", "cpad", "bsrc", "rem")

    def __init__(self, ptr, n, h, w, c, sn, sh, sw, es, base=None, coff=0, rg=True, keep=None):
        self.ptr, self.n, self.h, self.w, self.c = ptr, n, h, w, c
        self.sn, self.sh, self.sw, self.es = sn, sh, sw, es
        self.base = base if base is not None else self
        self.coff, self.rg, self.keep, self._cv = coff, rg, keep, None
        self.cpad = 0  # channels [c, cpad) are guaranteed zero (see cgen_view.cpad)
        self.bsrc = None  # the [n,1,1,c] tensor this one broadcasts over H x W (row / pixel strides 0), else None
        # f16 residual trunk: byte offset of the REMAINDER plane (same layout; value = this + remainder, include/cgen_hip.h
        # cgen_conv_args.out_rem), 0 = none.  Only residual epilogues read it; a conv input is the 16-bit tensor alone.
        self.rem = 0

    def broadcast(self, h, w):
        """A [n,1,1,c] tensor seen as [n,h,w,c] with row and pixel stride 0: spatially constant parents without the
        [B,ctx,R,R] buffer (SURVEY 8f row 2 -- every conv kernel takes arbitrary strides, and a tile's DMA of such a segment
        re-reads one 16-byte group per sample from cache instead of streaming H*W copies of it from HBM)."""
        assert self.h == 1 and self.w == 1 and not self.rg
        if h == 1 and w == 1:
            return self
        v = NT(self.ptr, self.n, h, w, self.c, self.sn, 0, 0, self.es, base=None, coff=0, rg=False, keep=self.keep)
        v.cpad, v.bsrc = self.cpad, self
        return v

    def cv(self):
        if self._cv is None:
            self._cv = View(self.ptr, self.sn, self.sh, self.sw, self.c, self.cpad)
        return self._cv

    def chan(self, a, b):
        assert 0 <= a < b <= self.c
        v = NT(self.ptr + a * self.es, self.n, self.h, self.w, b - a, self.sn, self.sh, self.sw, self.es,
               base=self.base, coff=self.coff + a, rg=self.rg, keep=self.keep)
        if b == self.c and self.cpad:  # a view that ends where the tensor ends keeps its zero padding
            v.cpad = self.cpad - a
        v.rem = self.rem
        return v

    def crop(self, r):
        """[:, :r, :r, :] -- only used on tensors that need no gradient (parents)."""
        if r == self.h and r == self.w:
            return self
        assert not self.rg
        if self.bsrc is not None:
            return self.bsrc.broadcast(r, r)
        v = NT(self.ptr, self.n, r, r, self.c, self.sn, self.sh, self.sw, self.es, base=None, coff=0, rg=False, keep=self.keep)
        v.cpad = self.cpad
        return v

    @property
    def shape(self):
        return (self.n, self.h, self.w, self.c)


class Arena:
    """Bump allocator over a few large device buffers; reset() per step keeps addresses stable across steps."""

    def __init__(self, device, chunk_bytes=1 << 30):
        self.device, self.chunk_bytes = device, chunk_bytes
        self.chunks, self.ci, self.off = [], 0, 0
        self.high_water = 0

    def reset(self):
        self.ci, self.off = 0, 0

    def alloc(self, nbytes):
        nbytes = _ceil(max(nbytes, 1), _ALIGN)
        while True:
            if self.ci >= len(self.chunks):
                size = max(self.chunk_bytes, nbytes)
                self.chunks.append(torch.empty(size, dtype=torch.uint8, device=self.device))
                self.off = 0
            buf = self.chunks[self.ci]
            base = buf.data_ptr()
            start = _ceil(base + self.off, _ALIGN) - base
            if start + nbytes <= buf.numel():
                self.off = start + nbytes
                self.high_water = max(self.high_water, sum(c.numel() for c in self.chunks[:self.ci]) + self.off)
                return base + start
            self.ci += 1
            self.off = 0


class ConvSite:
    """One nn.Conv2d of the model together with how its input channels are split into segments."""

    def __init__(self, name, conv, seg_c, seg_rg, index, as_1x1=False):
        self.name, self.conv, self.index = name, conv, index
        self.ks = conv.kernel_size[0]
        self.co, self.ci = conv.out_channels, conv.in_channels
        self.im2col = 0
        if as_1x1:  # thin-K stem: run on an im2col'd input as a 1x1 conv over ks*ks*Ci channels (same OIHW memory)
            self.im2col = self.ks
            self.ci *= self.ks * self.ks
            self.ks = 1
        assert sum(seg_c) == self.ci, (name, seg_c, self.ci)
        self.seg_c, self.seg_rg = tuple(seg_c), tuple(seg_rg)
        self.seg_off = [sum(seg_c[:i]) for i in range(len(seg_c))]
        self.taps = self.ks * self.ks
        # weight images (include/cgen_hip.h): rows x krow, column = tap * C8 + channel, C8 = sum ceil8(C_s)
        self.krow = _ceil(self.taps * sum(_ceil(c, 8) for c in seg_c), 32) + 32
        self.krow_dg = _ceil(self.taps * _ceil(self.co, 8), 32) + 32
        self.fwd_numel = _ceil(self.co, 16) * self.krow
        self.dg_numel = [(_ceil(c, 16) * self.krow_dg) if rg else 0 for c, rg in zip(seg_c, seg_rg)]
        self.img_fwd = None   # device address
        self.img_dg = [None] * len(seg_c)
        # fused light Block (csrc/block.hip): `blk3` = ("a", partner) for the Block's first conv, ("b", partner) for its second;
        # fragment-ordered weight images (include/cgen_hip.h, cgen_block3_args): device addresses, set by Engine.bind
        self.blk3 = None
        # fused default Block (csrc/block4.hip): `blk4` = (role 0..3, [the Block's four sites])
        self.blk4 = None
        self.frag = {}        # "a_fwd" / "b_fwd" / "a_dg" / ("b_dg", k); blk4: "p0_fwd" ... (plan_frag_images4)
        self.frag_numel = {}

    def plan_frag_images4(self):
        """Fragment images of the fused DEFAULT Block (cgen_block4, include/cgen_hip.h): phase p of the forward pass reads
        "p<p>_fwd" of conv p; phase p of the data gradient reads the transposed image of conv 3 - p ("p<p>_dg" lives at that conv)."""
        if self.blk4 is None:
            return
        role, sites = self.blk4
        b = sites[0].co
        if b > 64 or sites[1].ks != 3 or sites[2].ks != 3:
            return
        g, nmb = _ceil(b, 16) // 16, _ceil(b, 32) // 32
        if role == 0:
            self.frag_numel["p0_fwd"] = nmb * sum(_ceil(c, 32) // 32 for c in self.seg_c) * 2 * 512
            for k, (c, rg) in enumerate(zip(self.seg_c, self.seg_rg)):
                if rg:
                    self.frag_numel[("p3_dg", k)] = _ceil(c, 32) // 32 * g * 512
        elif role == 1:
            self.frag_numel["p1_fwd"] = self.frag_numel["p2_dg"] = nmb * 9 * g * 512
        elif role == 2:
            self.frag_numel["p2_fwd"] = self.frag_numel["p1_dg"] = nmb * 9 * g * 512
        else:
            self.frag_numel["p3_fwd"] = _ceil(self.co, 32) // 32 * g * 512
            self.frag_numel["p0_dg"] = nmb * (_ceil(self.co, 32) // 32) * 2 * 512

    def plan_frag_images(self):
        """Element counts of the fragment-ordered images this site needs as part of a fused Block (512 elements = one KiB
        fragment: 64 lanes x 8)."""
        if self.blk3 is None or self.ks != 3:
            return
        role, other = self.blk3
        if role == "a":   # conv1: [b][Ci][3][3]
            b = self.co
            nks = (9 * b + 15) // 16
            # (segments in whole 32-channel chunks; one image per 32-row block of the bottleneck: widths above 32 are the small-image instance's)
            self.frag_numel["a_fwd"] = _ceil(b, 32) // 32 * sum(_ceil(c, 32) // 32 for c in self.seg_c) * 18 * 512
            if len(self.seg_c) == 1 and b <= 16 and self.seg_c[0] % 32 == 0 and self.seg_c[0] <= 64:
                self.frag_numel["a16_fwd"] = 9 * (self.seg_c[0] // 32) * 512  # (the row-streaming instance: 16-row image, modes 6 / 7)
            for k, (c, rg) in enumerate(zip(self.seg_c, self.seg_rg)):
                if rg:
                    self.frag_numel[("b_dg", k)] = _ceil(c, 32) // 32 * nks * 512
        else:             # conv2: [Co][b][3][3]
            b = self.ci
            nks = (9 * b + 15) // 16
            self.frag_numel["b_fwd"] = _ceil(self.co, 32) // 32 * nks * 512
            self.frag_numel["a_dg"] = _ceil(b, 32) // 32 * (_ceil(_ceil(self.co, 8), 32) // 32) * 18 * 512
            if b <= 16 and self.co % 32 == 0 and self.co <= 64:
                self.frag_numel["a16_dg"] = 9 * (self.co // 32) * 512


class Engine(WgradMixin):
    CHUNK = 1024  # elements per block of the multi-tensor kernels (MT_CHUNK in conv.hip)

    def __init__(self, device, dtype="f32"):
        rawlib = _lib.require_gpu()
        self.lib = rawlib
        self.device = torch.device(device)
        assert self.device.type == "cuda"
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if dtype not in ("f32", "f16"):
            raise ValueError("compute dtype %r: the engine computes in 'f32' (exact f32 MFMA) or 'f16' (binary16 storage, f32 accumulate; "
                             "the bfloat16 storage of rounds 1-2 is gone -- use 'f16')" % (dtype,))
        self.dt = {"f32": F32, "f16": F16}[dtype]
        self.dtype_name = dtype
        self.es = 4 if self.dt == F32 else 2
        self.tdtype = torch.float32 if self.dt == F32 else (torch.bfloat16 if rawlib.h16_is_bf16 else torch.float16)  # (what the library was built for)
        # f16 engine: activation GRADIENTS are carried times a power-of-two loss scale so that they sit in binary16's normal
        # range (the seeds are ~1 / (B * dims) ~ 1e-6); the kernels that turn them into f32 parameter gradients (the split-K
        # reduce, the batch reduce of the decoder biases) multiply by 1 / scale.  Exact: scaling by 2^k commutes with rounding.
        # Set per backward pass by set_loss_scale(B * dims); 1 for the f32 engine.
        self.loss_scale = 1.0
        # back-off of that rule in powers of two (<= 0): lowered by TrainStep when a step was dropped for a non-finite gradient
        # (an activation gradient past binary16's 65504), raised again after a run of clean steps -- the reduce tables are keyed by
        # the scale, captured graphs are rebuilt by whoever changes it
        self.loss_scale_shift = 0
        # f16 engine: the residual trunk as (value, remainder) pairs -- see conv(trunk=True).  CGEN_TRUNK_REM: 0 never, 1 (default)
        # in inference passes (abduct / forward_latents / sample / a no-grad forward: the counterfactual loop, whose pixels gain
        # 2x accuracy from it), 2 also in recorded training passes.  Measured on ukbb192 B = 32: the planes double the epilogue
        # traffic of the trunk convs (HBM-bound at >= 96x96): counterfactuals/s -12.6 %, a training step -8.5 % -- for an ELBO
        # that is within 1.2e-5 of the reference either way, which is why training keeps the plain trunk.
        self.trunk_mode = int(os.environ.get("CGEN_TRUNK_REM", "1"))
        # f32 engine: convolutions of NON-recording passes (abduct / forward_latents / sample / a no-grad forward: the counterfactual
        # loop) hand cgen_conv2d CGEN_F32S -- f32 tensors, every product from split binary16 operands (three f16 MFMAs per K-step,
        # ~2^-21 relative): the path that meets north_star's 1e-3 on counterfactual pixels at a fifth of the f32 MFMA's cycles.
        # Recorded (training) passes stay on exact f32 MFMA chains (gradients leave binary16's range).  CGEN_F32_SPLIT=0: off.
        self.f32_split = int(os.environ.get("CGEN_F32_SPLIT", "1")) if self.dt == F32 else 0
        # CGEN_ABLATE="f1,d3" (planning tool, results are WRONG): what would the step cost if a light Block were one launch as
        # long as its HBM-bound half -- the upper bound of Block fusion on the critical path
        self._ablate = frozenset(t for t in os.environ.get("CGEN_ABLATE", "").split(",") if t)  # exact tokens
        if self._ablate:
            import sys
            print("causal-gen_amd: CGEN_ABLATE=%s -- TIMING-ONLY ablation, launches are skipped and every result is WRONG" % ",".join(sorted(self._ablate)),
                  file=sys.stderr, flush=True)
        self.trunk_maxres = int(os.environ.get("CGEN_TRUNK_REM_MAXRES", "100000"))  # planes only on images up to this side
        self.arena = Arena(self.device)
        self.tape, self.recording = _Tape(), False
        self.tape.eng = self
        self.grads = {}
        self.sites, self.site_by_id = [], {}
        self.stream = 0
        self._img_buf = None
        self._prep_tab = None
        self._wg_events = []
        self._wg_reduced, self._wg_seen = 0, set()
        self._in_side = False
        # (tape entries carry a strand tag: 1 = recorded inside `on_side(fn, bw=1)`.  The two-strand backward that used it -- rounds 2, 4
        #  and 5, bit-identical and slower under hipGraph replay each time, LABNOTES 9.9 / 10.4 -- was removed in round 6; the tag stays in
        #  the tape's tuples)
        self._bw_tag = 0
        self._rng_override = None
        self._dbg_names = {} if os.environ.get("CGEN_DEBUG_NAMES") else None  # id(tensor) -> producing conv (tools/ab_grads.py)
        self._riders = {}
        self.ride = os.environ.get("CGEN_RIDER", "1") != "0"
        self._partials = {}
        self._red_tabs = {}
        self.params = None          # list of nn.Parameter in model.parameters() order
        self.flat_p = self.flat_g = None
        self.p_off = {}
        self.pgrad_init = set()
        self.rng = None
        self.launches = 0
        self.prof = None  # {"conv_fwd": [flops, [(ev0, ev1), ...]], ...} when profiling is on
        # weight-gradient kernels are leaves of the backward graph and independent of each other: they are DEFERRED to the
        # end of the backward pass and issued round-robin on a few HIP streams (one fork, one join before the split-K
        # reduction).  The low-resolution ones are latency chains on a fraction of the CUs; several at a time fill the chip.
        self.wgrad_streams = int(os.environ.get("CGEN_WGRAD_STREAMS", "2"))
        # ... or, better, packed: ONE launch per kernel variant runs the workgroups of all deferred problems (cgen_conv2d_wgrad_batch_*)
        self.wgrad_batch = os.environ.get("CGEN_WGRAD_BATCH", "1") != "0"
        # background flushes: once this many GFLOP of weight gradients are pending they are issued on a side stream with a
        # capped grid, so that they fill the CUs the latency-bound backward chain leaves idle (0: one batch at the end)
        self.wgrad_flush_frac = [float(v) for v in os.environ.get("CGEN_WGRAD_FLUSH_FRAC", "0.7").split(",") if v]
        # Data gradients of two ADJACENT fused Blocks of the tape (a decoder layer's posterior and prior Block) in one launch
        # (cgen_block3_pair): backward() arms the first one, whose launch is held until the second reaches its own launch point.
        self.blk3_pair = os.environ.get("CGEN_BLK3_PAIR", "1") != "0"
        self._blk3_arm = 0      # 1: the next fused data-gradient launch is to be held; 2: one is held, waiting for its partner
        self._blk3_hold = None  # (args, launch counter at hold time, tensors touched, the held Block's late bookkeeping)
        self.blk3_pairs = 0     # pair launches of the last backward pass
        # ... and the same for data-gradient convs on the small-image path (cgen_conv2d_pair): the two gradient outputs of one conv
        # (cat[z, p_feat] of z_feat_proj, cat[h, pa, acts] of an unfused posterior Block), and the second convs of an unfused
        # posterior / prior Block pair, which backward() brings next to each other
        self.conv_pair = os.environ.get("CGEN_CONV_PAIR", "1") != "0"
        self._conv_arm = 0
        self._conv_hold = None
        self.conv_pairs = 0
        self.wgrad_bg_wgs = int(os.environ.get("CGEN_WGRAD_BG_WGS", "304"))
        self.wgrad_bg_reduce = os.environ.get("CGEN_WGRAD_BG_REDUCE", "1") != "0"
        self._wg_cum, self._wg_total, self._wg_nflush = 0.0, 0.0, 0
        self._wg_batches = {}
        self._wg_forked = False
        self._wg_pool = []
        self._wg_deferred = []
        # independent sub-graphs of the forward pass (prior / posterior Block of a decoder layer) run on two streams
        self.fwd_branch = os.environ.get("CGEN_FWD_BRANCH", "1") != "0"
        # (read once: the decoder loop asks per layer and per call)  two-stream decoder sections in non-recording passes (abduct,
        # eval forward); main chain enqueued first behind a fork (fork_mark)
        self.infer_branch = os.environ.get("CGEN_INFER_BRANCH", "1") != "0"
        self.fwd_mainfirst = os.environ.get("CGEN_FWD_MAINFIRST", "1") != "0"
        self._fwd_side = None
        self._side_join_pending = False
        # fused light Block, round 4 (csrc/block.hip, cgen_block3): 0 off, 1 forward only, 2 forward + data gradient; images at
        # least CGEN_BLK3_MINRES wide
        self.blk3_on = int(os.environ.get("CGEN_BLK3", "2")) if self.dt == F16 else 0
        self.blk3_minres = int(os.environ.get("CGEN_BLK3_MINRES", "16"))
        # images up to 14 pixels wide (12x12, 6x6): the small-image instance of cgen_block3, one launch per Block there too
        self.blk3_small = int(os.environ.get("CGEN_BLK3S", "1"))
        # fused default Block (vae.py:57-71; csrc/block4.hip): the four convs of a non-light Block as one launch, forward and data gradient
        self.blk4_on = int(os.environ.get("CGEN_BLK4", "1")) if self.dt == F16 else 0
        self.blk4_launches = 0
        # ... and the data gradients of a decoder layer's posterior and prior Blocks in one launch (cgen_block4_pair), as _bw_block3 does
        self.blk4_pair = os.environ.get("CGEN_BLK4_PAIR", "1") != "0"
        self.blk4_pairs = 0
        self._blk4_arm, self._blk4_hold = 0, None
        # the row-streaming instance (96x96 / 192x192 Blocks, csrc/block.hip blk3r): CGEN_BLK3R=0 off; where it serves a shape the side
        # policy below (blk3_res) does not apply
        self.blk3_rows = int(os.environ.get("CGEN_BLK3R", "1"))
        # ... at which image sides: CGEN_BLK3_RES for Blocks with one or two input segments (trunk / prior / down Blocks),
        # CGEN_BLK3_RES3 for three-segment Blocks (the posterior: cat[h, pa, acts]); a comma list of sides and lo-hi ranges,
        # "0" = every side >= CGEN_BLK3_MINRES.  Defaults = where the fused launch beats the two it replaces INSIDE the step on MI355X
        # (A/B of bench.py in one session): the wide, few-pixel layers -- 24x24 / 48x48 of ukbb192, 28x28 / 56x56 of
        # mimic224 -- and the posterior Block one resolution further (96x96 / 112x112); above that the two launches are HBM-bound
        # and a tile visit costs the fused kernel more than it saves
        self.blk3_res = self._side_ranges(os.environ.get("CGEN_BLK3_RES", "20-64"))
        self.blk3_res3 = self._side_ranges(os.environ.get("CGEN_BLK3_RES3", "20-112"))
        # data parallelism: once this fraction of the pass's weight-gradient work has been issued (and the background flush is
        # out), `on_split` is called with the side stream joined -- the gradients of every conv reduced so far are FINAL
        # (`early_final`), so their all-reduce can travel under the rest of the backward pass (train.TrainStep)
        self.split_frac = float(os.environ.get("CGEN_DP_SPLIT_FRAC", "0.8"))
        self.on_split = None
        self.early_final = None

    @property
    def trunk_rem(self):
        return self.dt != F32 and (self.trunk_mode >= 2 or (self.trunk_mode == 1 and not self.recording))

    def set_loss_scale(self, n_terms):
        """Loss scale of the coming backward pass: the seeds are O(1 / n_terms) (n_terms = batch * dims * accumulation steps);
        2^k with k = round(log2 n_terms) - 1 puts them at ~1/2 per pixel.  Measured on MI355X (tools/f16_vs_f32.py): the largest
        activation gradient of a ukbb192 / morphomnist / mimic224 pass is then ~270 / 14 / 260 (overflow at 65504) and 0.2 / 8 / 10 %
        of the non-zero elements are subnormal; the parameter gradients' deviation from the f32 path (median 1e-3 relative L2) does
        not move between k - 3 and k + 3: it is the forward operand rounding, not gradient underflow.  A non-finite gradient (a
        spike 200x above that) fails the step's NaN / grad_skip predicate like any other (trainer.py:69-77).
        ``CGEN_LOSS_SCALE_LOG2`` overrides k."""
        self.loss_scale = self.loss_scale_rule(n_terms, self.dt == F32)
        if self.dt != F32 and self.loss_scale_shift:
            self.loss_scale = max(1.0, self.loss_scale * 2.0 ** self.loss_scale_shift)
        return self.loss_scale

    @staticmethod
    def loss_scale_rule(n_terms, is_f32=False):
        """The rule itself (host logic, no GPU): 1 for f32, else a power of two, 2^(round(log2 n_terms) - 1), >= 1."""
        if is_f32:
            return 1.0
        k = os.environ.get("CGEN_LOSS_SCALE_LOG2")
        k = int(k) if k is not None else max(0, int(round(math.log2(max(float(n_terms), 1.0)))) - 1)
        return float(2 ** k)

    # ------------------------------------------------------------------ memory
    def begin(self):
        """Start a new step/pass: recycle the arena, clear the tape and gradient bookkeeping."""
        self.generation = getattr(self, "generation", 0) + 1  # a recorded pass is only valid within its generation
        self.kl_coef_override = None
        self.arena.reset()
        self.tape.clear()
        self.grads.clear()
        self._wg_events = []
        self._wg_reduced, self._wg_seen = 0, set()
        self._riders = {}
        self._blk3_arm = self._conv_arm = self._blk4_arm = 0  # (a pass that ended in an exception may have left a launch held)
        self._blk3_hold = self._conv_hold = self._blk4_hold = None
        self.pgrad_init = set()
        self._pnhwc, self._pgrad_tmp = {}, {}
        self._adopted = set()
        self._wg_deferred = []
        self._wg_cum, self._wg_nflush = 0.0, 0
        self._wg_forked = False
        self._split_done = False
        self.passes = 0
        self.stream = torch.cuda.current_stream(self.device).cuda_stream

    def new(self, n, h, w, c, rg=True, es=None, rem=False):
        """Fresh NHWC tensor.  The pixel stride is rounded up to 8 channels so every pixel starts on a 16-byte
        boundary (16-byte vector / LDS-DMA access needs it); the padding channels are never read as data.
        `rem`: with a remainder plane behind it (f16 residual trunk)."""
        es = self.es if es is None else es
        cp = _ceil(c, 8)
        nbytes = _ceil(n * h * w * cp * es, _ALIGN)
        ptr = self.arena.alloc(nbytes * (2 if rem else 1))
        t = NT(ptr, n, h, w, c, h * w * cp, w * cp, cp, es, rg=rg)
        if rem:
            t.rem = nbytes
        return t

    def new_f32(self, count):
        return self.arena.alloc(count * 4)

    def wrap_nhwc(self, t, rg=False):
        """torch tensor [N,H,W,C] contiguous in the engine dtype -> NT (keeps a reference)."""
        assert t.is_contiguous() and t.dtype == self.tdtype and t.device == self.device
        n, h, w, c = t.shape
        return NT(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, self.es, rg=rg, keep=t)

    def from_nchw(self, t, rg=False, sub=0.0, mul=1.0):
        """torch NCHW (f32 or u8) -> engine NHWC tensor.  Zero-copy when C == 1, f32 and no affine."""
        assert t.device == self.device and t.dim() == 4
        n, c, h, w = t.shape
        if t.dtype == torch.float32 and self.dt == F32 and sub == 0.0 and mul == 1.0:
            if c == 1 and t.is_contiguous():
                return NT(t.data_ptr(), n, h, w, 1, h * w, w, 1, 4, rg=rg, keep=t)
            if t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
                return NT(t.data_ptr(), n, h, w, c, h * w * c, w * c, c, 4, rg=rg, keep=t)
        assert t.dtype in (torch.float32, torch.uint8), t.dtype
        t = t.contiguous()
        out = self.new(n, h, w, c, rg=rg)
        out.keep = t
        if c % 8:  # ragged width: zero the padding channels once so whole 16-byte groups can be fetched by DMA
            self.fill(self._padded(out), 0.0)
            out.cpad = _ceil(c, 8)
        self.lib.nchw_to_nhwc(1 if t.dtype == torch.uint8 else 0, self.dt, n, c, h, w, t.data_ptr(), out.cv(), sub, mul,
                              self.stream)
        self.launches += 1
        return out

    def from_parents(self, t, h=None, w=None):
        """Parents -> engine tensor.  The reference hands the HVAE ``pa[..., None, None].repeat(1, 1, R, R)``
        (trainer.py:16-21, dscm.py:125-131): spatially constant.  When the caller passes that tensor WITHOUT materialising
        it -- an ``expand``-ed view (strides 0 over H and W), or [B,ctx] / [B,ctx,1,1] with the target size -- only the
        [B,1,1,ctx] vector is laid out and every consumer sees a stride-0 broadcast of it.  Any other 4-D tensor takes the
        general path (arbitrary spatially varying parents stay supported)."""
        c = compact_parents(t)
        if c is None:
            return self.from_nchw(t.to(self.device, torch.float32))
        if t.dim() == 4 and t.shape[2] > 1:
            h, w = int(t.shape[2]), int(t.shape[3])
        assert h is not None and w is not None
        small = self.from_nchw(c.to(self.device, torch.float32).contiguous()[:, :, None, None])
        return small.broadcast(h, w)

    def to_nchw(self, x):
        """engine NHWC tensor -> fresh torch f32 NCHW tensor."""
        out = torch.empty((x.n, x.c, x.h, x.w), dtype=torch.float32, device=self.device)
        self.lib.nhwc_to_nchw(self.dt, x.n, x.c, x.h, x.w, x.cv(), out.data_ptr(), self.stream)
        self.launches += 1
        return out

    def to_torch_cl(self, x):
        """engine tensor -> torch tensor of NCHW *shape* in channels-last memory (zero-copy view of a clone)."""
        t = torch.empty((x.n, x.h, x.w, x.c), dtype=self.tdtype, device=self.device)
        dst = self.wrap_nhwc(t)
        self.lib.axpby(self.dt, x.n, x.h, x.w, x.cv(), dst.cv(), 1.0, 1.0, 1 << 30, 0, self.stream)
        self.launches += 1
        return t.permute(0, 3, 1, 2)

    def copy_in(self, x):
        """Arena copy of a tensor that lives in torch-allocated memory."""
        out = self.new(x.n, x.h, x.w, x.c, rg=False)
        self.lib.axpby(self.dt, x.n, x.h, x.w, x.cv(), out.cv(), 1.0, 1.0, 1 << 30, 0, self.stream)
        self.launches += 1
        return out

    # ------------------------------------------------------------------ parameters and weight images
    def bind(self, model, sites):
        """Register the model's parameters (flattened into one f32 buffer) and its conv sites."""
        self.params = list(model.parameters())
        self._flatten()
        self.sites = sites
        self.site_by_id = {id(s.conv): s for s in sites}
        total = 0
        offs = []
        for s in sites:
            offs.append(total)
            total += _ceil(s.fwd_numel * self.es, _ALIGN)
            for d in s.dg_numel:
                offs.append(total)
                total += _ceil(d * self.es, _ALIGN)
            s.frag, s.frag_numel = {}, {}
            if self.dt == F16 and self.blk3_on:
                s.plan_frag_images()
            if self.dt == F16 and self.blk4_on:
                s.plan_frag_images4()
            for key, numel in s.frag_numel.items():
                offs.append(total)
                total += _ceil(numel * 2, _ALIGN)
        self._img_buf = torch.zeros(max(total, 1), dtype=torch.uint8, device=self.device)
        base = self._img_buf.data_ptr()
        it = iter(offs)
        for s in sites:
            s.img_fwd = base + next(it)
            for k in range(len(s.seg_c)):
                o = next(it)
                s.img_dg[k] = base + o if s.dg_numel[k] else None
            for key in s.frag_numel:
                s.frag[key] = base + next(it)
        self._build_prep_table()
        self._weights_version = None
        # set by whoever writes the flat parameter buffer through raw pointers (the fused AdamW / EMA kernel, which
        # torch's version counters cannot see): the next prepare_weights() re-images even when no _version moved
        self.weights_dirty = True

    def _flatten(self):
        ps = self.params
        n = sum(p.numel() for p in ps)
        if self.flat_p is not None and self.flat_p.numel() == n and all(
                p.data_ptr() == self.flat_p.data_ptr() + 4 * self.p_off[id(p)] for p in ps):
            return False
        flat = torch.empty(n, dtype=torch.float32, device=self.device)
        off = 0
        self.p_off = {}
        for p in ps:
            assert p.dtype == torch.float32, "parameters stay f32 (master weights)"
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat[off:off + k].view(p.shape)
            self.p_off[id(p)] = off
            off += k
        self.flat_p = flat
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._red_tabs = {}
        return True

    def check_params(self):
        """Parameters may have been moved (model.to(), load_state_dict keeps storage): re-bind if so."""
        if self._flatten() or self._prep_src_sig != self._src_sig():
            self._build_prep_table()
            self._weights_version = None

    def _src_sig(self):
        return (self.flat_p.data_ptr(), self._img_buf.data_ptr())

    def _build_prep_table(self):
        descs, csite, cidx = [], [], []
        for s in self.sites:
            w = s.conv.weight
            d = _lib.WprepDesc()
            d.src, d.dst = w.data_ptr(), s.img_fwd
            d.co, d.ci_total, d.ks, d.mode, d.nseg, d.seg_off = s.co, s.ci, s.ks, 0, len(s.seg_c), 0
            for k, c in enumerate(s.seg_c):
                d.seg_c[k] = c
            d.dtype, d.rows_pad, d.k_pad, d.numel = self.dt, _ceil(s.co, 16), s.krow, s.fwd_numel
            descs.append(d)
            for k, c in enumerate(s.seg_c):
                if not s.dg_numel[k]:
                    continue
                d = _lib.WprepDesc()
                d.src, d.dst = w.data_ptr(), s.img_dg[k]
                d.co, d.ci_total, d.ks, d.mode, d.nseg, d.seg_off = s.co, s.ci, s.ks, 1, 1, s.seg_off[k]
                d.seg_c[0] = c
                d.dtype, d.rows_pad, d.k_pad, d.numel = self.dt, _ceil(c, 16), s.krow_dg, s.dg_numel[k]
                descs.append(d)
        for s in self.sites:  # fragment-ordered images of the fused Block kernel (cgen_weight_prep modes 2-5)
            for key, numel in s.frag_numel.items():
                w = s.conv.weight
                d = _lib.WprepDesc()
                d.src, d.dst = w.data_ptr(), s.frag[key]
                d.co, d.ci_total, d.ks, d.nseg, d.seg_off = s.co, s.ci, 3, len(s.seg_c), 0
                for k, c in enumerate(s.seg_c):
                    d.seg_c[k] = c
                d.dtype, d.rows_pad, d.numel = self.dt, 0, numel
                if key == "a_fwd":  # (k_pad: fragments per 32-row block of the bottleneck)
                    d.mode, d.k_pad = 2, sum(_ceil(c, 32) // 32 for c in s.seg_c) * 18
                elif key == "a_dg":
                    d.mode, d.k_pad = 3, _ceil(_ceil(s.co, 8), 32) // 32 * 18
                elif key == "a16_fwd":
                    d.mode, d.k_pad = 6, s.seg_c[0] // 32
                elif key == "a16_dg":
                    d.mode, d.k_pad = 7, s.co // 32
                elif key == "b_fwd":
                    d.mode, d.k_pad = 4, (9 * s.ci + 15) // 16
                elif isinstance(key, str) and key[0] == "p" or (isinstance(key, tuple) and key[0] == "p3_dg"):
                    g16 = _ceil(s.blk4[1][0].co, 16) // 16  # (16-channel groups of the bottleneck)
                    if key == "p0_fwd":
                        d.mode, d.k_pad = 8, 2 * sum(_ceil(c, 32) // 32 for c in s.seg_c)
                    elif key == "p0_dg":
                        d.mode, d.k_pad = 9, 2 * (_ceil(s.co, 32) // 32)
                    elif key in ("p1_fwd", "p2_fwd"):
                        d.mode, d.k_pad = 10, g16
                    elif key in ("p1_dg", "p2_dg"):
                        d.mode, d.k_pad = 11, g16
                    elif key == "p3_fwd":
                        d.mode, d.k_pad = 12, g16
                    else:
                        k = key[1]
                        d.mode, d.k_pad, d.nseg, d.seg_off = 13, g16, 1, s.seg_off[k]
                        d.seg_c[0] = s.seg_c[k]
                    d.ks = s.ks
                else:
                    k = key[1]
                    d.mode, d.k_pad, d.nseg, d.seg_off = 5, (9 * s.co + 15) // 16, 1, s.seg_off[k]
                    d.seg_c[0] = s.seg_c[k]
                descs.append(d)
        for i, d in enumerate(descs):
            nch = (d.numel + self.CHUNK - 1) // self.CHUNK
            csite += [i] * nch
            cidx += list(range(nch))
        arr = (_lib.WprepDesc * len(descs))(*descs)
        self._prep_tab = (self._to_dev(bytes(arr)), torch.tensor(csite, dtype=torch.int32, device=self.device),
                          torch.tensor(cidx, dtype=torch.int32, device=self.device), len(csite))
        self._prep_src_sig = self._src_sig()

    def _to_dev(self, raw):
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)

    def prepare_weights(self, force=False):
        """OIHW f32 parameters -> forward/dgrad weight images (one multi-tensor launch).  Skipped when no
        parameter changed since the last call (inference)."""
        ver = sum(p._version for p in self.params)
        if not force and ver == self._weights_version and not self.weights_dirty:
            return
        self.weights_dirty = False
        d, cs, ci, n = self._prep_tab
        self.lib.weight_prep(d.data_ptr(), cs.data_ptr(), ci.data_ptr(), n, self.stream)
        self.launches += 1
        self._weights_version = ver

    def param_nhwc(self, p):
        """Device pointer to the f32 [h][w][C] image of a [1,C,h,w] parameter (decoder.bias, vae.py:211-218)."""
        _, c, h, w = p.shape
        if h * w == 1 or c == 1:
            return p.data_ptr()
        ptr = self._pnhwc.get(id(p))
        if ptr is None:
            ptr = self.arena.alloc(c * h * w * 4)
            v = NT(ptr, 1, h, w, c, h * w * c, w * c, c, 4, rg=False)
            self.lib.nchw_to_nhwc(0, F32, 1, c, h, w, p.data_ptr(), v.cv(), 0.0, 1.0, self.stream)
            self.launches += 1
            self._pnhwc[id(p)] = ptr
            if self._in_side:
                # created lazily inside a side-stream section, cached for everybody: the main stream must not read it before
                # this conversion has run (found by tools/fuzz_model.py: two concurrent replays sharing the decoder biases)
                torch.cuda.current_stream(self.device).wait_stream(self._fwd_side)
        return ptr

    def param_grad_ptr(self, p):
        return self.flat_g.data_ptr() + 4 * self.p_off[id(p)]

    def param_grad_view(self, p):
        o = self.p_off[id(p)]
        return self.flat_g[o:o + p.numel()].view(p.shape)

    # ------------------------------------------------------------------ forward ops
    def conv(self, site, segs, act=ACT_NONE, res1=None, res2=None, out=None, tape_hold=None, trunk=False):
        """`tape_hold` (a list): the backward entry goes there instead of onto the tape -- for an op that is LAUNCHED early (on
        the side stream) but keeps its place in the backward order (the caller extends the tape with the list later).
        `trunk`: the result is the next value of the residual trunk (h = h + f(h)): on the f16 engine it gets a remainder
        plane, and a `res1` that has one is added with it (value = hi + remainder, ~22 significant bits)."""
        x0 = segs[0]
        if res2 is not None and res2.rem and (res1 is None or not res1.rem):
            res1, res2 = res2, res1  # (only res1 carries a remainder plane)
        if out is None:
            out = self.new(x0.n, x0.h, x0.w, site.co, rem=trunk and self.trunk_rem and max(x0.h, x0.w) <= self.trunk_maxres)
            if site.co % 8:  # ragged width (e.g. the 4-channel bottleneck of a 16-wide Block): the kernel zero-fills the
                out.cpad = _ceil(site.co, 8)  # padding channels, which keeps the tensor DMA-clean for its consumers
        assert out.c == site.co and len(segs) == len(site.seg_c)
        a = _lib.ConvArgs()
        gn, gh, gw, vw = self._geom(site.ks, list(segs) + [out, res1, res2])
        a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act, a.dact = self.dt, gn, gh, gw, site.ks, len(segs), act, 0
        if self.f32_split and not self.recording:
            a.dtype = F32S
        for k, s in enumerate(segs):
            assert s.c == site.seg_c[k] and (s.n, s.h, s.w) == (x0.n, x0.h, x0.w), (site.name, k, s.shape, site.seg_c)
            a.seg[k] = vw(s)
        a.weight = site.img_fwd
        b = site.conv.bias
        a.bias = b.data_ptr() if b is not None else None
        a.out = vw(out)
        a.aux = NULL_VIEW
        a.res1 = vw(res1)
        a.res2 = vw(res2)
        a.out_rem = out.rem
        a.res1_rem = res1.rem if res1 is not None else 0
        if self._ablate and ((site.name.endswith(".conv.1") and "f1" in self._ablate) or ("r12" in self._ablate and x0.h <= 12)
                             or any(x0.h == r and ("r%d" % r) in self._ablate for r in (24, 48, 96, 192))):
            pass  # TIMING-ONLY ablation (CGEN_ABLATE, wrong results): the first conv of every Block is not launched
        else:
            self._timed("conv_fwd", site, x0, lambda: self.lib.conv2d(C.byref(a), self.stream))
        if self.recording:
            if self._dbg_names is not None:
                self._dbg_names[id(out.base)] = site.name
            (self.tape if tape_hold is None else tape_hold).append((self._bw_conv, (site, segs, act, out, res1, res2), self._bw_tag))
        return out

    @staticmethod
    def _side_ranges(spec):
        """'24,48,96-112' -> [(24, 24), (48, 48), (96, 112)]; '0' or '' -> [] (no restriction)"""
        out = []
        for v in spec.split(","):
            v = v.strip()
            if not v or v == "0":
                continue
            lo, _, hi = v.partition("-")
            out.append((int(lo), int(hi or lo)))
        return out

    def block2(self, site1, site2, segs, act, res1=None, trunk=False):
        """A whole light Block -- conv3x3(act(cat segs)) -> conv3x3(act(.)) (+ res1), vae.py:60-71,73-84 -- as ONE launch of
        cgen_block3 (csrc/block.hip) where the kernel serves the shape and the policy (blk3_res / blk3_res3) takes it, else as the
        two conv launches.  The bottleneck tensor is written either way (weight gradients, backward mask)."""
        x0 = segs[0]
        res_ok = self.blk3_res3 if len(segs) >= 3 else self.blk3_res
        small = self.blk3_small and x0.w <= 14 and x0.h <= 64 and min(x0.h, x0.w) >= 4  # (the small-image instance: csrc/block.hip blk3s)
        tile_ok = (min(x0.h, x0.w) >= self.blk3_minres and site1.co <= 32
                   and (not res_ok or any(lo <= x0.h <= hi for lo, hi in res_ok)))
        # the row-streaming instance (blk3r: 96x96 / 192x192, one segment): taken where the kernel says it serves the shape with it,
        # whatever the side policy of the tile instance says
        rows = (self.blk3_rows and len(segs) == 1 and "a16_fwd" in site1.frag and "a16_dg" in site2.frag and x0.w >= 48 and x0.h >= 16
                and not (trunk and self.trunk_rem))
        if (self.blk3_on and act == ACT_RELU and "a_fwd" in site1.frag and "b_fwd" in site2.frag and len(segs) <= 3
                and site1.co % 8 == 0 and site2.co % 8 == 0
                and ((small and site1.co <= 64) or tile_ok or rows)):
            out = self._block3_fwd(site1, site2, segs, res1, trunk, need_rows=rows and not tile_ok and not small)
            if out is not None:
                return out
        t = self.conv(site1, segs, act)
        return self.conv(site2, [t], act, res1=res1, trunk=trunk)

    def _block3_fwd(self, site1, site2, segs, res1, trunk=False, need_rows=False):
        """One launch of cgen_block3 for a light Block (forward); None when the kernel declines the layout.  A residual-trunk Block
        of a pass with remainder planes (section 1a) reads res1 as hi + rem and writes out as rn16(v), rn16(v - out), like conv()."""
        x0 = segs[0]
        a = _lib.Block3Args()
        a.dtype, a.n, a.h, a.w, a.nseg, a.nout, a.pre_act = self.dt, x0.n, x0.h, x0.w, len(segs), 1, 1
        for k, sg in enumerate(segs):
            a.seg[k] = sg.cv()
        b1, b2 = site1.conv.bias, site2.conv.bias
        a.w_a, a.bias_a = site1.frag["a_fwd"], (b1.data_ptr() if b1 is not None else None)
        a.w_a16 = site1.frag.get("a16_fwd") if self.blk3_rows else None
        t = self.new(x0.n, x0.h, x0.w, site1.co)
        out = self.new(x0.n, x0.h, x0.w, site2.co, rem=trunk and self.trunk_rem and max(x0.h, x0.w) <= self.trunk_maxres)
        a.mid, a.mid_aux = t.cv(), NULL_VIEW
        o = a.o[0]
        o.w, o.bias = site2.frag["b_fwd"], (b2.data_ptr() if b2 is not None else None)
        o.out, o.aux, o.res1 = out.cv(), NULL_VIEW, (res1.cv() if res1 is not None else NULL_VIEW)
        o.out_rem, o.res1_rem = out.rem, (res1.rem if res1 is not None else 0)
        sup = self.lib.block3_supported(C.byref(a))
        if not sup or (need_rows and sup != 2):  # (2: the row-streaming instance serves it)
            return None  # (the two tensors just allocated are simply not used: the arena is reset per step)
        if self._ablate and any(x0.h == r and ("r%d" % r) in self._ablate for r in (24, 48, 96, 192)):
            pass
        else:
            self._timed_blk("conv_fwd", site1, site2, x0, lambda: self.lib.block3(C.byref(a), self.stream))
        if self.recording:
            if self._dbg_names is not None:
                self._dbg_names[id(out.base)] = site2.name
            if self.blk3_on >= 2:
                self.tape.append((self._bw_block3, (site1, site2, segs, t, out, res1), self._bw_tag))
            else:
                self.tape.append((self._bw_conv, (site1, segs, ACT_RELU, t, None, None), self._bw_tag))
                self.tape.append((self._bw_conv, (site2, [t], ACT_RELU, out, res1, None), self._bw_tag))
        return out

    def _bw_block3(self, site1, site2, segs, t, out, res1):
        """Backward of a fused light Block: weight gradients as for the two convs (deferred, batched); the two data-gradient
        convs as ONE launch of cgen_block3 (pre_act = 0) -- with two outputs when two segments need a gradient (the posterior
        Block: h and the encoder activation) -- else the two conv launches."""
        g = self.grad_read(out)
        if g is None:
            self._blk3_flush()
            return
        act = ACT_RELU
        dsegs = [k for k, sg in enumerate(segs) if sg.rg and site1.seg_rg[k]]
        ok = (1 <= len(dsegs) <= 2 and "a_dg" in site2.frag and all(("b_dg", k) in site1.frag for k in dsegs)
              and all(segs[k].c % 8 == 0 for k in dsegs) and g.c % 8 == 0)
        if ok:
            probe = _lib.Block3Args()
            probe.dtype, probe.n, probe.h, probe.w, probe.nseg, probe.nout, probe.pre_act = self.dt, g.n, g.h, g.w, 1, len(dsegs), 0
            probe.seg[0] = g.cv()
            probe.w_a, probe.bias_a = site2.frag["a_dg"], None
            probe.w_a16 = site2.frag.get("a16_dg") if self.blk3_rows else None
            probe.mid, probe.mid_aux = t.cv(), t.cv()
            for j, k in enumerate(dsegs):
                sg = segs[k]
                probe.o[j].w, probe.o[j].bias = site1.frag[("b_dg", k)], None
                probe.o[j].out, probe.o[j].aux, probe.o[j].res1 = sg.cv(), sg.cv(), NULL_VIEW
            ok = bool(self.lib.block3_supported(C.byref(probe)))
        if not ok:
            self._blk3_flush()
            self._bw_conv(site2, [t], act, out, res1, None)
            self._bw_conv(site1, segs, act, t, None, None)
            return
        if res1 is not None and res1.rg:
            self._blk3_flush()  # (a residual Block is never the second of a pair: its residual gradient is a launch of its own)
            self._grad_residual(res1, g, out, [t])
        if self._needs_wgrad(site2) and "wg" not in self._ablate:
            self._wgrad(site2, [t], act, g)
        gt, acc_t = self.grad_write(t)
        assert not acc_t
        a = _lib.Block3Args()
        a.dtype, a.n, a.h, a.w, a.nseg, a.nout, a.pre_act = self.dt, g.n, g.h, g.w, 1, len(dsegs), 0
        a.seg[0] = g.cv()
        a.w_a, a.bias_a = site2.frag["a_dg"], None
        a.w_a16 = site2.frag.get("a16_dg") if self.blk3_rows else None
        a.mid, a.mid_aux = gt.cv(), t.cv()
        tgt = []
        for j, k in enumerate(dsegs):
            sg = segs[k]
            gv, prev, acc = self._dgrad_target(sg)
            tgt.append((k, gv, prev, acc))
            a.o[j].w, a.o[j].bias = site1.frag[("b_dg", k)], None
            a.o[j].out, a.o[j].aux = gv.cv(), sg.cv()
            a.o[j].res1 = prev.cv() if acc else NULL_VIEW
        x0 = segs[0]
        late = (lambda: self._wgrad(site1, segs, act, gt)) if (self._needs_wgrad(site1) and "wg" not in self._ablate) else (lambda: None)
        if self._ablate and any(x0.h == r and ("r%d" % r) in self._ablate for r in (24, 48, 96, 192)):
            self._blk3_flush()
        elif self.lib.block3_supported(C.byref(a)):
            # tensors this launch writes / reads (storage identity): a pair must not touch each other's
            wr = {id(gt.base)} | {id(gv.base) for (_, gv, _, _) in tgt}
            rd = {id(g.base), id(t.base)} | {id(segs[k].base) for (k, _, _, _) in tgt} | {id(prev.base) for (_, _, prev, acc) in tgt if acc}
            if self._blk3_arm == 1 and self.prof is None:
                # held: the partner launches both.  The weight gradient that READS gt is queued only after that launch (a background
                # flush triggered in between must not see it)
                self._blk3_hold = (a, self.launches, wr, rd, late, (site1, site2, x0))
                self._blk3_arm = 2
                return
            if self._blk3_hold is not None:
                ha, hl, hwr, hrd, hlate, hinfo = self._blk3_hold
                if (hl == self.launches and not (hwr & (wr | rd)) and not (wr & hrd)
                        and self.lib.block3_pair_supported(C.byref(ha), C.byref(a))):
                    self._blk3_hold, self._blk3_arm = None, 0
                    self.lib.block3_pair(C.byref(ha), C.byref(a), self.stream)
                    self.launches += 1
                    self.blk3_pairs += 1
                    hlate()
                    late()
                    return
                self._blk3_flush()
            self._timed_blk("conv_dgrad", site1, site2, x0, lambda: self.lib.block3(C.byref(a), self.stream))
        else:
            # The REAL gradient views are not served (the probe above saw the forward tensors' views; a gradient buffer may be laid
            # out differently: an out-of-place accumulate target, a copy-on-write -- ADVICE r4): the two data-gradient convs, into
            # the targets already acquired (the bookkeeping above has run and must not run twice)
            self._blk3_flush()
            self._dgrad_launch(site2, g, t, 0, act, t, gt, gt, False)
            for (k, gv, prev, acc) in tgt:
                self._dgrad_launch(site1, gt, segs[k], k, act, x0, gv, prev, acc)
        late()

    def block4(self, sites, segs, res1=None, trunk=False):
        """A whole default Block -- 1x1(gelu(cat segs)) -> 3x3(gelu) -> 3x3(gelu) -> 1x1(gelu) (+ res1), vae.py:57-71,73-84 -- as ONE
        launch of cgen_block4 (csrc/block4.hip); None when the kernel does not serve the shape (the caller then runs the four convs).
        The three bottleneck tensors are written (pre-activations: weight gradients, backward)."""
        s0, s1, s2, s3 = sites
        x0 = segs[0]
        if not (self.blk4_on and "p0_fwd" in s0.frag and "p3_fwd" in s3.frag and len(segs) <= 3 and s3.co % 8 == 0 and s3.co <= 256):
            return None
        a = _lib.Block4Args()
        a.dtype, a.n, a.h, a.w, a.nseg, a.nout, a.fwd, a.b = self.dt, x0.n, x0.h, x0.w, len(segs), 1, 1, s0.co
        for k, sg in enumerate(segs):
            assert sg.c == s0.seg_c[k] and (sg.n, sg.h, sg.w) == (x0.n, x0.h, x0.w), (s0.name, k, sg.shape, s0.seg_c)
            a.seg[k] = sg.cv()
        ts = []
        for k, (st, key) in enumerate(((s0, "p0_fwd"), (s1, "p1_fwd"), (s2, "p2_fwd"))):
            a.wimg[k] = st.frag[key]
            bp = st.conv.bias
            a.bias[k] = bp.data_ptr() if bp is not None else None
            t = self.new(x0.n, x0.h, x0.w, s0.co)
            if s0.co % 8:
                t.cpad = _ceil(s0.co, 8)  # (the kernel writes whole 8-channel groups: zeros in the padding)
            a.mid[k], a.mid_aux[k] = t.cv(), NULL_VIEW
            ts.append(t)
        out = self.new(x0.n, x0.h, x0.w, s3.co, rem=trunk and self.trunk_rem and max(x0.h, x0.w) <= self.trunk_maxres)
        o = a.o[0]
        b3 = s3.conv.bias
        o.w, o.bias = s3.frag["p3_fwd"], (b3.data_ptr() if b3 is not None else None)
        o.out, o.aux, o.res1 = out.cv(), NULL_VIEW, (res1.cv() if res1 is not None else NULL_VIEW)
        o.out_rem, o.res1_rem = out.rem, (res1.rem if res1 is not None else 0)
        if not self.lib.block4_supported(C.byref(a)):
            return None
        self._timed_blk4("conv_fwd", sites, x0, lambda: self.lib.block4(C.byref(a), self.stream))
        self.blk4_launches += 1
        if self.recording:
            if self._dbg_names is not None:
                self._dbg_names[id(out.base)] = s3.name
            self.tape.append((self._bw_block4, (sites, segs, tuple(ts), out, res1), self._bw_tag))
        return out

    def _bw_block4(self, sites, segs, ts, out, res1):
        """Backward of a fused default Block: weight gradients as for the four convs (deferred, batched); the four data-gradient
        convs + GELU derivatives as ONE launch of cgen_block4 (fwd = 0), with one output per differentiable input segment -- or half of
        a pair launch with the layer's other Block (cgen_block4_pair; armed by backward())."""
        g = self.grad_read(out)
        if g is None:
            self._blk4_flush()
            return
        s0, s1, s2, s3 = sites
        t0, t1, t2 = ts
        act = ACT_GELU
        x0 = segs[0]
        dsegs = [k for k, sg in enumerate(segs) if sg.rg and s0.seg_rg[k]]
        ok = (1 <= len(dsegs) <= 3 and "p0_dg" in s3.frag and "p1_dg" in s2.frag and "p2_dg" in s1.frag
              and all(("p3_dg", k) in s0.frag for k in dsegs) and all(segs[k].c % 8 == 0 for k in dsegs) and g.c % 8 == 0)
        if not ok:
            self._blk4_flush()
            self._bw_conv(s3, [t2], act, out, res1, None)
            self._bw_conv(s2, [t1], act, t2, None, None)
            self._bw_conv(s1, [t0], act, t1, None, None)
            self._bw_conv(s0, segs, act, t0, None, None)
            return
        if res1 is not None and res1.rg:
            self._blk4_flush()  # (a residual Block is never the second of a pair: its residual gradient is a launch of its own)
            # (the fused launch reads grad(out) WITH a halo while it writes grad(x): x may adopt grad(out)'s buffer only when its
            #  later accumulation goes out of place -- which is the case exactly when weight gradients are deferred, _dgrad_target)
            self._grad_residual(res1, g, out, [t2] if self._defer_wgrad() else [t2] + list(segs))
        wg = "wg" not in self._ablate
        if wg and self._needs_wgrad(s3):
            self._wgrad(s3, [t2], act, g)
        g2, acc2 = self.grad_write(t2)
        g1, acc1 = self.grad_write(t1)
        g0, acc0 = self.grad_write(t0)
        assert not (acc0 or acc1 or acc2)
        a = _lib.Block4Args()
        a.dtype, a.n, a.h, a.w, a.nseg, a.nout, a.fwd, a.b = self.dt, g.n, g.h, g.w, 1, len(dsegs), 0, s0.co
        a.seg[0] = g.cv()
        for k, (st, key, gm, tm) in enumerate(((s3, "p0_dg", g2, t2), (s2, "p1_dg", g1, t1), (s1, "p2_dg", g0, t0))):
            a.wimg[k], a.bias[k] = st.frag[key], None
            a.mid[k], a.mid_aux[k] = gm.cv(), tm.cv()
        tgt = []
        for j, k in enumerate(dsegs):
            sg = segs[k]
            gv, prev, acc = self._dgrad_target(sg)
            tgt.append((k, gv, prev, acc))
            a.o[j].w, a.o[j].bias = s0.frag[("p3_dg", k)], None
            a.o[j].out, a.o[j].aux = gv.cv(), sg.cv()
            a.o[j].res1 = prev.cv() if acc else NULL_VIEW
        def late():  # the weight gradients that READ what the launch writes: queued only behind it (a background flush must not see them earlier)
            if wg:
                if self._needs_wgrad(s2):
                    self._wgrad(s2, [t1], act, g2)
                if self._needs_wgrad(s1):
                    self._wgrad(s1, [t0], act, g1)
                if self._needs_wgrad(s0):
                    self._wgrad(s0, segs, act, g0)

        if self.lib.block4_supported(C.byref(a)):
            # tensors this launch writes / reads (storage identity): a pair must not touch each other's
            wr = {id(g2.base), id(g1.base), id(g0.base)} | {id(gv.base) for (_, gv, _, _) in tgt}
            rd = ({id(g.base), id(t0.base), id(t1.base), id(t2.base)} | {id(segs[k].base) for (k, _, _, _) in tgt}
                  | {id(prev.base) for (_, _, prev, acc) in tgt if acc})
            if self._blk4_arm == 1 and self.prof is None:
                self._blk4_hold = (a, self.launches, wr, rd, late, (sites, x0))  # held: the partner launches both
                self._blk4_arm = 2
                return
            if self._blk4_hold is not None:
                ha, hl, hwr, hrd, hlate, _ = self._blk4_hold
                if (hl == self.launches and not (hwr & (wr | rd)) and not (wr & hrd)
                        and self.lib.block4_pair_supported(C.byref(ha), C.byref(a))):
                    self._blk4_hold, self._blk4_arm = None, 0
                    self.lib.block4_pair(C.byref(ha), C.byref(a), self.stream)
                    self.launches += 1
                    self.blk4_pairs += 1
                    self.blk4_launches += 1
                    hlate()
                    late()
                    return
                self._blk4_flush()
            self._timed_blk4("conv_dgrad", sites, x0, lambda: self.lib.block4(C.byref(a), self.stream))
            self.blk4_launches += 1
        else:
            # the REAL gradient views are not served (an oddly laid-out accumulate target): the four data-gradient convs, into the
            # targets already acquired (the bookkeeping above has run and must not run twice)
            self._blk4_flush()
            self._dgrad_launch(s3, g, t2, 0, act, t2, g2, g2, False)
            self._dgrad_launch(s2, g2, t1, 0, act, t1, g1, g1, False)
            self._dgrad_launch(s1, g1, t0, 0, act, t0, g0, g0, False)
            for (k, gv, prev, acc) in tgt:
                self._dgrad_launch(s0, g0, segs[k], k, act, x0, gv, prev, acc)
            self._conv_flush()
        late()

    def _blk4_flush(self):
        """Launch a held fused default-Block data gradient on its own (its partner did not come, or cannot share the launch)."""
        self._blk4_arm = 0
        if self._blk4_hold is None:
            return
        ha, _, _, _, hlate, (sites, x0) = self._blk4_hold
        self._blk4_hold = None
        self._timed_blk4("conv_dgrad", sites, x0, lambda: self.lib.block4(C.byref(ha), self.stream))
        self.blk4_launches += 1
        hlate()

    def _timed_blk4(self, kind, sites, x0, fn):
        """A fused default-Block launch, tallied (when profiling) with the algorithmic FLOPs of the FOUR convs it executes."""
        self.launches += 1
        if self.prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        flops = 2.0 * sum(st.ci * st.taps * st.co for st in sites) * x0.n * x0.h * x0.w
        ent = self.prof.setdefault((kind + "_blk", 3, sites[0].ci, sites[3].co, x0.h), [0.0, [], 0])
        ent[0] += flops
        ent[1].append((e0, e1))
        ent[2] += 1

    def _blk3_flush(self):
        """Launch a held fused data gradient on its own (its partner did not come, or cannot share the launch)."""
        self._blk3_arm = 0
        if self._blk3_hold is None:
            return
        ha, _, _, _, hlate, (s1, s2, x0) = self._blk3_hold
        self._blk3_hold = None
        self._timed_blk("conv_dgrad", s1, s2, x0, lambda: self.lib.block3(C.byref(ha), self.stream))
        hlate()

    def _timed(self, kind, site, x0, fn, ci=None):
        """Launch `fn`; when profiling, bracket it with events on the launch stream and tally algorithmic FLOPs
        (2 * Ci * k*k * Co per output pixel) under (kind, shape)."""
        self.launches += 1
        if self.prof is None:
            return fn()
        ci = site.ci if ci is None else ci
        flops = 2.0 * ci * site.taps * site.co * x0.n * x0.h * x0.w
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        key = (kind, site.ks, ci, site.co, x0.h)
        ent = self.prof.setdefault(key, [0.0, [], 0])
        ent[0] += flops
        ent[1].append((e0, e1))
        ent[2] += 1

    def _timed_blk(self, kind, site1, site2, x0, fn):
        """A fused Block launch, tallied (when profiling) with the algorithmic FLOPs of BOTH convs it executes."""
        self.launches += 1
        if self.prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        flops = 2.0 * (site1.ci * site1.taps * site1.co + site2.ci * site2.taps * site2.co) * x0.n * x0.h * x0.w
        ent = self.prof.setdefault((kind + "_blk", 3, site1.ci, site2.co, x0.h), [0.0, [], 0])
        ent[0] += flops
        ent[1].append((e0, e1))
        ent[2] += 1

    def stem(self, site, x):
        """Encoder.stem (vae.py:104-110,126): the direct 7x7 kernel (csrc/elementwise.hip) for a 7x7 site -- its weight
        gradient reads x itself through the tiled kernel's 7x7 instance (bf16) or the generic kernel (f32) -- else im2col +
        1x1 conv over the 49 * Cin patch channels."""
        if site.im2col:
            return self.conv(site, [self.im2col(x, site.im2col)], ACT_NONE)
        out = self.new(x.n, x.h, x.w, site.co)
        b = site.conv.bias
        self._timed("conv_fwd", site, x, lambda: self.lib.stem_conv_fwd(
            self.dt, x.n, x.h, x.w, site.ci, site.ks, site.co, x.cv(), site.conv.weight.data_ptr(), b.data_ptr() if b is not None else None,
            out.cv(), self.stream))
        if self.recording:
            self.tape.append((self._bw_conv, (site, [x], ACT_NONE, out, None, None)))
        return out

    def im2col(self, x, ks):
        c = x.c * ks * ks
        out = self.new(x.n, x.h, x.w, c, rg=False)
        out.cpad = _ceil(c, 8)
        self.lib.im2col(self.dt, x.n, x.h, x.w, ks, x.cv(), out.cv(), self.stream)
        self.launches += 1
        return out

    def pool(self, x, d):
        """F.avg_pool2d(kernel = stride = d) for an integer down-rate, F.adaptive_avg_pool2d(int(W / d)) for a float one
        (vae.py:79-83)."""
        if isinstance(d, float):
            ho, wo = int(x.w / d), int(x.w / d)  # (the reference sizes both axes from the last one)
            out = self.new(x.n, ho, wo, x.c)
            self.lib.adaptive_avgpool_fwd(self.dt, x.n, x.h, x.w, ho, wo, x.cv(), out.cv(), self.stream)
        else:
            out = self.new(x.n, x.h // d, x.w // d, x.c)
            self.lib.avgpool_fwd(self.dt, x.n, out.h, out.w, d, x.cv(), out.cv(), self.stream)
        self.launches += 1
        if self.recording:
            self.tape.append((self._bw_pool, (x, out, d)))
        return out

    def upsample(self, x, res, bias_param=None):
        out = self.new(x.n, res, res, x.c)
        self.lib.upsample_fwd(self.dt, x.n, x.h, x.w, res, res, x.cv(), self.param_nhwc(bias_param) if bias_param is not None else None,
                              out.cv(), self.stream)
        self.launches += 1
        if self.recording:
            self.tape.append((self._bw_upsample, (x, out, bias_param)))
        return out

    def bcast(self, param, n):
        """param [1,C,h,w] repeated over the batch (vae.py:233)."""
        _, c, h, w = param.shape
        out = self.new(n, h, w, c)
        self.lib.batch_broadcast(self.dt, n, h, w, self.param_nhwc(param), out.cv(), self.stream)
        self.launches += 1
        if self.recording:
            self.tape.append((self._bw_bcast, (param, out)))
        return out

    def pad_br(self, x):
        """F.pad(x, [0,1,0,1]) (vae.py:131-133)."""
        out = self.new(x.n, x.h + 1, x.w + 1, x.c)
        self.fill(out, 0.0)
        inner = NT(out.ptr, x.n, x.h, x.w, x.c, out.sn, out.sh, out.sw, out.es, rg=False)
        self.lib.axpby(self.dt, x.n, x.h, x.w, x.cv(), inner.cv(), 1.0, 1.0, 1 << 30, 0, self.stream)
        self.launches += 1
        if self.recording:
            self.tape.append((self._bw_pad, (x, out)))
        return out

    @staticmethod
    def _padded(x):
        """View of x widened to its physical (8-aligned) channel count."""
        return NT(x.ptr, x.n, x.h, x.w, _ceil(x.c, 8), x.sn, x.sh, x.sw, x.es, rg=False)

    def fill(self, x, value):
        self.lib.axpby(self.dt, x.n, x.h, x.w, NULL_VIEW, x.cv(), float(value), 1.0, 1 << 30, 0, self.stream)
        self.launches += 1

    def scale_channels(self, x, c_from, factor):
        """out = x with channels >= c_from multiplied by factor (pa_sto, vae.py:244-247)."""
        if x.bsrc is not None:  # spatially constant: scale the [n,1,1,c] source, broadcast again
            return self.scale_channels(x.bsrc, c_from, factor).broadcast(x.h, x.w)
        out = self.new(x.n, x.h, x.w, x.c, rg=False)
        src, dst = x, out
        if x.cpad:  # carry the zero padding along
            src, dst = self._padded(x), self._padded(out)
            out.cpad = x.cpad
        self.lib.axpby(self.dt, x.n, x.h, x.w, src.cv(), dst.cv(), 1.0, float(factor), c_from, 0, self.stream)
        self.launches += 1
        return out

    def reparam_kl(self, q_loc, q_ls, p_loc, p_ls, eps, stream_id, logt, kl_ptr, kl_stride, fb=None):
        """`fb` = (S_ptr, S_stride, column offset) when free bits are on: per-(sample, channel) KL sums of this layer go to
        S[b*S_stride + col + ch], and the backward pass scales this layer's KL gradient by kl_chan_ptr[col + ch]."""
        z = self.new(q_loc.n, q_loc.h, q_loc.w, q_loc.c)
        if fb is not None:
            self.lib.kl_channel_sums(self.dt, z.n, z.h, z.w, z.c, q_loc.cv(), q_ls.cv(), p_loc.cv(), p_ls.cv(), logt,
                                     fb[0] + 4 * fb[2], fb[1], self.stream)
            self.launches += 1
        self.lib.reparam_kl_fwd(self.dt, z.n, z.h, z.w, z.c, q_loc.cv(), q_ls.cv(), p_loc.cv(), p_ls.cv(),
                                eps.cv() if eps is not None else NULL_VIEW, self.rng_ptr(), stream_id, logt, z.cv(), NULL_VIEW,
                                kl_ptr, kl_stride, self.stream)
        self.launches += 1
        if self.recording:
            self.tape.append((self._bw_reparam, (q_loc, q_ls, p_loc, p_ls, z, logt, None if fb is None else fb[2], self.kl_coef_override)))
        return z

    def sample_gaussian(self, loc, ls, eps, stream_id, logt):
        z = self.new(loc.n, loc.h, loc.w, loc.c)
        self.lib.sample_gaussian(self.dt, z.n, z.h, z.w, z.c, loc.cv(), ls.cv(), eps.cv() if eps is not None else NULL_VIEW,
                                 self.rng_ptr(), stream_id, logt, z.cv(), self.stream)
        self.launches += 1
        return z

    def rng_ptr(self):
        if self.rng is None:
            self.rng = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=self.device)
            self._rng_next = torch.empty_like(self.rng)
            self._rng_one = torch.tensor([0, 1], dtype=torch.int64, device=self.device)
        return self._rng_override if self._rng_override is not None else self.rng.data_ptr()

    def rng_next_ptr(self):
        """Pointer to a copy of the Philox state one pass ahead ([seed, offset + 1]): the second of two passes that run
        concurrently draws what it would have drawn had it run after the first (HVAE.forward_latents_pair)."""
        self.rng_ptr()
        torch.add(self.rng, self._rng_one, out=self._rng_next)
        return self._rng_next.data_ptr()

    def rng_advance(self, inc=1):
        self.lib.rng_advance(self.rng_ptr(), inc, self.stream)
        self.launches += 1

    # ------------------------------------------------------------------ gradient bookkeeping
    def _geom(self, ks, ts):
        """Launch geometry (n, h, w) and the view constructor for a conv over tensors `ts` (None entries allowed).
        A 1x1 conv does not care about the spatial structure: on tiny images (< 5x5, where the tiled kernels do not apply)
        pixel-contiguous tensors are presented as ONE image of 16-pixel rows, [1, P/16, 16, C], which the tiled /
        persistent kernels (and the packed weight-gradient launch) serve."""
        t0 = next(t for t in ts if t is not None)
        n, h, w = t0.n, t0.h, t0.w
        P = n * h * w
        if ks == 1 and h * w < 25 and P % 16 == 0 and P >= 256 and all(
                t is None or (t.sh == t.w * t.sw and t.sn == t.h * t.sh) for t in ts):
            return 1, P // 16, 16, (lambda t: NULL_VIEW if t is None else View(t.ptr, P * t.sw, 16 * t.sw, t.sw, t.c, t.cpad))
        return n, h, w, (lambda t: NULL_VIEW if t is None else t.cv())

    def _new_grad(self, n, h, w, c):
        g = self.new(n, h, w, c, rg=False)
        if c % 8:  # ragged width: zero the padding once (writers only touch [0, c)) so gradient consumers can DMA it
            self.fill(self._padded(g), 0.0)
            g.cpad = _ceil(c, 8)
        return g

    def _gentry(self, base):
        e = self.grads.get(id(base))
        if e is None:
            g = self._new_grad(base.n, base.h, base.w, base.c)
            e = [g, [], base]
            self.grads[id(base)] = e
        return e

    @staticmethod
    def _missing(ivs, a, b):
        out, cur = [], a
        for (s, e) in sorted(ivs):
            if e <= cur:
                continue
            if s >= b:
                break
            if s > cur:
                out.append((cur, min(s, b)))
            cur = max(cur, e)
            if cur >= b:
                break
        if cur < b:
            out.append((cur, b))
        return out

    # ------------------------------------------------------------------ ops used only by the config-1 model (simple_vae)
    def unary(self, x, op, param=0.0):
        """Stand-alone activation / clamp (ACT_* or UNARY_*), differentiable."""
        out = self.new(x.n, x.h, x.w, x.c, rg=x.rg)
        self.lib.unary_fwd(self.dt, op, float(param), x.n, x.h, x.w, x.c, x.cv(), out.cv(), self.stream)
        self.launches += 1
        if self.recording and x.rg:
            self.tape.append((self._bw_unary, (x, out, op, float(param))))
        return out

    def _bw_unary(self, x, out, op, param):
        g = self.grad_read(out)
        if g is None:
            return
        gv, acc = self.grad_write(x)
        self.lib.unary_bwd(self.dt, op, param, x.n, x.h, x.w, x.c, x.cv(), g.cv(), gv.cv(), 1 if acc else 0, self.stream)
        self.launches += 1

    def im2col_strided(self, x, ks, stride, pad):
        """[N,H,W,C] -> [N,Ho,Wo,C*ks*ks] patches (channel = c*ks*ks + tap); differentiable (col2im)."""
        ho, wo = (x.h + 2 * pad - ks) // stride + 1, (x.w + 2 * pad - ks) // stride + 1
        out = self.new(x.n, ho, wo, x.c * ks * ks, rg=x.rg)
        out.cpad = _ceil(out.c, 8)  # the kernel zeroes the padding channels of every pixel
        self.lib.im2col_strided(self.dt, x.n, x.h, x.w, ks, stride, pad, ho, wo, x.cv(), out.cv(), self.stream)
        self.launches += 1
        if self.recording and x.rg:
            self.tape.append((self._bw_im2col_strided, (x, out, ks, stride, pad)))
        return out

    def _bw_im2col_strided(self, x, out, ks, stride, pad):
        g = self.grad_read(out)
        if g is None:
            return
        gv, acc = self.grad_write(x)
        self.lib.col2im_strided(self.dt, x.n, x.h, x.w, ks, stride, pad, out.h, out.w, g.cv(), gv.cv(), 1 if acc else 0, self.stream)
        self.launches += 1

    def flatten_chw(self, x):
        """[N,H,W,C] -> [N,1,1,C*H*W] in (c, y, x) order: what ``t.reshape(N, -1)`` does to an NCHW tensor
        (simple_vae.py:62).  f32 engines only (the layout kernels convert to / from f32 NCHW)."""
        assert self.dt == F32, "flatten_chw: f32 engine only"
        k = x.c * x.h * x.w
        out = self.new(x.n, 1, 1, k, rg=x.rg)
        assert out.sn == k, "flatten_chw needs an unpadded vector (C*H*W multiple of 8)"
        self.lib.nhwc_to_nchw(self.dt, x.n, x.c, x.h, x.w, x.cv(), out.ptr, self.stream)
        self.launches += 1
        if self.recording and x.rg:
            self.tape.append((self._bw_flatten_chw, (x, out)))
        return out

    def _bw_flatten_chw(self, x, out):
        g = self.grad_read(out)
        if g is None:
            return
        gv, acc = self.grad_write(x)
        assert not acc and g.sn == x.c * x.h * x.w and gv.coff == 0 and gv.c == x.c
        self.lib.nchw_to_nhwc(0, self.dt, x.n, x.c, x.h, x.w, g.ptr, gv.cv(), 0.0, 1.0, self.stream)
        self.launches += 1

    def unflatten_chw(self, v, c, h, w):
        """[N,1,1,C*H*W] -> [N,H,W,C], the inverse of flatten_chw (``t.reshape(N, -1, 4, 4)``, simple_vae.py:310)."""
        assert self.dt == F32 and v.c == c * h * w and v.sn == v.c
        out = self.new(v.n, h, w, c, rg=v.rg)
        self.lib.nchw_to_nhwc(0, self.dt, v.n, c, h, w, v.ptr, out.cv(), 0.0, 1.0, self.stream)
        self.launches += 1
        if self.recording and v.rg:
            self.tape.append((self._bw_unflatten_chw, (v, out)))
        return out

    def _bw_unflatten_chw(self, v, out):
        g = self.grad_read(out)
        if g is None:
            return
        gv, acc = self.grad_write(v)
        assert not acc and gv.sn == v.c and gv.coff == 0 and gv.c == v.c
        self.lib.nhwc_to_nchw(self.dt, out.n, out.c, out.h, out.w, g.cv(), gv.ptr, self.stream)
        self.launches += 1

    # ------------------------------------------------------------------ two-stream sections of the forward pass
    def fork_mark(self):
        """An event on the main stream: `fork_side(after=ev)` later makes the side stream depend on the work up to HERE only, so the
        main stream's next launch can be enqueued BEFORE the side stream's.  That order matters under a hipGraph: the executor walks
        the captured graph depth-first along a node's edges in capture order, and the chain it follows first keeps the queue -- and
        owns the node where the two chains join.  With the side chain (z_feat_proj -> prior Block) captured first, the join
        (reparameterise + KL) landed on ITS queue and the critical chain (z_proj -> conv Block -> posterior Block) crossed queues
        twice per decoder layer, ~11 us each (LABNOTES 9.7)."""
        if not self.fwd_branch or self.prof is not None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def fork_side(self, after=None):
        """Returns True when a side stream is available; everything enqueued so far (up to the mark `after`) is visible to it."""
        if not self.fwd_branch or self.prof is not None:
            return False
        if self._fwd_side is None:
            self._fwd_side = torch.cuda.Stream(self.device)
        if after is not None:
            self._fwd_side.wait_event(after)
        else:
            self._fwd_side.wait_stream(torch.cuda.current_stream(self.device))
        return True

    def on_side(self, fn, bw=0):
        """Run `fn()` with every launch going to the side stream.  `bw` = 1: the backward of what fn records belongs to the side
        strand of backward() too (see _bw_tag)."""
        old, old_tag = self.stream, self._bw_tag
        self.stream = self._fwd_side.cuda_stream
        self._in_side = True
        self._bw_tag = bw
        try:
            return fn()
        finally:
            self.stream = old
            self._in_side = False
            self._bw_tag = old_tag

    def join_side(self):
        torch.cuda.current_stream(self.device).wait_stream(self._fwd_side)
        self._side_join_pending = False

    def _defer_wgrad(self):
        return self.wgrad_batch or (self.wgrad_streams > 1 and self.prof is None)

    def _frozen(self, t):
        """True when grad(t) lives in a buffer a deferred weight-gradient kernel will still read: only ADOPTED buffers
        (see _grad_residual) are ever written after their producer's backward ran.  Such a buffer must not be
        accumulated into in place."""
        e = self.grads.get(id(t.base))
        return self._defer_wgrad() and e is not None and id(e[0]) in self._adopted

    def _cow(self, t):
        """Copy-on-write of a frozen gradient buffer (rare: only non-conv ops accumulating into an adopted buffer)."""
        e = self.grads[id(t.base)]
        old, base = e[0], e[2]
        new = self._new_grad(base.n, base.h, base.w, base.c)
        self.lib.axpby(self.dt, old.n, old.h, old.w, old.cv(), new.cv(), 1.0, 1.0, 1 << 30, 0, self.stream)
        self.launches += 1
        e[0] = new

    def _land_rider(self, base):
        """A copy parked for the next reparam backward (see _grad_residual) must land before anyone else touches grad(base)."""
        job = self._riders.pop(id(base), None)
        if job is not None:
            gv, g, acc = job
            self.lib.axpby(self.dt, g.n, g.h, g.w, g.cv(), gv.cv(), 1.0, 1.0, 1 << 30, 1 if acc else 0, self.stream)
            self.launches += 1

    def grad_write(self, t, defer_hazard=False):
        """Gradient view for `t` plus whether it already holds a value (=> the writer must accumulate)."""
        assert t.rg
        if self._riders:
            self._land_rider(t.base)
        g, ivs, base = self._gentry(t.base)
        a, b = t.coff, t.coff + t.c
        miss = self._missing(ivs, a, b)
        if miss != [(a, b)] and not defer_hazard and self._frozen(t):
            self._cow(t)
            g = self.grads[id(t.base)][0]
        gv = g.chan(a, b)
        if not miss:
            return gv, True
        if miss == [(a, b)]:
            ivs.append((a, b))
            return gv, False
        for (s, e) in miss:  # partially initialised: zero the gaps, then accumulate
            self.fill(g.chan(s, e), 0.0)
            ivs.append((s, e))
        return gv, True

    def grad_read(self, t):
        """Gradient of `t` for reading (zero-filling channels nobody wrote); None when nothing flowed into it."""
        if self._riders:
            self._land_rider(t.base)
        e = self.grads.get(id(t.base))
        if e is None:
            return None
        g, ivs, _ = e
        a, b = t.coff, t.coff + t.c
        miss = self._missing(ivs, a, b)
        if miss == [(a, b)]:
            return None
        for (s, e2) in miss:
            self.fill(g.chan(s, e2), 0.0)
            ivs.append((s, e2))
        return g.chan(a, b)

    def grad_add(self, t, g, may_ride=False):
        gv, acc = self.grad_write(t)
        if (may_ride and self.ride and self.dt == F16 and self._defer_wgrad() and t.base is not t and id(t.base) not in self._riders
                and t.c % 8 == 0 and self._v16(gv) and self._v16(g) and id(self.grads[id(t.base)][0]) not in self._adopted):
            # channel slice of a wider gradient buffer (the prior Block's output): the copy rides on the reparam backward
            # that writes the neighbouring channels (grad buffers are immutable under deferral, so `g` can wait)
            self._riders[id(t.base)] = (gv, g, acc)
            return
        self.lib.axpby(self.dt, g.n, g.h, g.w, g.cv(), gv.cv(), 1.0, 1.0, 1 << 30, 1 if acc else 0, self.stream)
        self.launches += 1

    @staticmethod
    def _v16(v):
        """16-byte vector access is legal on this bf16 view."""
        return v.es == 2 and v.ptr % 16 == 0 and v.sn % 8 == 0 and v.sh % 8 == 0 and v.sw % 8 == 0 and v.n * v.sn * 2 < (1 << 31)

    def _grad_residual(self, r, g, out, segs):
        """d(out)/d(residual) = identity.  When `r` is a whole tensor that has no gradient yet and `out`'s gradient is a
        whole buffer, r simply ADOPTS that buffer (no copy): every later contribution to grad(r) is accumulated in place
        by a kernel epilogue, after this op's own reads of it have been enqueued (same stream => ordered).
        Not when r is also an input segment of this op (its dgrad would read the buffer with a halo while accumulating
        into it), and only one tensor may adopt a given buffer."""
        gbuf = self.grads[id(out.base)][0]
        whole_r = r.base is r and id(r) not in self.grads
        whole_g = out.base is out and g.c == out.c
        clash = any(sg.base is r for sg in segs) or id(gbuf) in self._adopted
        if whole_r and whole_g and not clash and (r.n, r.h, r.w, r.c, r.sn, r.sh, r.sw) == (g.n, g.h, g.w, g.c, g.sn, g.sh, g.sw):
            self.grads[id(r)] = [gbuf, [(0, r.c)], r]
            self._adopted.add(id(gbuf))
            return
        self.grad_add(r, g, may_ride=True)

    def flat_axpy(self, src_ptr, dst_ptr, count, alpha=1.0, accumulate=True):
        """dst (+)= alpha * src over `count` contiguous f32 (gradient accumulation across backward passes; fill with
        `alpha` when src_ptr is None)."""
        cols = 1024
        rows = count // cols

        def v(ptr, w, c):
            return NULL_VIEW if ptr is None else View(ptr, w * c, w * c, c, c, 0)

        if rows:
            self.lib.axpby(F32, 1, 1, rows, v(src_ptr, rows, cols), v(dst_ptr, rows, cols), alpha, 1.0, 1 << 30,
                           1 if accumulate else 0, self.stream)
            self.launches += 1
        rem = count - rows * cols
        if rem:
            off = 4 * rows * cols
            self.lib.axpby(F32, 1, 1, 1, v(None if src_ptr is None else src_ptr + off, 1, rem), v(dst_ptr + off, 1, rem), alpha, 1.0,
                           1 << 30, 1 if accumulate else 0, self.stream)
            self.launches += 1

    def seed_grad(self, t):
        """Gradient buffer of an output tensor, to be written by a loss kernel."""
        gv, acc = self.grad_write(t)
        assert not acc
        return gv

    # ------------------------------------------------------------------ backward
    def backward(self):
        if self.wgrad_flush_frac:  # total weight-gradient work of this pass: the background-flush marks are fractions of it
            self._wg_total = sum(2.0 * a[0].ci * a[0].taps * a[0].co * a[1][0].n * a[1][0].h * a[1][0].w
                                 for fn, a, _ in self.tape if fn == self._bw_conv and self._needs_wgrad(a[0]))
            self._wg_total += sum(2.0 * st.ci * st.taps * st.co * a[2][0].n * a[2][0].h * a[2][0].w
                                  for fn, a, _ in self.tape if fn == self._bw_block3 for st in a[:2] if self._needs_wgrad(st))
            self._wg_total += sum(2.0 * st.ci * st.taps * st.co * a[1][0].n * a[1][0].h * a[1][0].w
                                  for fn, a, _ in self.tape if fn == self._bw_block4 for st in a[0] if self._needs_wgrad(st))
        main_t = torch.cuda.current_stream(self.device)
        if self._side_join_pending:  # side-stream work of the forward pass nobody has joined yet (the stem's im2col)
            self.join_side()
        self.blk3_pairs = self.conv_pairs = self.blk4_pairs = 0
        skip = 0
        tape = self.tape
        for i in range(len(tape) - 1, -1, -1):
            fn, args, _ = tape[i]
            if fn == self._bw_block3:
                self._blk4_flush()
                # (armed when the NEXT entry is a fused Block of the same image size without a residual: the layer's prior Block)
                nxt = tape[i - 1] if i > 0 else None
                # ... and reads none of this Block's differentiable inputs: else its bookkeeping (an accumulate target, a
                # copy-on-write of the shared gradient) could launch work that must see this Block's result first
                if (self.blk3_pair and self._blk3_hold is None and nxt is not None and nxt[0] == self._bw_block3
                        and nxt[1][5] is None and nxt[1][2][0].h == args[2][0].h and nxt[1][2][0].w == args[2][0].w
                        and not ({id(v.base) for v in args[2] if v.rg} & {id(v.base) for v in nxt[1][2] if v.rg})
                        and not any(v.base is nxt[1][4].base for v in args[2])):
                    self._blk3_arm = 1
                fn(*args)
                if self._blk3_arm == 1:  # (it did not reach its launch point)
                    self._blk3_arm = 0
            elif fn == self._bw_block4:
                self._blk3_flush()
                # (armed when the NEXT entry is a fused default Block of the same image size without a residual -- the layer's prior
                #  Block -- that reads none of this Block's differentiable inputs nor its output: the rule of the light Blocks above)
                nxt = tape[i - 1] if i > 0 else None
                if (self.blk4_pair and self._blk4_hold is None and nxt is not None and nxt[0] == self._bw_block4
                        and nxt[1][4] is None and nxt[1][1][0].h == args[1][0].h and nxt[1][1][0].w == args[1][0].w
                        and not ({id(v.base) for v in args[1] if v.rg} & {id(v.base) for v in nxt[1][1] if v.rg})
                        and not any(v.base is nxt[1][3].base for v in args[1])):
                    self._blk4_arm = 1
                fn(*args)
                if self._blk4_arm == 1:  # (it did not reach its launch point)
                    self._blk4_arm = 0
            else:
                self._blk3_flush()
                self._blk4_flush()
                if skip:
                    skip -= 1
                    continue
                if self.conv_pair and fn == self._bw_conv and self.dt != F32 and self._two_unfused_blocks(i):
                    # X.conv2 | Y.conv2 in one launch, then X.conv1, Y.conv1 (their own gradient outputs pair inside _bw_conv)
                    self._conv_arm = 1
                    fn(*args)
                    if self._conv_arm == 1:
                        self._conv_arm = 0
                    y2, x1, y1 = tape[i - 2], tape[i - 1], tape[i - 3]
                    y2[0](*y2[1])
                    self._conv_flush()
                    x1[0](*x1[1])
                    self._conv_flush()
                    y1[0](*y1[1])
                    self._conv_flush()
                    skip = 3
                    continue
                fn(*args)
                self._conv_flush()
        self._blk3_flush()
        self._blk4_flush()
        self._conv_flush()
        for bid in list(self._riders):
            gv, g, acc = self._riders.pop(bid)
            self.lib.axpby(self.dt, g.n, g.h, g.w, g.cv(), gv.cv(), 1.0, 1.0, 1 << 30, 1 if acc else 0, self.stream)
            self.launches += 1
        self._reduce_wgrads()
        for p, ptr in self._pgrad_tmp.values():  # NHWC-accumulated gradients of [1,C,h,w] parameters -> NCHW
            _, c, h, w = p.shape
            v = NT(ptr, 1, h, w, c, h * w * c, w * c, c, 4, rg=False)
            self.lib.nhwc_to_nchw(F32, 1, c, h, w, v.cv(), self.param_grad_ptr(p), self.stream)
            self.launches += 1
        self.tape.clear()

    def _bw_conv(self, site, segs, act, out, res1, res2):
        g = self.grad_read(out)
        if g is None:
            return
        for r in (res1, res2):
            if r is not None and r.rg:
                self._grad_residual(r, g, out, segs)
        x0 = segs[0]
        if self._needs_wgrad(site) and "wg" not in self._ablate:  # ("wg": timing-only ablation, no weight gradients at all)
            self._wgrad(site, segs, act, g)
        todo = [k for k, s in enumerate(segs) if s.rg and site.seg_rg[k]]
        if self.conv_pair and len(todo) >= 2 and self._conv_hold is None and self._conv_arm == 0 and self.dt != F32:
            self._conv_arm = 1  # the first of this conv's data gradients waits for the second (independent outputs, same input)
        for k in todo:
            self._dgrad_one(site, g, segs[k], k, act, x0)
        if len(todo) >= 2:
            self._conv_flush()

    def _dgrad_one(self, site, g, s, k, act, x0):
        gv, prev, acc = self._dgrad_target(s)
        self._dgrad_launch(site, g, s, k, act, x0, gv, prev, acc)

    def _dgrad_launch(self, site, g, s, k, act, x0, gv, prev, acc):
        """The data-gradient conv of input segment k into an already acquired target (see _dgrad_target)."""
        a = _lib.ConvArgs()
        gn, gh, gw, vw = self._geom(site.ks, [g, gv, s, prev])
        a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act, a.dact = self.dt, gn, gh, gw, site.ks, 1, ACT_NONE, act
        a.seg[0] = vw(g)
        a.weight = site.img_dg[k]
        a.bias = None
        a.out = vw(gv)
        a.aux = vw(s) if act != ACT_NONE else NULL_VIEW
        a.res1 = vw(prev) if acc else NULL_VIEW
        a.res2 = NULL_VIEW
        if self._ablate and ((site.name.endswith(".conv.3") and "d3" in self._ablate) or ("r12" in self._ablate and x0.h <= 12)
                             or any(x0.h == r and ("r%d" % r) in self._ablate for r in (24, 48, 96, 192))):
            return  # TIMING-ONLY ablation: the data gradient of every Block's second conv is not launched
        if self._conv_arm or self._conv_hold is not None:
            wr = {id(gv.base)}
            rd = {id(g.base), id(s.base)} | ({id(prev.base)} if acc else set())
            if self._conv_arm == 1 and self.prof is None:
                self._conv_hold = (a, self.launches, wr, rd, (site, x0, s.c))
                self._conv_arm = 2
                return
            if self._conv_hold is not None:
                ha, hl, hwr, hrd, _ = self._conv_hold
                if (hl == self.launches and not (hwr & (wr | rd)) and not (wr & hrd)
                        and self.lib.conv2d_pair_supported(C.byref(ha), C.byref(a))):
                    self._conv_hold, self._conv_arm = None, 0
                    self.lib.conv2d_pair(C.byref(ha), C.byref(a), self.stream)
                    self.launches += 1
                    self.conv_pairs += 1
                    return
                self._conv_flush()
        self._timed("conv_dgrad", site, x0, lambda: self.lib.conv2d(C.byref(a), self.stream), ci=s.c)

    def _conv_flush(self):
        """Launch a held data-gradient conv on its own (no partner came, or it cannot share a launch)."""
        self._conv_arm = 0
        if self._conv_hold is None:
            return
        ha, _, _, _, (site, x0, ci) = self._conv_hold
        self._conv_hold = None
        self._timed("conv_dgrad", site, x0, lambda: self.lib.conv2d(C.byref(ha), self.stream), ci=ci)

    def _two_unfused_blocks(self, i):
        """tape[i], [i-1] = conv2, conv1 of Block X and tape[i-2], [i-3] = conv2, conv1 of Block Y (backward order), both unfused,
        without residuals, on one image size, and Y's output is not an input of X: the two Blocks are independent in the backward
        pass (the posterior and the prior Block of a decoder layer), so conv2 of X may be followed by conv2 of Y."""
        if i < 3:
            return False
        e = [self.tape[i - j] for j in range(4)]
        if any(x[0] != self._bw_conv for x in e):
            return False
        (s0, g0, _, o0, r01, r02), (s1, g1, _, o1, r11, r12), (s2, g2, _, o2, r21, r22), (s3, g3, _, o3, r31, r32) = [x[1] for x in e]
        if any(r is not None for r in (r01, r02, r11, r12, r21, r22, r31, r32)):
            return False
        if len(g0) != 1 or len(g2) != 1 or g0[0].base is not o1.base or g2[0].base is not o3.base:
            return False
        if any(sg.base is o2.base for sg in g1) or (o0.h, o0.w) != (o2.h, o2.w):
            return False
        if {id(v.base) for v in g1 if v.rg} & {id(v.base) for v in g3 if v.rg}:  # both would accumulate into one gradient
            return False
        return s0.ks == s2.ks

    def _dgrad_target(self, s):
        """Where the data gradient of input tensor `s` goes: (view to write, view to add when accumulating, accumulate?)."""
        gv, acc = self.grad_write(s, defer_hazard=True)
        prev = gv
        if acc and self._frozen(s):
            # grad(s) lives in an adopted buffer a deferred wgrad will still read: accumulate OUT of place (same
            # traffic: the kernel reads `prev` as a residual either way)
            if s.base is s:
                e = self.grads[id(s)]
                e[0] = self._new_grad(s.n, s.h, s.w, s.c)
                gv = e[0].chan(0, s.c)
            else:
                self._cow(s)
                gv = prev = self.grads[id(s.base)][0].chan(s.coff, s.coff + s.c)
        return gv, prev, acc

    def _bw_pool(self, x, out, d):
        g = self.grad_read(out)
        if g is None:
            return
        gv, acc = self.grad_write(x)
        if isinstance(d, float):
            self.lib.adaptive_avgpool_bwd(self.dt, x.n, x.h, x.w, out.h, out.w, g.cv(), gv.cv(), 1 if acc else 0, self.stream)
        else:
            self.lib.avgpool_bwd(self.dt, x.n, out.h, out.w, d, g.cv(), gv.cv(), 1 if acc else 0, self.stream)
        self.launches += 1

    def _param_reduce(self, param, g):
        acc = id(param) in self.pgrad_init
        _, c, h, w = param.shape
        if h * w == 1 or c == 1:
            dst = self.param_grad_ptr(param)
        else:
            ent = self._pgrad_tmp.get(id(param))
            if ent is None:
                ent = (param, self.arena.alloc(c * h * w * 4))
                self._pgrad_tmp[id(param)] = ent
            dst = ent[1]
        self.lib.batch_reduce(self.dt, g.n, g.h, g.w, g.cv(), dst, 1 if acc else 0, 1.0 / self.loss_scale, self.stream)
        self.launches += 1
        self.pgrad_init.add(id(param))

    def _bw_upsample(self, x, out, bias_param):
        g = self.grad_read(out)
        if g is None:
            return
        if bias_param is not None and bias_param.requires_grad:
            self._param_reduce(bias_param, g)
        if x.rg:
            gv, acc = self.grad_write(x)
            self.lib.upsample_bwd(self.dt, x.n, x.h, x.w, out.h, out.w, g.cv(), gv.cv(), 1 if acc else 0, self.stream)
            self.launches += 1

    def _bw_bcast(self, param, out):
        g = self.grad_read(out)
        if g is None or not param.requires_grad:
            return
        self._param_reduce(param, g)

    def _bw_pad(self, x, out):
        g = self.grad_read(out)
        if g is None:
            return
        inner = NT(g.ptr, x.n, x.h, x.w, x.c, g.sn, g.sh, g.sw, g.es, rg=False)
        self.grad_add(x, inner)

    def _bw_reparam(self, q_loc, q_ls, p_loc, p_ls, z, logt, fb_col=None, coef_ptr=None):
        """`coef_ptr`: device address of d(loss)/d(sum kl) for this layer when it is not the engine-wide one (the abduction
        passes of DSCM.forward draw z from q but contribute no KL term: their coefficient is a device-side zero)."""
        gz = self.grad_read(z)
        job = self._riders.pop(id(p_loc.base), None)  # (taken before grad_write would land it as a launch of its own)
        gql, a1 = self.grad_write(q_loc)
        gqs, a2 = self.grad_write(q_ls)
        gpl, a3 = self.grad_write(p_loc)
        gps, a4 = self.grad_write(p_ls)
        assert a1 == a2 and a3 == a4
        views = [q_loc, q_ls, p_loc, p_ls, z, gql, gqs, gpl, gps] + ([gz] if gz is not None else [])
        if job is not None and not (z.c % 8 == 0 and all(self._v16(v) for v in views)
                                    and (job[1].n, job[1].h, job[1].w) == (z.n, z.h, z.w)):
            gv, g, acc = job  # not the 16-byte bf16 path after all: the copy is its own launch
            self.lib.axpby(self.dt, g.n, g.h, g.w, g.cv(), gv.cv(), 1.0, 1.0, 1 << 30, 1 if acc else 0, self.stream)
            self.launches += 1
            job = None
        args = (self.dt, z.n, z.h, z.w, z.c, q_loc.cv(), q_ls.cv(), p_loc.cv(), p_ls.cv(), z.cv(), logt,
                gz.cv() if gz is not None else NULL_VIEW, self.kl_coef_ptr if coef_ptr is None else coef_ptr, 0,
                None if fb_col is None else self.kl_chan_ptr + 4 * fb_col, gql.cv(), gqs.cv(), gpl.cv(),
                gps.cv(), 1 if a1 else 0, 1 if a3 else 0)
        if job is None:
            self.lib.reparam_kl_bwd(*args, self.stream)
        else:
            gv, g, acc = job
            self.lib.reparam_kl_bwd_rider(*args, g.cv(), gv.cv(), 1 if acc else 0, self.stream)
        self.launches += 1
