"""HVAE image mechanism on MI355X -- drop-in for the reference's ``src/vae.py`` surface.

Same constructor (``HVAE(args)``), same four methods (``forward / sample / abduct / forward_latents``, vae.py:439-522),
same module tree and therefore the same ``state_dict`` keys and the same default-init RNG consumption
(``encoder.blocks.0.conv.1.weight`` ... ``likelihood.x_loc.bias``), so checkpoints, ``model.apply(init_bias)``,
``copy.deepcopy`` (EMA) and ``setup_tensorboard``'s introspection keep working.  The nn.Conv2d objects are parameter
*holders*: no module in this file has a PyTorch forward that computes anything.  All arithmetic is issued by
``HVAE`` through ``engine.Engine`` as fused HIP launches (libcgen_hip.so); there is no ATen / CPU fallback.

Layout: activations live NHWC in an arena; API tensors are accepted as NCHW (either memory format) and returned
as NCHW-shaped tensors.  ``HVAE.compute_dtype`` selects "f32" (exact f32-MFMA path, parity) or "f16".
"""
import math
import os
from typing import Dict, List, Optional

import numpy as np
import weakref

import torch
from torch import Tensor, nn

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, NULL_VIEW
from .engine import ConvSite, Engine

EPS = -9  # minimum logscale (vae.py:11)


def _flat_f32(name, *ts):
    lib = _lib.load()
    _lib.require_gpu()
    out = []
    shape, dev = ts[0].shape, ts[0].device
    for t in ts:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.shape == shape):
            raise _lib.CgenError(f"{name}: expects same-shaped tensors on one GPU (there is no CPU path)")
        out.append(t.detach().to(torch.float32).contiguous())
    return lib, out


def gaussian_kl(q_loc: Tensor, q_logscale: Tensor, p_loc: Tensor, p_logscale: Tensor) -> Tensor:
    """Element-wise KL(q || p) map, import-compatible with vae.py:14-25 (no clamps, no reduction).  The model itself uses
    the fused cgen_reparam_kl kernels; this is the stand-alone op for callers of the module-level name.  HIP only."""
    lib, (ql, qs, pl, ps) = _flat_f32("gaussian_kl", q_loc, q_logscale, p_loc, p_logscale)
    out = torch.empty_like(ql)
    lib.gaussian_kl_map(ql.numel(), ql.data_ptr(), qs.data_ptr(), pl.data_ptr(), ps.data_ptr(), out.data_ptr(),
                        torch.cuda.current_stream(ql.device).cuda_stream)
    return out


_FREE_RNG = {}


def sample_gaussian(loc: Tensor, logscale: Tensor) -> Tensor:
    """loc + exp(logscale) * N(0, 1) (vae.py:28-30) with the device Philox generator, seeded from torch.initial_seed();
    every call advances the counter.  HIP only."""
    lib, (l, s) = _flat_f32("sample_gaussian", loc, logscale)
    dev = l.device
    rng = _FREE_RNG.get(dev)
    if rng is None:
        rng = _FREE_RNG[dev] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)
    out = torch.empty_like(l)
    n = l.numel()
    st = torch.cuda.current_stream(dev).cuda_stream

    def view(t):
        return _lib.View(t.data_ptr(), n, n, 1, 1, 0)

    lib.sample_gaussian(_lib.F32, 1, 1, n, 1, view(l), view(s), NULL_VIEW, rng.data_ptr(), 0x5A17, 0.0, view(out), st)
    lib.rng_advance(rng.data_ptr(), (n + 3) // 4 + 1, st)
    return out


_SA_ENGINES = weakref.WeakKeyDictionary()  # standalone engines of holder modules used outside an HVAE (inference only)


def _stem_site(name, conv, index, dtype=None):
    """The 7x7 stem: a real 7x7 site when the direct kernel serves it (Engine.stem), else im2col + 1x1 over 49 * Cin channels."""
    lib = _lib.load()
    direct = (os.environ.get("CGEN_STEM_DIRECT", "1") != "0" and conv.kernel_size[0] == 7
              and lib.stem_conv_supported(_lib.F16 if dtype == "f16" else _lib.F32, conv.in_channels, 7, conv.out_channels))
    if direct:
        return ConvSite(name, conv, [conv.in_channels], [False], index)
    return ConvSite(name, conv, [conv.in_channels * conv.kernel_size[0] ** 2], [False], index, as_1x1=True)


def _block_sites(sites, name, blk, seg_c, seg_rg):
    cs = blk.convs()
    first = len(sites)
    sites.append(ConvSite(f"{name}.conv.1", cs[0], seg_c, seg_rg, len(sites)))
    for j, c in enumerate(cs[1:]):
        sites.append(ConvSite(f"{name}.conv.{3 + 2 * j}", c, [c.in_channels], [True], len(sites)))
    if blk.light and len(cs) == 2 and cs[0].kernel_size[0] == 3 and cs[1].kernel_size[0] == 3:
        # the two 3x3 convs of a light Block run as ONE launch (Engine.block2 -> cgen_block3): fragment-ordered weight images
        sites[first].blk3, sites[first + 1].blk3 = ("a", sites[first + 1]), ("b", sites[first])
    if not blk.light and len(cs) == 4 and cs[1].kernel_size[0] == 3 and cs[2].kernel_size[0] == 3:
        # the four convs of a default Block run as ONE launch (Engine.block4 -> cgen_block4): fragment-ordered weight images
        four = sites[first:first + 4]
        for r in range(4):
            sites[first + r].blk4 = (r, four)
    if hasattr(blk, "width_proj"):
        sites.append(ConvSite(f"{name}.width_proj", blk.width_proj, seg_c, seg_rg, len(sites)))


def _decoder_sites(sites, dec, prefix, zd, ctx):
    """Every conv of the decoder with its input segmentation (the virtual torch.cat's of vae.py:176,188,294,300)."""
    for i, b in enumerate(dec.blocks):
        w = b.in_width
        n = f"{prefix}blocks.{i}"
        if b.cond_prior:
            _block_sites(sites, n + ".prior", b.prior, [w, ctx], [True, False])
        else:
            _block_sites(sites, n + ".prior", b.prior, [w], [True])
        if b.stochastic:
            _block_sites(sites, n + ".posterior", b.posterior, [w, ctx, w], [True, False, True])
        sites.append(ConvSite(n + ".z_proj", b.z_proj, [zd, ctx], [True, False], len(sites)))
        if not b.q_correction:
            sites.append(ConvSite(n + ".z_feat_proj", b.z_feat_proj, [zd, w], [True, True], len(sites)))
        _block_sites(sites, n + ".conv", b.conv, [w], [True])


def run_block(eng, blk, segs):
    """Block.forward (vae.py:73-84) as fused launches."""
    site = lambda conv: eng.site_by_id[id(conv)]
    act = ACT_RELU if blk.light else ACT_GELU
    cs = blk.convs()
    res = None
    if blk.residual:
        x = segs[0]
        if x.c != cs[-1].out_channels:
            res = eng.conv(site(blk.width_proj), segs, ACT_NONE)
        else:
            res = x
    if blk.light and len(cs) == 2:  # the two 3x3 convs of a light Block: one fused launch where the kernel serves the shape
        h = eng.block2(site(cs[0]), site(cs[1]), segs, act, res1=res, trunk=blk.residual)
    else:
        h = None
        if not blk.light and len(cs) == 4 and site(cs[0]).blk4 is not None:  # a default Block: one fused launch where the kernel serves it
            h = eng.block4([site(c) for c in cs], segs, res1=res, trunk=blk.residual)
        if h is None:
            # (a residual Block's last conv writes the next value of the trunk: h = x + f(x), vae.py:78)
            h = eng.conv(site(cs[0]), segs, act, res1=res if len(cs) == 1 else None, trunk=blk.residual and len(cs) == 1)
            for j, c in enumerate(cs[1:]):
                last = j == len(cs) - 2
                h = eng.conv(site(c), [h], act, res1=res if last else None, trunk=blk.residual and last)
    if blk.d:
        h = eng.pool(h, blk.d)  # int: avg_pool2d; float: adaptive_avg_pool2d (vae.py:79-83)
    return h


def run_encoder(eng, enc, x):
    """Encoder.forward (vae.py:112-134): {resolution: activation}."""
    stem = eng.site_by_id[id(enc.stem)]
    h = eng.stem(stem, x)
    acts = {}
    for blk in enc.blocks:
        h = run_block(eng, blk, [h])
        if h.h % 2 and h.h > 1:
            h = eng.pad_br(h)
        acts[h.w] = h
    return acts


def _standalone_engine(mod, make_sites):
    """Engine of a holder module used on its own (``Block(...)(x)``, ``Encoder(args)(x)``, ``DGaussNet(args).predict(h)``):
    the reference's classes are importable and callable by themselves.  Inference only -- training runs through
    ``HVAE``, whose engine owns the tape.  ``mod.compute_dtype`` ("f32" default, or "f16") picks the kernels."""
    dev = next(mod.parameters()).device
    dt = getattr(mod, "compute_dtype", "f32")
    eng = _SA_ENGINES.get(mod)
    if eng is None or eng.device != dev or eng.dtype_name != dt:
        if dev.type != "cuda":
            raise _lib.CgenError(f"{type(mod).__name__} runs on an MI355X only: move it to the GPU; there is no CPU fallback")
        eng = Engine(dev, dt)
        eng.bind(mod, make_sites())
        _SA_ENGINES[mod] = eng
    else:
        eng.check_params()
    eng.begin()
    eng.recording = False
    eng.prepare_weights()
    return eng


def _no_standalone_grad(mod, *tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise RuntimeError(f"{type(mod).__name__} on its own is inference-only here: gradients flow through HVAE.forward "
                           "(one tape for the whole model); call it under torch.no_grad() or detach the input")


class Block(nn.Module):
    """Parameter holder mirroring vae.py:33-84 (two variants, optional width_proj)."""

    def __init__(self, in_width, bottleneck, out_width, kernel_size=3, residual=True, down_rate=None, version=None):
        super().__init__()
        self.d = down_rate
        self.residual = residual
        self.light = version == "light"
        padding = 0 if kernel_size == 1 else 1
        if self.light:
            activation = nn.ReLU()
            self.conv = nn.Sequential(activation, nn.Conv2d(in_width, bottleneck, kernel_size, 1, padding),
                                      activation, nn.Conv2d(bottleneck, out_width, kernel_size, 1, padding))
        else:
            activation = nn.GELU()
            self.conv = nn.Sequential(activation, nn.Conv2d(in_width, bottleneck, 1, 1),
                                      activation, nn.Conv2d(bottleneck, bottleneck, kernel_size, 1, padding),
                                      activation, nn.Conv2d(bottleneck, bottleneck, kernel_size, 1, padding),
                                      activation, nn.Conv2d(bottleneck, out_width, 1, 1))
        if self.residual and (self.d or in_width > out_width):
            self.width_proj = nn.Conv2d(in_width, out_width, 1, 1)

    def convs(self):
        return [m for m in self.conv if isinstance(m, nn.Conv2d)]

    @torch.no_grad()
    def _forward_standalone(self, x):
        def sites():
            out = []
            _block_sites(out, "block", self, [self.convs()[0].in_channels], [True])
            return out

        eng = _standalone_engine(self, sites)
        return eng.to_nchw(run_block(eng, self, [eng.from_nchw(x.to(eng.device, torch.float32))]))

    def forward(self, x):
        """vae.py:73-84 on the HIP engine (standalone use: inference only; inside an HVAE the model's engine runs the Block)."""
        _no_standalone_grad(self, x)
        return self._forward_standalone(x)


class Encoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        stages = []
        for i, stage in enumerate(args.enc_arch.split(",")):
            start = stage.index("b") + 1
            end = stage.index("d") if "d" in stage else None
            n_blocks = int(stage[start:end])
            if i == 0:
                self.stem = nn.Conv2d(args.input_channels, args.widths[0], kernel_size=7, stride=1, padding=3)
            stages += [(args.widths[i], None) for _ in range(n_blocks)]
            if "d" in stage:
                stages += [(args.widths[i + 1], int(stage[stage.index("d") + 1]))]
        blocks = []
        for i, (width, d) in enumerate(stages):
            prev_width = stages[max(0, i - 1)][0]
            blocks.append(Block(prev_width, int(prev_width / args.bottleneck), width, down_rate=d, version=args.vr))
        for b in blocks:
            b.conv[-1].weight.data *= np.sqrt(1 / len(blocks))
        self.blocks = nn.ModuleList(blocks)

    @torch.no_grad()
    def _forward_standalone(self, x):
        def sites():
            out = [_stem_site("stem", self.stem, 0)]
            for i, b in enumerate(self.blocks):
                _block_sites(out, f"blocks.{i}", b, [b.convs()[0].in_channels], [True])
            return out

        eng = _standalone_engine(self, sites)
        acts = run_encoder(eng, self, eng.from_nchw(x.to(eng.device, torch.float32)))
        return {r: eng.to_nchw(a) for r, a in acts.items()}

    def forward(self, x):
        """vae.py:112-134 on the HIP engine: {resolution: activation [B,C,r,r]} (standalone use: inference only)."""
        _no_standalone_grad(self, x)
        return self._forward_standalone(x)


class DecoderBlock(nn.Module):
    def __init__(self, args, in_width, out_width, resolution):
        super().__init__()
        bottleneck = int(in_width / args.bottleneck)
        self.res = resolution
        self.stochastic = self.res <= args.z_max_res
        self.z_dim = args.z_dim
        self.cond_prior = args.cond_prior
        self.q_correction = args.q_correction
        self.in_width, self.out_width = in_width, out_width
        k = 3 if self.res > 2 else 1
        self.prior = Block(in_width + args.context_dim if self.cond_prior else in_width, bottleneck,
                           2 * self.z_dim + in_width, kernel_size=k, residual=False, version=args.vr)
        if self.stochastic:
            self.posterior = Block(2 * in_width + args.context_dim, bottleneck, 2 * self.z_dim, kernel_size=k,
                                   residual=False, version=args.vr)
        self.z_proj = nn.Conv2d(self.z_dim + args.context_dim, in_width, 1)
        if not self.q_correction:
            self.z_feat_proj = nn.Conv2d(self.z_dim + in_width, out_width, 1)
        self.conv = Block(in_width, bottleneck, out_width, kernel_size=k, version=args.vr)

    # ---- standalone use (inference only): the two halves of a decoder layer on a private engine
    def _sa_engine(self):
        ctx = self.z_proj.in_channels - self.z_dim
        w = self.in_width

        def sites():
            out = []
            _block_sites(out, "prior", self.prior, [w, ctx] if self.cond_prior else [w], [True, False] if self.cond_prior else [True])
            if self.stochastic:
                _block_sites(out, "posterior", self.posterior, [w, ctx, w], [True, False, True])
            return out

        return _standalone_engine(self, sites)

    @staticmethod
    def _logt(ls, t):
        return ls if t is None else ls + float(torch.as_tensor(t, dtype=torch.float32).log())

    def forward_prior(self, z, pa=None, t=None):
        """vae.py:170-182: (p_loc, p_logscale [+ log t], p_features) from the prior Block on ``z`` (``cat[z, pa]`` when the prior
        is conditional -- a virtual concat: two input segments of the first conv)."""
        _no_standalone_grad(self, z, pa)
        with torch.no_grad():
            return self._prior_sa(z, pa, t)

    def _prior_sa(self, z, pa, t):
        eng = self._sa_engine()
        zt = eng.from_nchw(z.to(eng.device, torch.float32))
        segs = [zt]
        if self.cond_prior:
            segs.append(eng.from_parents(pa, zt.h, zt.w).crop(zt.h))
        out = eng.to_nchw(run_block(eng, self.prior, segs))
        zd = self.z_dim
        return out[:, :zd], self._logt(out[:, zd:2 * zd], t), out[:, 2 * zd:]

    def forward_posterior(self, z, x, pa, t=None):
        """vae.py:184-192: (q_loc, q_logscale [+ log t]) from the posterior Block on the virtual ``cat[z, pa, x]``."""
        if not self.stochastic:
            raise AttributeError("this DecoderBlock has no posterior (res > z_max_res)")
        _no_standalone_grad(self, z, x, pa)
        with torch.no_grad():
            return self._posterior_sa(z, x, pa, t)

    def _posterior_sa(self, z, x, pa, t):
        eng = self._sa_engine()
        zt = eng.from_nchw(z.to(eng.device, torch.float32))
        xt = eng.from_nchw(x.to(eng.device, torch.float32))
        out = eng.to_nchw(run_block(eng, self.posterior, [zt, eng.from_parents(pa, zt.h, zt.w).crop(zt.h), xt]))
        q_loc, q_ls = out.chunk(2, dim=1)
        return q_loc, self._logt(q_ls, t)

    def forward(self, *a, **k):
        raise RuntimeError("DecoderBlock has no forward() in the reference either (vae.py:137-192): call forward_prior / "
                           "forward_posterior, or run the layer through Decoder.forward / HVAE")


class Decoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        stages = []
        for i, stage in enumerate(args.dec_arch.split(",")):
            res = int(stage.split("b")[0])
            n_blocks = int(stage[stage.index("b") + 1:])
            stages += [(res, args.widths[::-1][i]) for _ in range(n_blocks)]
        blocks = []
        for i, (res, width) in enumerate(stages):
            next_width = stages[min(len(stages) - 1, i + 1)][1]
            blocks.append(DecoderBlock(args, width, next_width, res))
        scale = np.sqrt(1 / len(blocks))
        for b in blocks:  # vae.py:303-308
            b.z_proj.weight.data *= scale
            b.conv.conv[-1].weight.data *= scale
            b.prior.conv[-1].weight.data *= 0.0
        self.blocks = nn.ModuleList(blocks)
        self.all_res = list(np.unique([stages[i][0] for i in range(len(stages))]))
        bias = []
        for i, res in enumerate(self.all_res):
            if res <= args.bias_max_res:
                bias.append(nn.Parameter(torch.zeros(1, args.widths[::-1][i], res, res)))
        self.bias = nn.ParameterList(bias)
        self.cond_prior = args.cond_prior
        self.is_drop_cond = True if "morphomnist" in args.hps else False

    @torch.no_grad()
    def drop_cond(self):
        """One categorical per step for the whole batch (vae.py:310-319).  Drawn on the host CPU generator so that
        data-parallel ranks seeded alike share the draw."""
        opt = int(torch.distributions.Categorical(1 / 3 * torch.ones(3)).sample())
        return {0: (0, 1), 1: (1, 0), 2: (1, 1)}[opt]

    # ---- standalone use (inference only): the HVAE's own decoder pass (HVAE._decode) on a private engine
    def _host(self):
        h = self.__dict__.get("_sa_host")
        if h is None:
            h = self.__dict__["_sa_host"] = _DecoderHost(self)
        return h

    @torch.no_grad()
    def _forward_standalone(self, parents, x=None, t=None, abduct=False, latents=()):
        host = self._host()
        zd = self.blocks[0].z_dim
        ctx = self.blocks[0].z_proj.in_channels - zd

        def sites():
            out = []
            _decoder_sites(out, self, "", zd, ctx)
            return out

        eng = _standalone_engine(self, sites)
        eng.rng_advance(1)
        R = max(b.res for b in self.blocks)
        pa = eng.from_parents(parents, R, R)
        acts = None if x is None else {int(r): eng.from_nchw(v.to(eng.device, torch.float32)) for r, v in x.items()}
        lat = [None if z is None else eng.from_nchw(z.to(eng.device, torch.float32)) for z in latents]
        drop = self.drop_cond() if (self.training and self.cond_prior) else (1, 1)
        collect = "qp" if acts is not None else ("p" if (abduct and self.cond_prior) else None)
        h, out = HVAE._decode(host, eng, pa, acts=acts, t=t, latents=lat, collect=collect, drop=drop)
        stats = []
        if acts is not None:
            for z, ql, qs, pl, ps in out:
                ql, qs, pl, ps = (eng.to_nchw(v) for v in (ql, qs, pl, ps))
                st = dict(kl=gaussian_kl(ql, qs, pl, ps))
                if abduct:
                    zt = eng.to_nchw(z)
                    st["z"] = {"z": zt, "q_loc": ql, "q_logscale": qs} if self.cond_prior else zt
                stats.append(st)
        elif collect == "p":
            stats = [dict(z={"p_loc": eng.to_nchw(pl), "p_logscale": eng.to_nchw(ps)}) for pl, ps in out]
        return eng.to_nchw(h), stats

    def forward(self, parents, x=None, t=None, abduct=False, latents=()):
        """vae.py:222-301 on the HIP engine: (h, stats).  ``x`` is the encoder's {res: activation} dict (posterior pass: every
        stochastic block contributes ``dict(kl=...)``, plus ``z`` when ``abduct``), or None (prior sampling / replay of
        ``latents``).  Standalone use is inference only; inside an HVAE the model's engine runs the decoder."""
        _no_standalone_grad(self, parents, *(x.values() if x is not None else ()))
        return self._forward_standalone(parents, x, t, abduct, latents)


class _DecoderHost:
    """What HVAE._decode needs from its model when a Decoder runs on its own."""

    def __init__(self, dec):
        self.decoder = dec
        self.__dict__["noise"] = None

    def _site(self, eng, conv):
        return eng.site_by_id[id(conv)]

    def _run_block(self, eng, blk, segs):
        return run_block(eng, blk, segs)

    def _next_eps(self, eng, shape_nhwc):
        src = self.decoder.__dict__.get("noise")  # optional list of NCHW eps tensors, consumed in draw order (parity tests)
        if not src:
            return None  # Philox inside the kernels
        e = src.pop(0)
        n, h, w, c = shape_nhwc
        assert tuple(e.shape) == (n, c, h, w), (tuple(e.shape), shape_nhwc)
        return eng.from_nchw(e.to(eng.device, torch.float32))

    def _scratch_kl(self, eng, B, res, zd):
        return eng.new_f32(B * _lib.load().reparam_kl_chunks(res, res, zd))


class DGaussNet(nn.Module):
    """Discretised-Gaussian likelihood head (vae.py:322-422): parameter holder + kernel front-end."""
    kind = "dgauss"

    def __init__(self, args):
        super().__init__()
        self.x_loc = nn.Conv2d(args.widths[0], args.input_channels, kernel_size=1, stride=1)
        self.x_logscale = nn.Conv2d(args.widths[0], args.input_channels, kernel_size=1, stride=1)
        self.channels = args.input_channels
        if args.input_channels == 3:
            self.channel_coeffs = nn.Conv2d(args.widths[0], 3, kernel_size=1, stride=1)
        if args.std_init > 0:
            nn.init.zeros_(self.x_logscale.weight)
            nn.init.constant_(self.x_logscale.bias, np.log(args.std_init))
            covariance = args.x_like.split("_")[0]
            if covariance == "fixed":
                self.x_logscale.weight.requires_grad = False
                self.x_logscale.bias.requires_grad = False
            elif covariance == "shared":
                self.x_logscale.weight.requires_grad = False
                self.x_logscale.bias.requires_grad = True

    def heads(self):
        hs = [self.x_loc, self.x_logscale]
        if self.channels == 3:
            hs.append(self.channel_coeffs)
        return hs

    def out_channels(self):
        return 2 * self.channels + (3 if self.channels == 3 else 0)

    # ---- standalone use (inference only; inside an HVAE the model's engine drives these kernels)
    def _standalone_params(self, h):
        def sites():
            return [ConvSite(nm, cv, [cv.in_channels], [True], i)
                    for i, (nm, cv) in enumerate(zip(("x_loc", "x_logscale", "channel_coeffs"), self.heads()))]

        eng = _standalone_engine(self, sites)
        ht = eng.from_nchw(h.to(eng.device, torch.float32))
        buf = eng.new(ht.n, ht.h, ht.w, self.out_channels())
        o = 0
        for cv in self.heads():
            eng.conv(eng.site_by_id[id(cv)], [ht], ACT_NONE, out=buf.chan(o, o + cv.out_channels))
            o += cv.out_channels
        return eng, buf

    @torch.no_grad()
    def _forward_standalone(self, h, x=None, t=None):
        eng, buf = self._standalone_params(h)
        C, B, R, W = self.channels, buf.n, buf.h, buf.w
        loc = torch.empty((B, C, R, W), dtype=torch.float32, device=eng.device)
        logscale = torch.empty_like(loc)
        xv = eng.from_nchw(x.to(eng.device, torch.float32)).cv() if (x is not None and C == 3) else NULL_VIEW
        logt = 0.0 if t is None else float(torch.tensor(t).log())
        # one launch (cgen_dgauss_params): EPS clamp, + log t, and for RGB the autoregressive means of vae.py:357-383
        eng.lib.dgauss_params(eng.dt, B, R, W, C, buf.cv(), xv, logt, loc.data_ptr(), logscale.data_ptr(), eng.stream)
        return loc, logscale

    def forward(self, h, x=None, t=None):
        """vae.py:352-386: (loc, logscale) of the pixel distribution given the decoder's last hidden state."""
        _no_standalone_grad(self, h, x)
        return self._forward_standalone(h, x, t)

    @torch.no_grad()
    def nll(self, h, x):
        """vae.py:393-411: per-sample negative log-likelihood in nats / dim (`cgen_dgauss_nll_fwd`)."""
        eng, buf = self._standalone_params(h)
        lib = eng.lib
        B, R, W = buf.n, buf.h, buf.w
        xt = eng.from_nchw(x.to(eng.device, torch.float32))
        nchunk = lib.like_chunks(R, W)
        part = torch.empty((B, nchunk), dtype=torch.float32, device=eng.device)
        lib.dgauss_nll_fwd(eng.dt, B, R, W, self.channels, buf.cv(), xt.cv(), part.data_ptr(), eng.stream)
        return part.sum(1) / float(self.channels * R * W)

    @torch.no_grad()
    def sample(self, h, return_loc=True, t=None):
        """vae.py:413-422 (`cgen_dgauss_sample`): (x clamped to [-1, 1], scale)."""
        if not return_loc and t is not None and self.channels == 3:
            raise TypeError("'float' object is not subscriptable")  # vae.py:418 hands t over as `x`: the reference raises here
        eng, buf = self._standalone_params(h)
        B, R, W, C = buf.n, buf.h, buf.w, self.channels
        xo = torch.empty((B, C, R, W), dtype=torch.float32, device=eng.device)
        so = torch.empty_like(xo)
        if not return_loc:
            eng.rng_advance(1)
        eng.lib.dgauss_sample(eng.dt, B, R, W, C, buf.cv(), 0.0, None if return_loc else eng.rng_ptr(), 978, xo.data_ptr(),
                              so.data_ptr(), eng.stream)
        return xo, so


class _HVAEFunction(torch.autograd.Function):
    """Bridges ``out['elbo'].backward()`` to the engine's tape.  Parameter gradients are written by the HIP
    kernels straight into the engine's flat gradient buffer and attached as ``p.grad`` views."""

    @staticmethod
    def forward(ctx, trigger, model, x, parents, beta):
        ctx.model = model
        out3 = model._run_forward(x, parents, beta, record=True)
        ctx.beta = float(beta)
        return out3[0], out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_elbo, g_nll, g_kl):
        ctx.model._run_backward(g_elbo, g_nll, g_kl, ctx.beta)
        return None, None, None, None, None


class _DSCMFunction(torch.autograd.Function):
    """``DSCM.forward``'s image half as one differentiable node: outputs (elbo, nll, kl, mean cf_x, var cf_x); the
    backward pass seeds the factual likelihood / KL gradients and d loss / d cf_x and sweeps the shared tape once."""

    @staticmethod
    def forward(ctx, trigger, model, x, parents, cf_parents_list, beta, t_abduct):
        ctx.model = model
        ctx.beta = float(beta)
        out3, cf_mean, var = model._run_dscm_forward(x, parents, cf_parents_list, beta, t_abduct)
        if var is None:
            var = torch.empty(0, device=cf_mean.device)
        ctx.mark_non_differentiable(var)
        return out3[0], out3[1], out3[2], cf_mean, var

    @staticmethod
    def backward(ctx, g_elbo, g_nll, g_kl, g_cf, g_var):
        ctx.model._run_dscm_backward(g_elbo, g_nll, g_kl, g_cf, ctx.beta)
        return None, None, None, None, None, None, None


class HVAE(nn.Module):
    compute_dtype = "f32"

    def __init__(self, args):
        super().__init__()
        args.vr = "light" if "ukbb" in args.hps else None  # vae.py:428
        self.encoder = Encoder(args)
        self.decoder = Decoder(args)
        if args.x_like.split("_")[1] == "dgauss":
            self.likelihood = DGaussNet(args)
        else:
            raise NotImplementedError(f"{args.x_like} not implemented.")
        self.cond_prior = args.cond_prior
        self.free_bits = args.kl_free_bits
        self.light = args.vr == "light"
        self.z_dim, self.context_dim = args.z_dim, args.context_dim
        self.input_channels = args.input_channels
        self.q_correction = args.q_correction
        self._reset_runtime()

    _RUNTIME_KEYS = ("_eng", "_trigger", "noise", "_saved", "_coef", "_fb_buf", "_grad_prev", "_beta_dev", "_saved_gen", "_saved_cf",
                     "_coef_keep")

    def _reset_runtime(self):
        for k in self._RUNTIME_KEYS:
            self.__dict__[k] = None  # noise: optional list of NCHW eps tensors consumed in draw order (parity tests)

    # engine state is per-instance and never copied (copy.deepcopy(model) for the EMA builds its own lazily)
    def __deepcopy__(self, memo):
        import copy as _copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._RUNTIME_KEYS:
                new.__dict__[k] = _copy.deepcopy(v, memo)
        new._reset_runtime()
        return new

    # ------------------------------------------------------------------ engine plumbing
    def _act(self):
        return ACT_RELU if self.light else ACT_GELU

    def engine(self) -> Engine:
        dev = next(self.parameters()).device
        eng = self.__dict__["_eng"]
        if eng is None or eng.device != dev or eng.dtype_name != self.compute_dtype:
            if dev.type != "cuda":
                raise _lib.CgenError("HVAE runs on an MI355X only: move the model to the GPU (model.to('cuda')); "
                                     "there is no CPU fallback")
            eng = Engine(dev, self.compute_dtype)
            eng.bind(self, self._make_sites())
            self.__dict__["_eng"] = eng
        else:
            eng.check_params()
        return eng

    def _make_sites(self):
        """Enumerate every conv with its input segmentation (the virtual torch.cat's of vae.py:176,188,294,300)."""
        sites = []

        def add(name, conv, seg_c, seg_rg, as_1x1=False):
            sites.append(ConvSite(name, conv, seg_c, seg_rg, len(sites), as_1x1=as_1x1))

        def add_block(name, blk, seg_c, seg_rg):
            _block_sites(sites, name, blk, seg_c, seg_rg)

        C = self.input_channels
        sites.append(_stem_site("encoder.stem", self.encoder.stem, len(sites), self.compute_dtype))
        for i, b in enumerate(self.encoder.blocks):
            add_block(f"encoder.blocks.{i}", b, [b.convs()[0].in_channels], [True])
        _decoder_sites(sites, self.decoder, "decoder.", self.z_dim, self.context_dim)
        lk = self.likelihood
        if lk.kind == "dgauss":
            for nme, cv in zip(("x_loc", "x_logscale", "channel_coeffs"), lk.heads()):
                add("likelihood." + nme, cv, [cv.in_channels], [True])
        else:
            add("likelihood.conv", lk.conv, [lk.conv.in_channels], [True])
        return sites

    def _site(self, eng, conv):
        return eng.site_by_id[id(conv)]

    # ------------------------------------------------------------------ graph pieces
    def _run_block(self, eng, blk, segs):
        return run_block(eng, blk, segs)

    def _encode(self, eng, x):
        return run_encoder(eng, self.encoder, x)

    def _next_eps(self, eng, shape_nhwc):
        src = self.__dict__["noise"]
        if src is None:
            return None
        e = src.pop(0)
        n, h, w, c = shape_nhwc
        assert tuple(e.shape) == (n, c, h, w), (tuple(e.shape), shape_nhwc)
        return eng.from_nchw(e.to(eng.device, torch.float32))

    def _decode(self, eng, parents, acts=None, t=None, latents=None, collect=None, drop=(1, 1), kl=None, fb=None):
        """Decoder.forward (vae.py:222-301).  `collect`: None | "z" | "q" (q stats for cond-prior abduction) |
        "p" (prior stats).  `kl` = (ptr, stride, offsets) when the KL is wanted."""
        dec = self.decoder
        B = parents.n
        bias = {int(p.shape[2]): p for p in dec.bias}
        logt = 0.0 if t is None else float(torch.tensor(t).log())
        h = z = eng.bcast(bias[1], B)
        pa_sto_full = parents
        if dec.is_drop_cond and drop[0] != 1:
            pa_sto_full = eng.scale_channels(parents, 2, drop[0])
        out = []
        latents = [] if latents is None else latents
        # Philox stream ids: unique per (decoder pass within this engine step, stochastic layer)
        eng.passes = getattr(eng, "passes", 0) + 1
        sid = 1000 * eng.passes
        # Software pipeline over layers (training / abduction pass): z_feat_proj of layer k, the upsampling of its output and
        # the prior Block of layer k + 1 form a chain that only needs z_k and p_feat_k -- it runs on the side stream while the
        # main stream does z_proj, the conv Block and the posterior Block of layer k + 1.  Same kernels, same tape order.
        # (inference passes too -- the abduction pass of the counterfactual loop: +1.5 % cf/s, +3 % on the plain trunk; CGEN_INFER_BRANCH=0 off)
        rec2 = eng.recording or eng.infer_branch
        pipeline = acts is not None and rec2 and eng.fwd_branch and eng.prof is None
        if pipeline:
            for p_ in dec.bias:  # (lazily built NHWC images: build them before any side-stream section needs one)
                eng.param_nhwc(p_)
        side_ahead = False
        for i, blk in enumerate(dec.blocks):
            res = blk.res
            pa = parents.crop(res)
            pa_sto = pa_sto_full.crop(res)
            if h.h < res:
                bp = bias.get(res)
                same = z is h
                h = eng.upsample(h, res, bp)
                if not blk.q_correction:
                    if same:
                        z = h
                    elif side_ahead:
                        z_lo = z
                        z = eng.on_side(lambda: eng.upsample(z_lo, res, bp), bw=1)
                    else:
                        z = eng.upsample(z, res, bp)
            p_in = h if blk.q_correction else z
            run_prior = lambda: self._run_block(eng, blk.prior, [p_in, pa_sto] if blk.cond_prior else [p_in])
            # the prior and the posterior Block of a layer are independent: two streams (one fork / join per layer)
            want_two = blk.stochastic and acts is not None and rec2
            # (a fresh fork: the posterior Block -- the main chain -- is enqueued first, the prior Block behind the mark: Engine.fork_mark)
            mark = eng.fork_mark() if want_two and not side_ahead else None
            t_q0 = len(eng.tape)
            qout = self._run_block(eng, blk.posterior, [h, pa, acts[res]]) if mark is not None else None
            t_q1 = len(eng.tape)
            two = want_two and (side_ahead or eng.fork_side(after=mark))
            if side_ahead and not two:
                eng.join_side()
                side_ahead = False
            pout = eng.on_side(run_prior, bw=1) if two else run_prior()
            if t_q1 > t_q0 and eng.recording:
                # tape order [prior][posterior] whichever was launched first: backward() then enqueues the posterior Block's
                # backward (main strand) BEFORE the prior Block's (side strand) behind the fork -- the main chain must be the first
                # edge out of a fork for a captured graph to keep it on its queue (Engine.fork_mark)
                eng.tape[t_q0:] = eng.tape[t_q1:] + eng.tape[t_q0:t_q1]
            zd = blk.z_dim
            p_loc, p_ls, p_feat = pout.chan(0, zd), pout.chan(zd, 2 * zd), pout.chan(2 * zd, pout.c)
            if blk.stochastic:
                sid += 1
                if acts is not None:
                    if qout is None:
                        qout = self._run_block(eng, blk.posterior, [h, pa, acts[res]])
                    if two:
                        eng.join_side()
                        side_ahead = False
                    q_loc, q_ls = qout.chan(0, zd), qout.chan(zd, 2 * zd)
                    eps = self._next_eps(eng, q_loc.shape)
                    kptr = kl[0] + 4 * kl[2][i] if kl is not None else self._scratch_kl(eng, B, res, zd)
                    kstride = kl[1] if kl is not None else _lib.load().reparam_kl_chunks(res, res, zd)
                    fbl = None if fb is None else (fb[0], fb[1], fb[2][i])
                    z = eng.reparam_kl(q_loc, q_ls, p_loc, p_ls, eps, sid, logt, kptr, kstride, fb=fbl)
                    if collect == "z":
                        out.append(z)
                    elif collect == "q":
                        out.append((z, q_loc, q_ls))
                    elif collect == "qp":  # (standalone Decoder.forward: everything a stats dict needs)
                        out.append((z, q_loc, q_ls, p_loc, p_ls))
                else:
                    zi = latents[i] if i < len(latents) else None
                    if zi is not None:
                        z = zi
                    else:
                        eps = self._next_eps(eng, p_loc.shape)
                        z = eng.sample_gaussian(p_loc, p_ls, eps, sid, logt)
                        if i >= len(latents) and collect == "p":
                            out.append((p_loc, p_ls))
            else:
                z = p_loc
            z_cur = z
            feat = not blk.q_correction and i + 1 < len(dec.blocks)
            hold = []
            # the next layer's prior chain (z_feat_proj -> prior Block) on the side stream, forked HERE but enqueued behind z_proj:
            # the main chain must be the first edge out of the fork for a captured graph to keep it on one queue (Engine.fork_mark)
            ahead = feat and pipeline and two and dec.blocks[i + 1].stochastic and not dec.blocks[i + 1].q_correction
            mark = eng.fork_mark() if ahead and eng.fwd_mainfirst else None
            if ahead and mark is None and eng.fork_side():
                z = eng.on_side(lambda: eng.conv(self._site(eng, blk.z_feat_proj), [z_cur, p_feat], ACT_NONE, tape_hold=hold))
                side_ahead = True
                feat = False
            h = eng.conv(self._site(eng, blk.z_proj), [z_cur, pa], ACT_NONE, res1=h, res2=p_feat, trunk=True)
            t_zp = len(eng.tape)
            if mark is not None and eng.fork_side(after=mark):
                z = eng.on_side(lambda: eng.conv(self._site(eng, blk.z_feat_proj), [z_cur, p_feat], ACT_NONE, tape_hold=hold))
                side_ahead = True
                feat = False
            h = self._run_block(eng, blk.conv, [h])
            # z_feat_proj's backward goes between the conv Block's and z_proj's (tape: right behind z_proj): it is the op that joins the
            # side strand of backward() (it reads the gradient the prior Block's backward wrote), so it sits late; and it stays in
            # FRONT of z_proj's backward, whose residual copy into grad(p_feat) then still rides on the reparam backward (a rider
            # parked before another writer of that buffer would have to land as a launch of its own: +5.6 us per layer, measured)
            eng.tape[t_zp:t_zp] = hold
            if feat:
                z = eng.conv(self._site(eng, blk.z_feat_proj), [z_cur, p_feat], ACT_NONE)
        if side_ahead:
            eng.join_side()
        return h, out

    def _scratch_kl(self, eng, B, res, zd):
        return eng.new_f32(B * _lib.load().reparam_kl_chunks(res, res, zd))

    def _kl_layout(self, eng):
        lib = _lib.load()
        offs, tot = {}, 0
        for i, blk in enumerate(self.decoder.blocks):
            if blk.stochastic:
                offs[i] = tot
                tot += lib.reparam_kl_chunks(blk.res, blk.res, blk.z_dim)
        return offs, tot

    def _likelihood_params(self, eng, h):
        """The 1x1 heads write side by side into one buffer: [loc | logscale | coeffs] (or the 100 DMoL logits)."""
        lk = self.likelihood
        if lk.kind == "dgauss":
            buf = eng.new(h.n, h.h, h.w, lk.out_channels())
            o = 0
            for cv in lk.heads():
                eng.conv(self._site(eng, cv), [h], ACT_NONE, out=buf.chan(o, o + cv.out_channels))
                o += cv.out_channels
            return buf
        return eng.conv(self._site(eng, lk.conv), [h], ACT_NONE)

    def _pa_nt(self, eng, parents):
        """Parents as an engine tensor: a stride-0 broadcast of [B,1,1,ctx] when the caller did not materialise the
        spatial expansion (engine.from_parents), the general [B,R,R,ctx] layout otherwise."""
        assert parents.shape[1] == self.context_dim
        R = max(b.res for b in self.decoder.blocks)  # (only used for [B,ctx] / [B,ctx,1,1] input; crop() sizes each use)
        return eng.from_parents(parents, R, R)

    def _prep_inputs(self, eng, x, parents):
        assert x.dim() == 4 and parents.dim() in (2, 4) and parents.shape[1] == self.context_dim
        xin = None
        if x is not None:
            x = x.to(eng.device)
            # raw u8 pixels: trainer.py:17's (x - 127.5) / 127.5 is fused into the layout kernel
            xin = eng.from_nchw(x, rg=False, sub=127.5, mul=1.0 / 127.5) if x.dtype == torch.uint8 else eng.from_nchw(x, rg=False)
        pa = self._pa_nt(eng, parents)
        return xin, pa

    # ------------------------------------------------------------------ training forward / backward
    def _run_forward(self, x, parents, beta, record):
        eng = self.engine()
        eng.begin()
        eng.recording = record
        eng.prepare_weights(force=record)  # a training step always re-images: fused AdamW bypasses torch's version counter
        if self.__dict__["noise"] is None:
            eng.rng_advance(1)
        xin, pa = self._prep_inputs(eng, x, parents)
        out3 = self._elbo_pass(eng, xin, pa, beta)
        eng.recording = False
        return out3

    def _elbo_pass(self, eng, xin, pa, beta):
        """HVAE.forward (vae.py:439-458) on prepared inputs; records the tape when eng.recording."""
        drop = (1, 1)
        if self.training and self.cond_prior:
            drop = self.decoder.drop_cond()
        B, R, Cx = xin.n, xin.h, xin.c
        lib = eng.lib
        offs, kl_total = self._kl_layout(eng)
        kl_ptr = eng.new_f32(B * max(kl_total, 1))
        acts = self._encode(eng, xin)
        fb = None
        if self.free_bits > 0:
            # vae.py:443-449: per-layer, per-channel batch means of the KL, floored at free_bits.  Under data parallelism the
            # batch is the GLOBAL batch: the per-channel sums are all-reduced before the floor (below).
            cols, ncol = {}, 0
            for i, blk in enumerate(self.decoder.blocks):
                if blk.stochastic:
                    cols[i] = ncol
                    ncol += blk.z_dim
            s_buf = torch.empty(B * ncol + ncol, dtype=torch.float32, device=eng.device)  # S[B][ncol] then chan_mask[ncol]
            fb = (s_buf.data_ptr(), ncol, cols)
        h, _ = self._decode(eng, pa, acts=acts, drop=drop, kl=(kl_ptr, kl_total, offs), fb=fb)
        if fb is not None and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # one small exchange: every row of S is replaced by the global per-channel mean, so that the finalize kernel
            # floors (and masks the gradient of) the same batch statistic on every rank; the per-sample gradient weight
            # stays 1/B_local because the gradient all-reduce averages over ranks (equal per-rank batches).  Not
            # capturable: the trainer runs this configuration without a hipGraph.
            world = torch.distributed.get_world_size()
            S = s_buf[:B * ncol].view(B, ncol)
            col = S.sum(0)
            torch.distributed.all_reduce(col)
            S.copy_((col / float(B * world)).expand(B, ncol))
        params = self._likelihood_params(eng, h)
        nchunk = lib.like_chunks(R, R)
        nll_ptr = eng.new_f32(B * nchunk)
        lk = self.likelihood
        if lk.kind == "dgauss":
            lib.dgauss_nll_fwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), nll_ptr, eng.stream)
        else:
            lib.dmol_nll_fwd(eng.dt, B, R, R, params.cv(), xin.cv(), nll_ptr, eng.stream)
        out3 = torch.empty(3, dtype=torch.float32, device=eng.device)
        dims = float(Cx * R * R)
        if fb is None:
            lib.elbo_finalize(B, nll_ptr, nchunk, dims, kl_ptr, kl_total, dims, float(beta), self.__dict__.get("_beta_dev"),
                              out3.data_ptr(), eng.stream)
        else:
            mask_ptr = fb[0] + 4 * B * fb[1]
            lib.elbo_finalize_fb(B, nll_ptr, nchunk, dims, fb[0], fb[1], dims, float(self.free_bits), float(beta),
                                 self.__dict__.get("_beta_dev"), out3.data_ptr(), mask_ptr, eng.stream)
            eng.kl_chan_ptr = mask_ptr
            self.__dict__["_fb_buf"] = s_buf  # keep alive until the backward pass
        eng.launches += 2
        self.__dict__["_saved"] = (params, xin, B, R, Cx, dims)
        self.__dict__["_saved_gen"] = eng.generation
        return out3

    # ------------------------------------------------------------------ DSCM.forward with a differentiable counterfactual branch
    _dscm_differentiable = True

    def _run_dscm_forward(self, x, parents, cf_parents_list, beta, t_abduct):
        """dscm.py:40-72 as ONE recorded engine step: the factual ELBO pass, then per particle an abduction pass (encoder +
        posterior decoder pass drawing z ~ q at temperature t_abduct; it contributes no KL term) whose final hidden state is
        also the reconstruction replay's (same latents, same parents, same kernels -- HVAE.abduct_with_reconstruction) and a
        replay of those latents under the counterfactual parents.  All passes share the arena and the tape, so one reverse
        sweep delivers d loss / d theta through cf_x as train_cf.py:159-183 needs.
        Returns (out3 = [elbo, nll, kl], mean cf_x [B,C,R,R], var cf_x or None)."""
        from .dscm import cf_pixels

        if self.likelihood.kind != "dgauss" or getattr(self.likelihood, "logit_space", False):
            raise NotImplementedError(
                "DSCM.forward under autograd (dscm.py:40-72 as train_cf.py:159-183 uses it) is built for the DGaussNet head: with DmolNet "
                "the counterfactual pixels are the soft mixture mean of dmol.py:218-245, whose gradient w.r.t. the 100 logits per pixel "
                "has no HIP kernel (cgen_dmol_decode is forward only).  The reference never fine-tunes a DMoL HVAE (config 3 is the "
                "survey's construct); call DSCM.forward under torch.no_grad() with this head")
        eng = self.engine()
        eng.begin()
        eng.recording = True
        eng.prepare_weights(force=True)
        if self.__dict__["noise"] is None:
            eng.rng_advance(1)
        xin, pa = self._prep_inputs(eng, x, parents)
        out3 = self._elbo_pass(eng, xin, pa, beta)
        if getattr(eng, "_zero4", None) is None:
            eng._zero4 = torch.zeros(4, dtype=torch.float32, device=eng.device)
        P = len(cf_parents_list)
        x_nchw = eng.to_nchw(xin)
        sx = torch.zeros_like(x_nchw) if P > 1 else None
        sx2 = torch.zeros_like(x_nchw) if P > 1 else None
        passes, cf_x = [], None
        for cfp_t in cf_parents_list:
            cfp = self._pa_nt(eng, cfp_t)
            eng.kl_coef_override = eng._zero4.data_ptr()
            acts = self._encode(eng, xin)
            h_ab, qs = self._decode(eng, pa, acts=acts, t=t_abduct, collect="q" if self.cond_prior else "z")
            eng.kl_coef_override = None
            zs = [q[0] for q in qs] if self.cond_prior else qs
            rec_params = self._likelihood_params(eng, h_ab)
            h_cf, _ = self._decode(eng, cfp, latents=zs)
            cf_params = self._likelihood_params(eng, h_cf)
            (rl, rs), (cl, cs) = self._decode_params(eng, rec_params), self._decode_params(eng, cf_params)
            cf_x = cf_pixels(x_nchw, rl, rs, cl, cs, sx, sx2)
            passes.append((rec_params, cf_params))
        eng.recording = False
        self.__dict__["_saved_cf"] = (passes, xin)
        if P > 1:
            return out3, sx / P, (sx2 - sx ** 2 / P) / P
        return out3, cf_x, None

    def _decode_params(self, eng, params):
        """DGaussNet.sample(h) with return_loc=True on ready-made head outputs -> (loc, scale) as NCHW f32."""
        B, R, Cx = params.n, params.h, self.input_channels
        xo = torch.empty((B, Cx, R, R), dtype=torch.float32, device=eng.device)
        so = torch.empty_like(xo)
        eng.lib.dgauss_sample(eng.dt, B, R, R, Cx, params.cv(), 0.0, None, 978, xo.data_ptr(), so.data_ptr(), eng.stream)
        eng.launches += 1
        return xo, so

    def _run_dscm_backward(self, g_elbo, g_nll, g_kl, g_cf, beta):
        eng = self.__dict__["_eng"]
        passes, xin = self.__dict__["_saved_cf"]
        if g_cf is not None:
            g = g_cf.to(eng.device, torch.float32).contiguous()
            eng.stream = torch.cuda.current_stream(eng.device).cuda_stream
            B, R, Cx = xin.n, xin.h, xin.c
            S = eng.set_loss_scale(B * float(Cx * R * R))  # (the same value _run_backward sets for this pass)
            for rec_params, cf_params in passes:
                g_rec, g_cfp = eng.seed_grad(rec_params), eng.seed_grad(cf_params)
                eng.lib.cf_dgauss_bwd(eng.dt, B, R, R, Cx, rec_params.cv(), cf_params.cv(), xin.cv(), g.data_ptr(), S / len(passes),
                                      g_rec.cv(), g_cfp.cv(), eng.stream)
                eng.launches += 1
            self.__dict__["_coef_keep"] = g
        self._run_backward(g_elbo, g_nll, g_kl, beta)

    def _run_backward(self, g_elbo, g_nll, g_kl, beta):
        eng = self.__dict__["_eng"]
        if eng is None or eng.generation != self.__dict__.get("_saved_gen"):
            # a later pass (abduct / forward_latents / sample / another forward) recycled the arena and the tape this
            # result was recorded on: its activations are gone.  Fail loudly instead of returning no gradients.
            raise RuntimeError("HVAE backward: the engine state of this forward pass has been recycled by a later HVAE call; "
                               "call backward() before running another pass on the same model (DSCM.forward records its "
                               "counterfactual passes on the same tape for exactly this reason)")
        params, xin, B, R, Cx, dims = self.__dict__["_saved"]
        lib = eng.lib
        z = torch.zeros((), device=eng.device)
        ge = g_elbo if g_elbo is not None else z
        gn = g_nll if g_nll is not None else z
        gk = g_kl if g_kl is not None else z
        # d/d(sum_b nll_b) and d/d(sum_b kl_b): scalar glue on 0-dim tensors
        S = eng.set_loss_scale(B * dims)  # (1 for f32; the reduces that produce parameter gradients divide it out again)
        coef = torch.stack([(ge + gn) * (S / (B * dims)), (ge * beta + gk) * (S / (B * dims))]).float().contiguous()
        self.__dict__["_coef"] = coef
        eng.stream = torch.cuda.current_stream(eng.device).cuda_stream
        eng.kl_coef_ptr = coef.data_ptr() + 4
        gparams = eng.seed_grad(params)
        if getattr(self.likelihood, "logit_space", False):  # simple_vae's GaussNet: same dequantisation noise as the forward pass
            u, snap = self.__dict__["_gauss_noise"][:2]
            lib.gauss_nll_bwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), u, snap.data_ptr(), 977, coef.data_ptr(), 0, gparams.cv(),
                              eng.stream)
        elif self.likelihood.kind == "dgauss":
            lib.dgauss_nll_bwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), coef.data_ptr(), 0, gparams.cv(), eng.stream)
        else:
            lib.dmol_nll_bwd(eng.dt, B, R, R, params.cv(), xin.cv(), coef.data_ptr(), 0, gparams.cv(), eng.stream)
        eng.launches += 1
        # torch semantics: a backward pass ACCUMULATES into existing .grad (trainer.py:64-67, accu_steps > 1).  The engine
        # overwrites its flat gradient buffer, so when gradients are already attached they are parked and added back.
        live = [p for p in self.parameters() if p.grad is not None]
        for p in live:
            if p.grad.data_ptr() != eng.param_grad_view(p).data_ptr():
                raise RuntimeError("HVAE parameters' .grad must stay the engine's views (use zero_grad() to reset them)")
        prev = None
        if live:
            prev = self.__dict__.get("_grad_prev")
            if prev is None or prev.numel() != eng.flat_g.numel():
                prev = self.__dict__["_grad_prev"] = torch.empty_like(eng.flat_g)
            eng.flat_axpy(eng.flat_g.data_ptr(), prev.data_ptr(), prev.numel(), accumulate=False)
        eng.backward()
        if prev is not None:
            had = {id(p) for p in live}
            for p in self.parameters():  # .grad is None means zero: nothing to add back for those
                if id(p) not in had and id(p) in eng.pgrad_init:
                    o = eng.p_off[id(p)]
                    eng.flat_axpy(None, prev.data_ptr() + 4 * o, p.numel(), alpha=0.0, accumulate=False)
            eng.flat_axpy(prev.data_ptr(), eng.flat_g.data_ptr(), prev.numel())
        for p in self.parameters():
            if id(p) in eng.pgrad_init and p.grad is None:
                p.grad = eng.param_grad_view(p)

    def forward(self, x: Tensor, parents: Tensor, beta: int = 1) -> Dict[str, Tensor]:
        """vae.py:439-458 -> {elbo, nll, kl} as 0-dim tensors in nats/dim; ``elbo`` is differentiable."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            trig = self.__dict__["_trigger"]
            dev = next(self.parameters()).device
            if trig is None or trig.device != dev:
                trig = torch.zeros(1, device=dev, requires_grad=True)
                self.__dict__["_trigger"] = trig
            elbo, nll, kl = _HVAEFunction.apply(trig, self, x, parents, beta)
        else:
            out3 = self._run_forward(x, parents, beta, record=False)
            elbo, nll, kl = out3[0], out3[1], out3[2]
        return dict(elbo=elbo, nll=nll, kl=kl)

    # ------------------------------------------------------------------ inference API
    def _sample_likelihood(self, eng, h, return_loc=True, t=None):
        params = self._likelihood_params(eng, h)
        lk = self.likelihood
        B, R, Cx = h.n, h.h, self.input_channels
        xo = torch.empty((B, Cx, R, R), dtype=torch.float32, device=eng.device)
        so = torch.empty_like(xo)
        if lk.kind == "dgauss":
            if not return_loc and t is not None and Cx == 3:
                # vae.py:418 passes t in the position of x: the reference indexes a float here and raises
                raise TypeError("'float' object is not subscriptable")
            # vae.py:416-420: neither branch applies the temperature (return_loc=True calls forward(h); return_loc=False
            # hands t over as `x`, which the grayscale path never reads)
            eng.lib.dgauss_sample(eng.dt, B, R, R, Cx, params.cv(), 0.0, None if return_loc else eng.rng_ptr(), 978,
                                  xo.data_ptr(), so.data_ptr(), eng.stream)
        else:
            if not return_loc:
                mode = 2
            elif "top" in lk.mask:  # dmol.py:178-180: "top3" -> int(mask[-1]), must be < 10
                mode = 10 + int(lk.mask[-1])
                assert 1 <= mode - 10 < 10, "invalid top_k"
            else:
                mode = {"soft": 0, "hard": 1}[lk.mask]
            logt = 0.0 if t is None else float(torch.tensor(t).log())
            eng.lib.dmol_decode(eng.dt, B, R, R, params.cv(), mode, eng.rng_ptr(), 977, logt, xo.data_ptr(), so.data_ptr(),
                                eng.stream)
        eng.launches += 1
        return xo, so

    def _begin_inference(self):
        eng = self.engine()
        eng.begin()
        eng.recording = False
        eng.prepare_weights()
        if self.__dict__["noise"] is None:
            eng.rng_advance(1)
        return eng

    @torch.no_grad()
    def sample(self, parents: Tensor, return_loc: bool = True, t: Optional[float] = None):
        """vae.py:460-464."""
        eng = self._begin_inference()
        pa = self._pa_nt(eng, parents)
        h, _ = self._decode(eng, pa, t=t)
        return self._sample_likelihood(eng, h, return_loc, t)

    @torch.no_grad()
    def abduct(self, x: Tensor, parents: Tensor, cf_parents: Optional[Tensor] = None, alpha: float = 0.5,
               t: Optional[float] = None):
        """vae.py:466-514.  Returns the exogenous z list, the list of {z,q_loc,q_logscale} dicts (cond. prior), or the
        mediator z* list when ``cf_parents`` is given."""
        return self._abduct(x, parents, cf_parents, alpha, t, False)

    @torch.no_grad()
    def abduct_with_reconstruction(self, x: Tensor, parents: Tensor, t: Optional[float] = None):
        """``abduct(x, parents, t=t)`` plus ``forward_latents(that, parents)`` without running the decoder twice: the
        posterior pass that produces the latents already holds the hidden state the replay would rebuild (same latents, same
        parents, same kernels), so the reconstruction is the likelihood head on it.  Returns (abduct's result, (loc, scale))
        -- bit-identical to the two calls (tests/test_gpu_model.py)."""
        out = self._abduct(x, parents, None, 0.5, t, True)
        if self.__dict__["noise"] is None:
            self.engine().rng_advance(1)  # the Philox state moves on as if forward_latents had been a call of its own
        return out

    def _abduct(self, x, parents, cf_parents, alpha, t, with_rec):
        eng = self._begin_inference()
        xin, pa = self._prep_inputs(eng, x, parents)
        acts = self._encode(eng, xin)
        logt = 0.0 if t is None else float(torch.tensor(t).log())
        if not self.cond_prior:
            h, zs = self._decode(eng, pa, acts=acts, t=t, collect="z")
            zs = [eng.to_torch_cl(z) for z in zs]
            return (zs, self._sample_likelihood(eng, h, True, None)) if with_rec else zs
        h, qs = self._decode(eng, pa, acts=acts, t=t, collect="q")
        if cf_parents is None:
            out = []
            for z, ql, qs_ in qs:
                d = dict(z=eng.to_torch_cl(z), q_loc=eng.to_torch_cl(ql), q_logscale=eng.to_torch_cl(qs_))
                if logt != 0.0:
                    d["q_logscale"] = d["q_logscale"] + logt  # the reference stores q_logscale + log t
                out.append(d)
            return (out, self._sample_likelihood(eng, h, True, None)) if with_rec else out
        assert not with_rec
        cfp = self._pa_nt(eng, cf_parents)
        _, ps = self._decode(eng, cfp, t=t, collect="p")
        assert len(ps) == len(qs)
        outs = []
        for (z, ql, qs_), (pl, pls) in zip(qs, ps):
            o = eng.new(z.n, z.h, z.w, z.c, rg=False)
            eng.lib.mediator_mix(eng.dt, z.n, z.h, z.w, z.c, z.cv(), ql.cv(), qs_.cv(), pl.cv(), pls.cv(), float(alpha),
                                 float(t) if t is not None else -1.0, logt, 0, o.cv(), eng.stream)
            eng.launches += 1
            outs.append(eng.to_torch_cl(o))
        return outs

    def _latents_in(self, eng, latents):
        ins = []
        for z in latents:
            if z is None:
                ins.append(None)
                continue
            if isinstance(z, dict):
                z = z["z"]
            z = z.to(eng.device)
            if z.dtype == eng.tdtype and z.permute(0, 2, 3, 1).is_contiguous():
                nt = eng.wrap_nhwc(z.permute(0, 2, 3, 1))
                ins.append(nt)
            else:
                ins.append(eng.from_nchw(z.float()))
        return ins

    @torch.no_grad()
    def forward_latents_pair(self, latents: List[Tensor], parents_a: Tensor, parents_b: Tensor, t: Optional[float] = None):
        """Two replays of the SAME latents under two parent settings (dscm.py:53-54: reconstruction and counterfactual), as
        two concurrent streams: each replay is a ~400-kernel latency chain that fills a fraction of the chip, and the two
        share nothing but read-only inputs.  Returns ((loc_a, scale_a), (loc_b, scale_b)) == (forward_latents(l, a),
        forward_latents(l, b))."""
        eng = self._begin_inference()
        pa_a = self._pa_nt(eng, parents_a)
        pa_b = self._pa_nt(eng, parents_b)
        lat = self._latents_in(eng, latents)

        def replay(pa):
            h, _ = self._decode(eng, pa, latents=lat, t=t)
            return self._sample_likelihood(eng, h, True, t)

        for p in self.decoder.bias:  # shared, lazily built, rebuilt every pass: build them BEFORE the streams fork
            eng.param_nhwc(p)
        philox = self.__dict__["noise"] is None
        if not eng.fork_side():
            ra = replay(pa_a)
            if philox:
                eng.rng_advance(1)  # what the second forward_latents call's _begin_inference would have done
            eng.passes = 0          # ... and its Philox stream ids start over
            return ra, replay(pa_b)
        nxt = eng.rng_next_ptr() if philox else None
        ra = eng.on_side(lambda: replay(pa_a))
        eng._rng_override = nxt
        eng.passes = 0  # the second replay numbers its Philox streams as a pass of its own
        try:
            rb = replay(pa_b)
        finally:
            eng._rng_override = None
        eng.join_side()
        if philox:
            eng.rng_advance(1)
        return ra, rb

    @torch.no_grad()
    def forward_latents(self, latents: List[Tensor], parents: Tensor, t: Optional[float] = None):
        """vae.py:516-522: replay (possibly partial) latents under `parents`; returns (loc in [-1,1], scale)."""
        eng = self._begin_inference()
        pa = self._pa_nt(eng, parents)
        h, _ = self._decode(eng, pa, latents=self._latents_in(eng, latents), t=t)
        return self._sample_likelihood(eng, h, True, t)
