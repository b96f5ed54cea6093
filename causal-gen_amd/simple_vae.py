"""Config 1 of the reference on MI355X -- drop-in for ``src/simple_vae.py`` (``VAE``: strided-conv encoder, linear
bottleneck with an optional conditional prior, upsample+conv decoder; 234 690 parameters at the morphomnist preset).

Same constructor, same module tree (``encoder.conv.0 ... decoder.prior.fc.2 ... likelihood.x_logscale``: identical
``state_dict`` keys and default-init RNG consumption) and the same four methods as the reference class.  Like ``vae.HVAE``
the modules are parameter holders; all arithmetic is issued through ``engine.Engine`` as HIP launches and there is no
CPU path.  The strided convolutions (5x5/s2/p1, 3x3/s2/p1) run as ``cgen_im2col_strided`` + a 1x1 conv, the Linear
layers as 1x1 convs on [B,1,1,F] tensors, LeakyReLU / clamp(min) as ``cgen_unary`` ops.  f32 only (plumbing / parity
config, SURVEY 8d: no roofline), discretised-Gaussian likelihood with one input channel.
"""
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, NULL_VIEW, UNARY_ADD, UNARY_CLAMP_MIN, UNARY_LEAKY_RELU
from .engine import ConvSite
from .vae import DGaussNet as _HvaeDGaussNet, HVAE, _HVAEFunction  # noqa: F401  (the autograd bridge is shared)

EPS = -9
EPS_z = -9
LEAK = 0.01  # nn.LeakyReLU() default slope


def _holder(name):
    def forward(self, *a, **k):
        raise RuntimeError(f"{name} is a parameter holder; run it through simple_vae.VAE (HIP engine)")
    return forward


class Encoder(nn.Module):
    """simple_vae.py:34-57."""

    def __init__(self, args):
        super().__init__()
        n = args.hidden_dim // 4
        act = nn.LeakyReLU()
        self.conv = nn.Sequential(nn.Conv2d(args.input_channels, n, kernel_size=5, stride=2, padding=1), act,
                                  nn.Conv2d(n, n, kernel_size=3, stride=2, padding=1), act,
                                  nn.Conv2d(n, n, kernel_size=3, stride=2, padding=1), act)
        self.fc = nn.Sequential(nn.Linear(n * 4 * 4, args.hidden_dim), act)
        self.embed = nn.Sequential(nn.Linear(args.hidden_dim + args.context_dim, args.hidden_dim), act)
        self.z_loc = nn.Linear(args.hidden_dim, args.z_dim)
        self.z_logscale = nn.Linear(args.hidden_dim, args.z_dim)

    forward = _holder("Encoder")


class CondPrior(nn.Module):
    """simple_vae.py:73-89."""

    def __init__(self, args):
        super().__init__()
        act = nn.LeakyReLU()
        self.fc = nn.Sequential(nn.Linear(args.context_dim, args.hidden_dim), act, nn.Linear(args.hidden_dim, args.hidden_dim), act)
        self.z_loc = nn.Linear(args.hidden_dim, args.z_dim)
        self.z_logscale = nn.Linear(args.hidden_dim, args.z_dim)
        self.p_feat = nn.Linear(args.hidden_dim, args.z_dim)
        nn.init.zeros_(self.z_loc.weight)
        nn.init.zeros_(self.z_loc.bias)
        nn.init.zeros_(self.z_logscale.weight)
        nn.init.zeros_(self.z_logscale.bias)

    forward = _holder("CondPrior")


class Decoder(nn.Module):
    """simple_vae.py:250-280, 313-321."""

    def __init__(self, args):
        super().__init__()
        self.cond_prior = args.cond_prior
        in_width = args.z_dim + args.context_dim
        if self.cond_prior:
            self.prior = CondPrior(args)
            in_width += args.z_dim
        else:
            self.register_buffer("p_loc", torch.zeros(1, args.z_dim))
            self.register_buffer("p_scale", torch.ones(1, args.z_dim))
        n = args.hidden_dim // 4
        act = nn.ReLU()
        self.fc = nn.Sequential(nn.Linear(in_width, args.hidden_dim), act, nn.Linear(args.hidden_dim, n * 4 * 4), act)
        self.conv = nn.Sequential(nn.Upsample(scale_factor=2, mode="nearest"), nn.Conv2d(n, n, kernel_size=3, stride=1, padding=1), act,
                                  nn.Upsample(scale_factor=2, mode="nearest"), nn.Conv2d(n, n, kernel_size=3, stride=1, padding=1), act,
                                  nn.Upsample(scale_factor=2, mode="nearest"), nn.Conv2d(n, 16, kernel_size=5, stride=1, padding=2), act)

    forward = _holder("Decoder")

    def drop_cond(self):
        opt = torch.distributions.Categorical(1 / 3 * torch.ones(3)).sample()
        return {0: (0, 1), 1: (1, 0), 2: (1, 1)}[int(opt)]


class _LinearSite:
    """Lets a Linear stand where the engine expects an nn.Conv2d: [out, in] is the same memory as [out, in, 1, 1]."""

    def __init__(self, lin):
        self.weight, self.bias = lin.weight, lin.bias
        self.kernel_size, self.out_channels, self.in_channels = (1, 1), lin.out_features, lin.in_features


class DGaussNet(_HvaeDGaussNet):
    """simple_vae.py:103-171: as vae.py's head but RGB channels are independent (no ``channel_coeffs``): the likelihood
    kernels see a [loc(C) | logscale(C)] view."""

    def __init__(self, args):
        nn.Module.__init__(self)
        self.x_loc = nn.Conv2d(args.widths[0], args.input_channels, kernel_size=1, stride=1)
        self.x_logscale = nn.Conv2d(args.widths[0], args.input_channels, kernel_size=1, stride=1)
        self.channels = args.input_channels
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
        return [self.x_loc, self.x_logscale]

    def out_channels(self):
        return 2 * self.channels


class GaussNet(DGaussNet):
    """simple_vae.py:173-248: Gaussian in logit space on dequantised pixels (x_like = *_gauss); same two heads."""
    logit_space = True


class VAE(HVAE):
    _dscm_differentiable = False  # (the counterfactual fine-tuning of train_cf.py is run with the HVAE)
    compute_dtype = "f32"

    def __init__(self, args):
        nn.Module.__init__(self)
        args.hidden_dim = 128  # simple_vae.py:327
        self.cond_prior = args.cond_prior
        self.encoder = Encoder(args)
        self.decoder = Decoder(args)
        x_dist = args.x_like.split("_")[1]
        if x_dist == "dgauss" and args.input_channels in (1, 3):
            self.likelihood = DGaussNet(args)
        elif x_dist == "gauss" and args.input_channels in (1, 3):
            self.likelihood = GaussNet(args)
        elif x_dist == "dmol" and args.input_channels == 3:  # simple_vae.py:336-339
            from .dmol import DmolNet

            self.likelihood = DmolNet(args)
        else:
            raise NotImplementedError(f"simple_vae on the HIP path: x_like=*_dgauss / *_gauss (one or three channels) or *_dmol (three) "
                                      f"(got {args.x_like}, {args.input_channels} channels)")
        self.free_bits = 0.0
        self.z_dim, self.context_dim, self.input_channels = args.z_dim, args.context_dim, args.input_channels
        self.hidden_dim = args.hidden_dim
        self._reset_runtime()

    # ------------------------------------------------------------------ sites
    def _make_sites(self):
        sites, self.__dict__["_lin"] = [], {}

        def add(name, mod, seg_c, seg_rg, as_1x1=False):
            if isinstance(mod, nn.Linear):
                ad = self.__dict__["_lin"][id(mod)] = _LinearSite(mod)
                sites.append(ConvSite(name, ad, seg_c, seg_rg, len(sites)))
                self.__dict__["_lin"][id(mod)] = sites[-1]
            else:
                sites.append(ConvSite(name, mod, seg_c, seg_rg, len(sites), as_1x1=as_1x1))
                self.__dict__["_lin"][id(mod)] = sites[-1]

        e, d, hd, zd, ctx = self.encoder, self.decoder, self.hidden_dim, self.z_dim, self.context_dim
        n = hd // 4
        add("encoder.conv.0", e.conv[0], [self.input_channels * 25], [False], as_1x1=True)
        add("encoder.conv.2", e.conv[2], [n * 9], [True], as_1x1=True)
        add("encoder.conv.4", e.conv[4], [n * 9], [True], as_1x1=True)
        add("encoder.fc.0", e.fc[0], [n * 16], [True])
        add("encoder.embed.0", e.embed[0], [hd, ctx], [True, False])
        add("encoder.z_loc", e.z_loc, [hd], [True])
        add("encoder.z_logscale", e.z_logscale, [hd], [True])
        if self.cond_prior:
            p = d.prior
            add("decoder.prior.fc.0", p.fc[0], [ctx], [False])
            add("decoder.prior.fc.2", p.fc[2], [hd], [True])
            add("decoder.prior.z_loc", p.z_loc, [hd], [True])
            add("decoder.prior.z_logscale", p.z_logscale, [hd], [True])
            add("decoder.prior.p_feat", p.p_feat, [hd], [True])
            add("decoder.fc.0", d.fc[0], [zd, zd, ctx], [True, True, False])
        else:
            add("decoder.fc.0", d.fc[0], [zd, ctx], [True, False])
        add("decoder.fc.2", d.fc[2], [hd], [True])
        add("decoder.conv.1", d.conv[1], [n], [True])
        add("decoder.conv.4", d.conv[4], [n], [True])
        add("decoder.conv.7", d.conv[7], [n * 25], [True], as_1x1=True)  # 5x5/p2 as im2col + 1x1 as well
        if self.likelihood.kind == "dgauss":
            for nme, cv in zip(("x_loc", "x_logscale"), self.likelihood.heads()):
                add("likelihood." + nme, cv, [cv.in_channels], [True])
        else:
            add("likelihood.conv", self.likelihood.conv, [self.likelihood.conv.in_channels], [True])
        return sites

    def _s(self, mod):
        return self.__dict__["_lin"][id(mod)]

    # ------------------------------------------------------------------ graph pieces
    def _vec(self, eng, y):
        """parents [B,ctx] or [B,ctx,R,R] (takes [:, :, 0, 0]; simple_vae.py:64-65) -> [B,1,1,ctx], no gradient."""
        y = y.to(eng.device, torch.float32)
        if y.dim() > 2:
            y = y[:, :, 0, 0]
        return eng.from_nchw(y.contiguous()[:, :, None, None], rg=False)

    def _lrelu(self, eng, x):
        return eng.unary(x, UNARY_LEAKY_RELU, LEAK)

    def _logscale(self, eng, x, t=None):
        x = eng.unary(x, UNARY_CLAMP_MIN, float(EPS_z))
        return x if t is None else eng.unary(x, UNARY_ADD, float(np.log(t)))

    def _encode(self, eng, x, y):
        """Encoder.forward, simple_vae.py:59-70 (t is never passed by the reference's callers)."""
        e = self.encoder
        h = x
        for i, (ks, st, pd) in zip((0, 2, 4), ((5, 2, 1), (3, 2, 1), (3, 2, 1))):
            h = self._lrelu(eng, eng.conv(self._s(e.conv[i]), [eng.im2col_strided(h, ks, st, pd)], ACT_NONE))
        h = self._lrelu(eng, eng.conv(self._s(e.fc[0]), [eng.flatten_chw(h)], ACT_NONE))
        h = self._lrelu(eng, eng.conv(self._s(e.embed[0]), [h, y], ACT_NONE))
        return eng.conv(self._s(e.z_loc), [h], ACT_NONE), self._logscale(eng, eng.conv(self._s(e.z_logscale), [h], ACT_NONE))

    def _prior(self, eng, y, t=None):
        """CondPrior.forward, simple_vae.py:91-100."""
        p = self.decoder.prior
        h = self._lrelu(eng, eng.conv(self._s(p.fc[0]), [y], ACT_NONE))
        h = self._lrelu(eng, eng.conv(self._s(p.fc[2]), [h], ACT_NONE))
        return (eng.conv(self._s(p.z_loc), [h], ACT_NONE), self._logscale(eng, eng.conv(self._s(p.z_logscale), [h], ACT_NONE), t),
                eng.conv(self._s(p.p_feat), [h], ACT_NONE))

    def _prior_stats(self, eng, y, B, t, drop1):
        y1 = y if drop1 == 1 else eng.scale_channels(y, 2, drop1)
        if self.cond_prior:
            return self._prior(eng, y1, t)
        d = self.decoder
        # (rg=True although the N(0,I) prior is a buffer: the fused KL backward writes all four gradients, these two unread)
        p_loc = eng.from_nchw(d.p_loc.repeat(B, 1)[:, :, None, None].contiguous(), rg=True)
        p_ls = eng.from_nchw(d.p_scale.log().repeat(B, 1)[:, :, None, None].contiguous(), rg=True)
        if t is not None:
            p_ls = eng.unary(p_ls, UNARY_ADD, float(np.log(t)))
        return p_loc, p_ls, None

    def _decode_z(self, eng, y, z, p_feat, drop2):
        """Decoder.forward from the latent on (simple_vae.py:305-311); the ReLUs are fused into the consumers."""
        d = self.decoder
        y2 = y if drop2 == 1 else eng.scale_channels(y, 2, drop2)
        segs = [p_feat, z, y2] if self.cond_prior else [z, y2]
        h = eng.conv(self._s(d.fc[0]), segs, ACT_NONE)
        h = eng.conv(self._s(d.fc[2]), [h], ACT_RELU)
        h = eng.unflatten_chw(eng.unary(h, ACT_RELU), self.hidden_dim // 4, 4, 4)
        h = eng.conv(self._s(d.conv[1]), [eng.upsample(h, 8)], ACT_NONE)
        h = eng.conv(self._s(d.conv[4]), [eng.upsample(h, 16)], ACT_RELU)
        h = eng.upsample(eng.unary(h, ACT_RELU), 32)
        h = eng.conv(self._s(d.conv[7]), [eng.im2col_strided(h, 5, 1, 2)], ACT_NONE)
        return eng.unary(h, ACT_RELU)

    def _eps(self, eng, shape_nc):
        src = self.__dict__["noise"]
        if src is None:
            return None
        e = src.pop(0)
        assert tuple(e.shape) == tuple(shape_nc), (tuple(e.shape), shape_nc)
        return eng.from_nchw(e.to(eng.device, torch.float32)[:, :, None, None].contiguous())

    # ------------------------------------------------------------------ training forward (VAE.forward, simple_vae.py:343-352)
    def _run_forward(self, x, parents, beta, record):
        eng = self.engine()
        eng.begin()
        eng.recording = record
        eng.prepare_weights(force=record)
        if self.__dict__["noise"] is None:
            eng.rng_advance(1)
        xin = eng.from_nchw(x.to(eng.device), rg=False)
        y = self._vec(eng, parents)
        drop = (1, 1)
        if self.training and self.cond_prior:
            drop = self.decoder.drop_cond()
        B, R, Cx = xin.n, xin.h, xin.c
        lib = eng.lib
        q_loc, q_ls = self._encode(eng, xin, y)
        p_loc, p_ls, p_feat = self._prior_stats(eng, y, B, None, drop[0])
        nch = lib.reparam_kl_chunks(1, 1, self.z_dim)
        kl_ptr = eng.new_f32(B * nch)
        z = eng.reparam_kl(q_loc, q_ls, p_loc, p_ls, self._eps(eng, (B, self.z_dim)), 1, 0.0, kl_ptr, nch)
        h = self._decode_z(eng, y, z, p_feat, drop[1])
        params = self._likelihood_params(eng, h)
        nchunk = lib.like_chunks(R, R)
        nll_ptr = eng.new_f32(B * nchunk)
        if getattr(self.likelihood, "logit_space", False):
            # dequantisation noise (simple_vae.py:222): injected for tests (`dequant_noise`, NCHW in [0,1)), else Philox
            # uniforms keyed by a snapshot of this pass's state, which the backward pass reads again
            un = self.__dict__.get("dequant_noise")
            u_nt = None if un is None else eng.from_nchw(un.to(eng.device, torch.float32).contiguous(), rg=False)
            u = NULL_VIEW if u_nt is None else u_nt.cv()
            eng.rng_ptr()
            snap = eng.rng.clone()
            self.__dict__["_gauss_noise"] = (u, snap, u_nt)  # (u_nt keeps the injected tensor alive until the backward pass)
            lib.gauss_nll_fwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), u, snap.data_ptr(), 977, nll_ptr, eng.stream)
        elif self.likelihood.kind == "dgauss":
            lib.dgauss_nll_fwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), nll_ptr, eng.stream)
        else:
            lib.dmol_nll_fwd(eng.dt, B, R, R, params.cv(), xin.cv(), nll_ptr, eng.stream)
        out3 = torch.empty(3, dtype=torch.float32, device=eng.device)
        dims = float(Cx * R * R)
        lib.elbo_finalize(B, nll_ptr, nchunk, dims, kl_ptr, nch, dims, float(beta), self.__dict__.get("_beta_dev"),
                          out3.data_ptr(), eng.stream)
        eng.launches += 2
        eng.recording = False
        self.__dict__["_saved"] = (params, xin, B, R, Cx, dims)
        self.__dict__["_saved_gen"] = eng.generation
        return out3

    # ------------------------------------------------------------------ inference API
    def _like_sample(self, eng, h, return_loc, t):
        """DGaussNet.sample of simple_vae.py:162-171 (the temperature IS applied when return_loc=False); DmolNet.sample is
        the one HVAE uses (dmol.py:234-245)."""
        if self.likelihood.kind != "dgauss":
            return self._sample_likelihood(eng, h, return_loc, t)
        params = self._likelihood_params(eng, h)
        B, R, Cx = h.n, h.h, self.input_channels
        xo = torch.empty((B, Cx, R, R), dtype=torch.float32, device=eng.device)
        so = torch.empty_like(xo)
        logt = 0.0 if (return_loc or t is None) else float(np.log(t))
        if getattr(self.likelihood, "logit_space", False):  # GaussNet.sample: the temperature scales `scale` in both modes
            eng.lib.gauss_sample(eng.dt, B, R, R, Cx, params.cv(), 0.0 if t is None else float(np.log(t)),
                                 None if return_loc else eng.rng_ptr(), 979, xo.data_ptr(), so.data_ptr(), eng.stream)
        else:
            eng.lib.dgauss_sample(eng.dt, B, R, R, Cx, params.cv(), logt, None if return_loc else eng.rng_ptr(), 979,
                                  xo.data_ptr(), so.data_ptr(), eng.stream)
        eng.launches += 1
        return xo, so

    @torch.no_grad()
    def sample(self, parents: Tensor, return_loc: bool = True, t: Optional[float] = None):
        """simple_vae.py:354-358."""
        eng = self._begin_inference()
        y = self._vec(eng, parents)
        p_loc, p_ls, p_feat = self._prior_stats(eng, y, y.n, t, 1)
        z = eng.sample_gaussian(p_loc, p_ls, self._eps(eng, (y.n, self.z_dim)), 2, 0.0)
        return self._like_sample(eng, self._decode_z(eng, y, z, p_feat, 1), return_loc, t)

    @torch.no_grad()
    def abduct(self, x: Tensor, parents: Tensor, cf_parents: Optional[Tensor] = None, alpha: float = 0.5,
               t: Optional[float] = None) -> List:
        """simple_vae.py:360-404.  NB the mediator here mixes VARIANCES linearly (a*var_q + (1-a)*var_p), unlike vae.py."""
        eng = self._begin_inference()
        xin = eng.from_nchw(x.to(eng.device), rg=False)
        y = self._vec(eng, parents)
        q_loc, q_ls = self._encode(eng, xin, y)
        z = eng.sample_gaussian(q_loc, q_ls, self._eps(eng, (xin.n, self.z_dim)), 3, 0.0)
        to2d = lambda v: eng.to_torch_cl(v).reshape(xin.n, -1)  # noqa: E731
        if not self.cond_prior:
            return [to2d(z)]
        if cf_parents is None:
            return [dict(z=to2d(z), q_loc=to2d(q_loc), q_logscale=to2d(q_ls))]
        p_loc, p_ls, _ = self._prior(eng, self._vec(eng, cf_parents), t)  # log t already inside p_ls; q_ls carries none
        o = eng.new(z.n, 1, 1, z.c, rg=False)
        eng.lib.mediator_mix(eng.dt, z.n, 1, 1, z.c, z.cv(), q_loc.cv(), q_ls.cv(), p_loc.cv(), p_ls.cv(), float(alpha),
                             float(t) if t is not None else -1.0, 0.0, 1, o.cv(), eng.stream)
        eng.launches += 1
        return [to2d(o)]

    def abduct_with_reconstruction(self, x: Tensor, parents: Tensor, t: Optional[float] = None):
        """What dscm.counterfactual asks of an image mechanism (see HVAE.abduct_with_reconstruction); with one latent and
        a three-conv decoder there is nothing worth sharing between the passes, so this is the two calls."""
        zs = self.abduct(x, parents, t=t)
        return zs, self.forward_latents([z["z"] if isinstance(z, dict) else z for z in zs], parents)

    def forward_latents_pair(self, latents: List[Tensor], parents_a: Tensor, parents_b: Tensor, t: Optional[float] = None):
        return self.forward_latents(latents, parents_a, t=t), self.forward_latents(latents, parents_b, t=t)

    @torch.no_grad()
    def forward_latents(self, latents: List[Tensor], parents: Tensor, return_loc: bool = True, t: Optional[float] = None):
        """simple_vae.py:406-415."""
        eng = self._begin_inference()
        y = self._vec(eng, parents)
        p_feat = self._prior(eng, y, t)[2] if self.cond_prior else None
        z = eng.from_nchw(latents[0].to(eng.device, torch.float32).reshape(y.n, -1, 1, 1).contiguous(), rg=False)
        return self._like_sample(eng, self._decode_z(eng, y, z, p_feat, 1), return_loc, t)
