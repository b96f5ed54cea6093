"""Functional CPU restatement of the reference HVAE (src/vae.py) -- ORACLE, test-only.

Everything here works on a plain ``state_dict`` (reference key names, e.g.
``decoder.blocks.3.prior.conv.1.weight``) and an hparams namespace; there are
no nn.Modules.  Convolutions / pooling / interpolation are the stock ATen ops
the reference itself dispatches -- this file restates the
reference's *own* arithmetic and wiring:

  gaussian_kl / sample_gaussian ........ vae.py:14-30
  Block ................................ vae.py:33-84
  Encoder .............................. vae.py:87-134
  DecoderBlock ......................... vae.py:137-192
  Decoder.forward / drop_cond .......... vae.py:222-301, 310-319
  DGaussNet ............................ vae.py:322-422
  HVAE.forward/sample/abduct/forward_latents  vae.py:439-522

Pinned by tests/golden/*.pt (made by oracle/make_golden.py from the imported
reference).  Noise is injectable: ``noise`` is None (draw torch.randn_like on
the global generator in the reference's order) or a list consumed in order.
"""
import math
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

MIN_LOGSCALE = -9.0  # vae.py:11 (EPS)


# --------------------------------------------------------------------------- arch
def _is_light(hp):
    return "ukbb" in hp.hps  # vae.py:428 sets args.vr


def encoder_spec(hp):
    """[(in_width, bottleneck, out_width, down_rate)] per encoder block (vae.py:92-120)."""
    plan = []
    for i, stage in enumerate(hp.enc_arch.split(",")):
        lo = stage.index("b") + 1
        hi = stage.index("d") if "d" in stage else None
        plan += [(hp.widths[i], None)] * int(stage[lo:hi])
        if "d" in stage:  # the rate is ONE character
            plan.append((hp.widths[i + 1], int(stage[stage.index("d") + 1])))
    out = []
    for i, (w, d) in enumerate(plan):
        w_in = plan[max(0, i - 1)][0]
        out.append((w_in, int(w_in / hp.bottleneck), w, d))
    return out


def decoder_spec(hp):
    """[(res, in_width, out_width)] per decoder block + bias table (vae.py:198-218)."""
    rev = hp.widths[::-1]
    plan = []
    for i, stage in enumerate(hp.dec_arch.split(",")):
        res = int(stage.split("b")[0])
        plan += [(res, rev[i])] * int(stage[stage.index("b") + 1:])
    blocks = [(r, w, plan[min(len(plan) - 1, i + 1)][1]) for i, (r, w) in enumerate(plan)]
    all_res = [int(r) for r in np.unique([r for r, _ in plan])]
    bias = [(r, rev[i]) for i, r in enumerate(all_res) if r <= hp.bias_max_res]
    return blocks, bias


def conv_slots(light):
    """Indices of the Conv2d entries inside Block.conv (vae.py:49-68)."""
    return (1, 3) if light else (1, 3, 5, 7)


# --------------------------------------------------------------------------- init
def init_state_dict(hp, likelihood="dgauss", zero_conv_bias=True):
    """Fresh parameters, consuming torch's global RNG in the reference's construction order
    (encoder: stem, blocks | decoder: per block prior, posterior, z_proj, z_feat_proj, conv | likelihood),
    then the rescalings of vae.py:121-122, 303-308 and main.py:51-55's bias zeroing."""
    light = _is_light(hp)
    sd = OrderedDict()

    def conv(key, ci, co, k):
        m = torch.nn.Conv2d(ci, co, k)  # default init = what the reference gets
        sd[key + ".weight"] = m.weight.detach().clone()
        sd[key + ".bias"] = m.bias.detach().clone()

    def block(prefix, ci, b, co, k, residual, down):
        if light:
            conv(f"{prefix}.conv.1", ci, b, k)
            conv(f"{prefix}.conv.3", b, co, k)
        else:
            conv(f"{prefix}.conv.1", ci, b, 1)
            conv(f"{prefix}.conv.3", b, b, k)
            conv(f"{prefix}.conv.5", b, b, k)
            conv(f"{prefix}.conv.7", b, co, 1)
        if residual and (down or ci > co):
            conv(f"{prefix}.width_proj", ci, co, 1)

    enc = encoder_spec(hp)
    conv("encoder.stem", hp.input_channels, hp.widths[0], 7)
    for i, (ci, b, co, d) in enumerate(enc):
        block(f"encoder.blocks.{i}", ci, b, co, 3, True, d)
    last = conv_slots(light)[-1]
    for i in range(len(enc)):
        sd[f"encoder.blocks.{i}.conv.{last}.weight"] *= np.sqrt(1 / len(enc))

    dec, bias = decoder_spec(hp)
    for i, (res, w, w_next) in enumerate(dec):
        p = f"decoder.blocks.{i}"
        b = int(w / hp.bottleneck)
        k = 3 if res > 2 else 1
        block(p + ".prior", w + hp.context_dim if hp.cond_prior else w, b, 2 * hp.z_dim + w, k, False, None)
        if res <= hp.z_max_res:
            block(p + ".posterior", 2 * w + hp.context_dim, b, 2 * hp.z_dim, k, False, None)
        conv(p + ".z_proj", hp.z_dim + hp.context_dim, w, 1)
        if not hp.q_correction:
            conv(p + ".z_feat_proj", hp.z_dim + w, w_next, 1)
        block(p + ".conv", w, b, w_next, k, True, None)
    s = np.sqrt(1 / len(dec))
    for i in range(len(dec)):
        p = f"decoder.blocks.{i}"
        sd[p + ".z_proj.weight"] *= s
        sd[f"{p}.conv.conv.{last}.weight"] *= s
        sd[f"{p}.prior.conv.{last}.weight"] *= 0.0
    for j, (res, w) in enumerate(bias):
        sd[f"decoder.bias.{j}"] = torch.zeros(1, w, res, res)

    if likelihood == "dgauss":
        conv("likelihood.x_loc", hp.widths[0], hp.input_channels, 1)
        conv("likelihood.x_logscale", hp.widths[0], hp.input_channels, 1)
        if hp.input_channels == 3:
            conv("likelihood.channel_coeffs", hp.widths[0], 3, 1)
        if hp.std_init > 0:  # vae.py:335-337
            sd["likelihood.x_logscale.weight"].zero_()
            sd["likelihood.x_logscale.bias"].fill_(float(np.log(hp.std_init)))
    else:  # dmol.py:218-226
        conv("likelihood.conv", hp.widths[0], 100, 1)
    if zero_conv_bias:
        for k_ in sd:
            if k_.endswith(".bias") and not k_.startswith("decoder.bias"):
                if not (hp.std_init > 0 and k_ == "likelihood.x_logscale.bias"):
                    sd[k_].zero_()
    # decoder.bias.* must come after decoder.blocks.* and before likelihood in key order
    ordered = OrderedDict()
    for k_ in sd:
        if not k_.startswith("likelihood") and not k_.startswith("decoder.bias"):
            ordered[k_] = sd[k_]
    for k_ in sd:
        if k_.startswith("decoder.bias"):
            ordered[k_] = sd[k_]
    for k_ in sd:
        if k_.startswith("likelihood"):
            ordered[k_] = sd[k_]
    return ordered


# --------------------------------------------------------------------------- math
def gaussian_kl(q_loc, q_logscale, p_loc, p_logscale):
    """vae.py:18-25 -- no clamps on the logscales."""
    return (-0.5 + p_logscale - q_logscale
            + 0.5 * (q_logscale.exp().pow(2) + (q_loc - p_loc).pow(2)) / p_logscale.exp().pow(2))


class _Noise:
    """Hands out eps tensors in draw order and records them (vae.py:30 uses randn_like)."""

    def __init__(self, source=None):
        self.source = list(source) if source is not None else None
        self.drawn = []

    def __call__(self, like):
        if self.source is None:
            e = torch.randn_like(like)
        else:
            e = self.source.pop(0).to(like.dtype)
            assert e.shape == like.shape, (e.shape, like.shape)
        self.drawn.append(e)
        return e


def _reparam(loc, logscale, noise):
    return loc + logscale.exp() * noise(loc)


def _act(light, t):
    return F.relu(t) if light else F.gelu(t)


def _block(sd, prefix, x, light, k, residual, down):
    """vae.py:73-84."""
    pad = 0 if k == 1 else 1
    h = x
    if light:
        h = F.conv2d(_act(True, h), sd[f"{prefix}.conv.1.weight"], sd[f"{prefix}.conv.1.bias"], padding=pad)
        h = F.conv2d(_act(True, h), sd[f"{prefix}.conv.3.weight"], sd[f"{prefix}.conv.3.bias"], padding=pad)
    else:
        h = F.conv2d(_act(False, h), sd[f"{prefix}.conv.1.weight"], sd[f"{prefix}.conv.1.bias"])
        h = F.conv2d(_act(False, h), sd[f"{prefix}.conv.3.weight"], sd[f"{prefix}.conv.3.bias"], padding=pad)
        h = F.conv2d(_act(False, h), sd[f"{prefix}.conv.5.weight"], sd[f"{prefix}.conv.5.bias"], padding=pad)
        h = F.conv2d(_act(False, h), sd[f"{prefix}.conv.7.weight"], sd[f"{prefix}.conv.7.bias"])
    if residual:
        if x.shape[1] != h.shape[1]:
            x = F.conv2d(x, sd[f"{prefix}.width_proj.weight"], sd[f"{prefix}.width_proj.bias"])
        h = x + h
    if down:
        if isinstance(down, float):
            h = F.adaptive_avg_pool2d(h, int(h.shape[-1] / down))
        else:
            h = F.avg_pool2d(h, kernel_size=down, stride=down)
    return h


def encode(sd, hp, x):
    """vae.py:125-134 -> {res: activation}."""
    light = _is_light(hp)
    h = F.conv2d(x, sd["encoder.stem.weight"], sd["encoder.stem.bias"], padding=3)
    acts = {}
    for i, (_, _, _, d) in enumerate(encoder_spec(hp)):
        h = _block(sd, f"encoder.blocks.{i}", h, light, 3, True, d)
        r = h.shape[2]
        if r % 2 and r > 1:
            h = F.pad(h, [0, 1, 0, 1])
        acts[h.size(-1)] = h
    return acts


def decode(sd, hp, parents, acts=None, t=None, abduct=False, latents=None, noise=None,
           drop=(1, 1), trace=None):
    """Decoder.forward, vae.py:222-301.  ``drop`` = (p_sto, p_det) as drawn by drop_cond
    (only honoured for morphomnist presets, vae.py:219, 244-249).  ``trace`` (a dict) collects
    per-block intermediates for the fixtures."""
    light = _is_light(hp)
    noise = noise if isinstance(noise, _Noise) else _Noise(noise)
    latents = [] if latents is None else latents
    blocks, bias_tab = decoder_spec(hp)
    bias = {r: sd[f"decoder.bias.{j}"] for j, (r, _) in enumerate(bias_tab)}
    zd = hp.z_dim
    logt = None if t is None else torch.tensor(t).log()
    drop_y = "morphomnist" in hp.hps
    h = z = bias[1].repeat(parents.shape[0], 1, 1, 1)
    stats = []
    b = 0
    for i, (res, w, w_next) in enumerate(blocks):
        p = f"decoder.blocks.{i}"
        k = 3 if res > 2 else 1
        pa = parents[..., :res, :res]
        pa_sto = pa
        if drop_y:
            pa_sto = pa.clone()
            pa_sto[:, 2:] = pa_sto[:, 2:] * drop[0]
        if h.size(-1) < res:
            b = bias[res] if res in bias else 0
            h = b + F.interpolate(h, scale_factor=res / h.shape[-1])
        if hp.q_correction:
            p_in = h
        else:
            p_in = b + F.interpolate(z, scale_factor=res / z.shape[-1]) if z.size(-1) < res else z
        # forward_prior, vae.py:169-183
        pin = torch.cat([p_in, pa_sto], dim=1) if hp.cond_prior else p_in
        pout = _block(sd, p + ".prior", pin, light, k, False, None)
        p_loc, p_ls, p_feat = pout[:, :zd], pout[:, zd:2 * zd], pout[:, 2 * zd:]
        if logt is not None:
            p_ls = p_ls + logt
        stochastic = res <= hp.z_max_res
        if stochastic:
            if acts is not None:
                # forward_posterior, vae.py:185-192
                qin = torch.cat([h, pa, acts[res]], dim=1)
                q_loc, q_ls = _block(sd, p + ".posterior", qin, light, k, False, None).chunk(2, dim=1)
                if logt is not None:
                    q_ls = q_ls + logt
                z = _reparam(q_loc, q_ls, noise)
                stat = dict(kl=gaussian_kl(q_loc, q_ls, p_loc, p_ls))
                if abduct:
                    stat["z"] = dict(z=z, q_loc=q_loc, q_logscale=q_ls) if hp.cond_prior else z
                stats.append(stat)
            else:
                zi = latents[i] if i < len(latents) else None
                if zi is not None:
                    z = zi
                else:
                    z = _reparam(p_loc, p_ls, noise)
                    # NB vae.py:281-289: the {p_loc,p_logscale} record is only appended on the
                    # *exception* path (list too short), not when the entry is None.
                    if i >= len(latents) and abduct and hp.cond_prior:
                        stats.append(dict(z=dict(p_loc=p_loc, p_logscale=p_ls)))
        else:
            z = p_loc
        if trace is not None:
            trace.setdefault("z", []).append(z)
            trace.setdefault("p_loc", []).append(p_loc)
        h = h + p_feat
        h = h + F.conv2d(torch.cat([z, pa], dim=1), sd[p + ".z_proj.weight"], sd[p + ".z_proj.bias"])
        h = _block(sd, p + ".conv", h, light, k, True, None)
        if not hp.q_correction and i + 1 < len(blocks):
            z = F.conv2d(torch.cat([z, p_feat], dim=1), sd[p + ".z_feat_proj.weight"], sd[p + ".z_feat_proj.bias"])
    return h, stats


# --------------------------------------------------------------------------- DGauss likelihood
def dgauss_params(sd, hp, h, x=None, t=None):
    """DGaussNet.forward, vae.py:352-386."""
    loc = F.conv2d(h, sd["likelihood.x_loc.weight"], sd["likelihood.x_loc.bias"])
    ls = F.conv2d(h, sd["likelihood.x_logscale.weight"], sd["likelihood.x_logscale.bias"]).clamp(min=MIN_LOGSCALE)
    if "likelihood.channel_coeffs.weight" in sd:
        c = torch.tanh(F.conv2d(h, sd["likelihood.channel_coeffs.weight"], sd["likelihood.channel_coeffs.bias"]))
        if x is None:
            r = loc[:, 0].clamp(-1, 1)
            g = (loc[:, 1] + c[:, 0] * r).clamp(-1, 1)
            bl = (loc[:, 2] + c[:, 1] * r + c[:, 2] * g).clamp(-1, 1)
        else:
            r = loc[:, 0]
            g = loc[:, 1] + c[:, 0] * x[:, 0]
            bl = loc[:, 2] + c[:, 1] * x[:, 0] + c[:, 2] * x[:, 1]
        loc = torch.stack([r, g, bl], dim=1)
    if t is not None:
        ls = ls + torch.tensor(t).log()
    return loc, ls


def _tanh_cdf(u):
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (u + 0.044715 * torch.pow(u, 3))))


def dgauss_nll_from_params(loc, ls, x):
    """vae.py:393-411 given (loc, logscale)."""
    d = x - loc
    inv = torch.exp(-ls)
    cp = _tanh_cdf(inv * (d + 1.0 / 255.0))
    cm = _tanh_cdf(inv * (d - 1.0 / 255.0))
    lp = torch.where(
        x < -0.999, torch.log(cp.clamp(min=1e-12)),
        torch.where(x > 0.999, torch.log((1.0 - cm).clamp(min=1e-12)), torch.log((cp - cm).clamp(min=1e-12))))
    return -1.0 * lp.mean(dim=(1, 2, 3))


def dgauss_nll(sd, hp, h, x):
    loc, ls = dgauss_params(sd, hp, h, x)
    return dgauss_nll_from_params(loc, ls, x)


def dgauss_sample(sd, hp, h, return_loc=True, t=None, noise=None):
    """vae.py:413-422.  With return_loc=False the reference passes ``t`` positionally into the
    ``x`` slot: for C=1 the temperature is dropped; reproduced as-is."""
    if return_loc:
        x, ls = dgauss_params(sd, hp, h)
    else:
        loc, ls = dgauss_params(sd, hp, h, x=t)
        x = loc + torch.exp(ls) * (noise(loc) if noise is not None else torch.randn_like(loc))
    return x.clamp(-1.0, 1.0), ls.exp()


# --------------------------------------------------------------------------- model-level API
def _likelihood_kind(sd):
    return "dmol" if "likelihood.conv.weight" in sd else "dgauss"


def likelihood_nll(sd, hp, h, x):
    if _likelihood_kind(sd) == "dmol":
        from . import dmol_ref
        return dmol_ref.dmolnet_nll(sd, h, x)
    return dgauss_nll(sd, hp, h, x)


def likelihood_sample(sd, hp, h, return_loc=True, t=None):
    if _likelihood_kind(sd) == "dmol":
        from . import dmol_ref
        return dmol_ref.dmolnet_sample(sd, h, return_loc=return_loc, t=t)
    return dgauss_sample(sd, hp, h, return_loc=return_loc, t=t)


def hvae_forward(sd, hp, x, parents, beta=1, noise=None, drop=(1, 1), want_stats=False):
    """HVAE.forward, vae.py:439-458 -> dict(elbo, nll, kl) in nats/dim."""
    acts = encode(sd, hp, x)
    trace = {} if want_stats else None
    h, stats = decode(sd, hp, parents, acts=acts, noise=noise, drop=drop, trace=trace)
    nll = likelihood_nll(sd, hp, h, x)
    if hp.kl_free_bits > 0:
        fb = torch.tensor(hp.kl_free_bits).type_as(nll)
        kl = 0.0
        for s in stats:
            kl = kl + torch.maximum(fb, s["kl"].sum(dim=(2, 3)).mean(dim=0)).sum()
    else:
        kl = torch.zeros_like(nll)
        for s in stats:
            kl = kl + s["kl"].sum(dim=(1, 2, 3))
    kl = (kl / np.prod(x.shape[1:])).mean()
    nll = nll.mean()
    out = dict(elbo=nll + beta * kl, nll=nll, kl=kl)
    if want_stats:
        out["_h"] = h
        out["_kl_maps"] = [s["kl"] for s in stats]
        out["_z"] = trace["z"]
    return out


def hvae_sample(sd, hp, parents, return_loc=True, t=None, noise=None):
    """vae.py:460-464."""
    h, _ = decode(sd, hp, parents, t=t, noise=noise)
    return likelihood_sample(sd, hp, h, return_loc, t=t)


def hvae_abduct(sd, hp, x, parents, cf_parents=None, alpha=0.5, t=None, noise=None):
    """vae.py:466-514: exogenous z list, cond-prior dict list, or mediator z* when cf_parents given."""
    noise = noise if isinstance(noise, _Noise) else _Noise(noise)
    acts = encode(sd, hp, x)
    _, qs = decode(sd, hp, parents, acts=acts, abduct=True, t=t, noise=noise)
    qs = [s["z"] for s in qs]
    if not (hp.cond_prior and cf_parents is not None):
        return qs
    _, ps = decode(sd, hp, cf_parents, abduct=True, t=t, noise=noise)
    ps = [s["z"] for s in ps]
    out = []
    for q, p in zip(qs, ps):
        q_scale = q["q_logscale"].exp()
        u = (q["z"] - q["q_loc"]) / q_scale
        p_var = p["p_logscale"].exp().pow(2)
        r_loc = alpha * q["q_loc"] + (1 - alpha) * p["p_loc"]
        r_scale = (alpha ** 2 * q_scale.pow(2) + (1 - alpha) ** 2 * p_var).sqrt()
        if t is not None:
            r_scale = r_scale * t
        out.append(r_loc + r_scale * u)
    return out


def hvae_forward_latents(sd, hp, latents, parents, t=None, noise=None):
    """vae.py:516-522."""
    h, _ = decode(sd, hp, parents, latents=latents, t=t, noise=noise)
    return likelihood_sample(sd, hp, h, t=t)


def drop_cond_draw():
    """vae.py:310-319: one categorical per step -> (p_sto, p_det)."""
    opt = int(torch.distributions.Categorical(torch.ones(3) / 3).sample())
    return {0: (0, 1), 1: (1, 0), 2: (1, 1)}[opt]


def count_params(sd):
    return sum(int(v.numel()) for v in sd.values())
