"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's config-1 model, ``src/simple_vae.py`` (VAE with a
strided-conv encoder, linear bottleneck and upsample+conv decoder, 234 690 parameters at the morphomnist preset).
Functional and ``state_dict``-driven like hvae_ref.py; every function cites the reference lines it follows.  Only
``tests/`` may import this module.  Pinned by tests/golden/simple_vae_c1.pt and simple_vae_c1x.pt (made from the imported reference by
oracle/make_golden.py).  Likelihood: the discretised Gaussian of simple_vae.py:103-171 (``x_like = *_dgauss``), or dmol.DmolNet via dmol_ref.py
(``*_dmol``, simple_vae.py:336-339).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = -9.0
LEAK = 0.01  # nn.LeakyReLU() default slope (simple_vae.py:13)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _vec(y):
    return y[:, :, 0, 0] if y.dim() > 2 else y  # simple_vae.py:64-65, 95-96, 283-284


def gaussian_kl(q_loc, q_ls, p_loc, p_ls):
    """simple_vae.py:17-26."""
    return -0.5 + p_ls - q_ls + 0.5 * (q_ls.exp().pow(2) + (q_loc - p_loc).pow(2)) / p_ls.exp().pow(2)


def encode(sd, x, y, t=None):
    """Encoder.forward, simple_vae.py:59-70: 5x5/s2/p1 -> 3x3/s2/p1 -> 3x3/s2/p1 (LeakyReLU after each), fc, embed."""
    h = F.leaky_relu(F.conv2d(x, sd["encoder.conv.0.weight"], sd["encoder.conv.0.bias"], stride=2, padding=1), LEAK)
    h = F.leaky_relu(F.conv2d(h, sd["encoder.conv.2.weight"], sd["encoder.conv.2.bias"], stride=2, padding=1), LEAK)
    h = F.leaky_relu(F.conv2d(h, sd["encoder.conv.4.weight"], sd["encoder.conv.4.bias"], stride=2, padding=1), LEAK)
    h = F.leaky_relu(_lin(sd, "encoder.fc.0", h.reshape(x.size(0), -1)), LEAK)
    h = F.leaky_relu(_lin(sd, "encoder.embed.0", torch.cat((h, _vec(y)), dim=-1)), LEAK)
    loc, ls = _lin(sd, "encoder.z_loc", h), _lin(sd, "encoder.z_logscale", h).clamp(min=EPS)
    if t is not None:
        ls = ls + math.log(t)
    return loc, ls


def cond_prior(sd, y, t=None):
    """CondPrior.forward, simple_vae.py:91-100."""
    h = F.leaky_relu(_lin(sd, "decoder.prior.fc.0", _vec(y)), LEAK)
    h = F.leaky_relu(_lin(sd, "decoder.prior.fc.2", h), LEAK)
    loc, ls = _lin(sd, "decoder.prior.z_loc", h), _lin(sd, "decoder.prior.z_logscale", h).clamp(min=EPS)
    if t is not None:
        ls = ls + math.log(t)
    return loc, ls, _lin(sd, "decoder.prior.p_feat", h)


def decode(sd, hp, y, z=None, t=None, drop=(1, 1), eps=None):
    """Decoder.forward, simple_vae.py:282-311.  ``drop`` = (p1, p2) of drop_cond; ``eps`` replaces randn when z is None."""
    y = _vec(y)
    y1, y2 = y.clone(), y.clone()
    y1[:, 2:] = y1[:, 2:] * drop[0]
    y2[:, 2:] = y2[:, 2:] * drop[1]
    if hp.cond_prior:
        p_loc, p_ls, p_feat = cond_prior(sd, y1, t)
    else:
        p_loc = sd["decoder.p_loc"].repeat(y.shape[0], 1)
        p_ls = sd["decoder.p_scale"].log().repeat(y.shape[0], 1)
        if t is not None:
            p_ls = p_ls + math.log(t)
    if z is None:
        e = torch.randn_like(p_loc) if eps is None else eps
        z = p_loc + p_ls.exp() * e
    if hp.cond_prior:
        z = torch.cat((p_feat, z), dim=-1)
    h = torch.cat((z, y2), dim=-1)
    h = F.relu(_lin(sd, "decoder.fc.0", h))
    h = F.relu(_lin(sd, "decoder.fc.2", h)).reshape(h.size(0), -1, 4, 4)
    for i in (1, 4, 7):  # Upsample(x2, nearest) -> conv -> ReLU, simple_vae.py:270-280 (3x3, 3x3, 5x5/p2)
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        w = sd[f"decoder.conv.{i}.weight"]
        h = F.relu(F.conv2d(h, w, sd[f"decoder.conv.{i}.bias"], padding=w.shape[-1] // 2))
    return h, (p_loc, p_ls)


def _approx_cdf(x):
    return 0.5 * (1.0 + torch.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * torch.pow(x, 3))))


def like_params(sd, h, t=None):
    """DGaussNet.forward, simple_vae.py:130-134."""
    loc = F.conv2d(h, sd["likelihood.x_loc.weight"], sd["likelihood.x_loc.bias"])
    ls = F.conv2d(h, sd["likelihood.x_logscale.weight"], sd["likelihood.x_logscale.bias"]).clamp(min=EPS)
    if t is not None:
        ls = ls + math.log(t)
    return loc, ls


def _is_dmol(sd):
    return "likelihood.conv.weight" in sd  # x_like = *_dmol: dmol.DmolNet (simple_vae.py:336-339)


def _is_logit(hp):
    return hp is not None and getattr(hp, "x_like", "diag_dgauss").split("_")[1] == "gauss"  # GaussNet, simple_vae.py:332-333


def gauss_nll(sd, h, x, u=None):
    """GaussNet.nll, simple_vae.py:215-229: dequantise with u ~ U[0,1), logit(x / 256) (torch's SigmoidTransform.inv clamps
    its argument to [tiny, 1 - eps]), Normal log-density summed over (C,H,W) / D.  No log-determinant term."""
    loc, ls = like_params(sd, h)
    v = ((x + 1.0) * 127.5 + (torch.rand_like(x) if u is None else u)) / 256.0
    fi = torch.finfo(v.dtype)
    v = v.clamp(min=fi.tiny, max=1.0 - fi.eps)
    tgt = v.log() - (-v).log1p()
    lp = -0.5 * ((tgt - loc) / ls.exp()) ** 2 - ls - 0.5 * math.log(2 * math.pi)
    return -1.0 * lp.sum(dim=(1, 2, 3)) / np.prod(x.shape[1:])


def gauss_sample(sd, h, return_loc=True, t=None, eps=None):
    """GaussNet.sample, simple_vae.py:231-238 (the temperature enters the returned scale in both modes)."""
    loc, ls = like_params(sd, h, t)
    x = loc if return_loc else loc + ls.exp() * (torch.randn_like(loc) if eps is None else eps)
    x = torch.sigmoid(x) * 256.0
    return torch.clamp((x - 128) / 128, min=-1.0, max=1.0), ls.exp()


def nll(sd, h, x, hp=None, u=None):
    """DGaussNet.nll, simple_vae.py:141-160 (or DmolNet.nll, dmol.py:229-232; or GaussNet.nll)."""
    if _is_dmol(sd):
        from . import dmol_ref
        return dmol_ref.dmolnet_nll(sd, h, x)
    if _is_logit(hp):
        return gauss_nll(sd, h, x, u)
    loc, ls = like_params(sd, h)
    c, inv = x - loc, torch.exp(-ls)
    cp, cm = _approx_cdf(inv * (c + 1.0 / 255.0)), _approx_cdf(inv * (c - 1.0 / 255.0))
    lp = torch.where(x < -0.999, torch.log(cp.clamp(min=1e-12)),
                     torch.where(x > 0.999, torch.log((1.0 - cm).clamp(min=1e-12)), torch.log((cp - cm).clamp(min=1e-12))))
    return -1.0 * lp.mean(dim=(1, 2, 3))


def like_sample(sd, h, return_loc=True, t=None, eps=None, hp=None):
    """DGaussNet.sample, simple_vae.py:162-171 (here t IS applied when return_loc=False); DmolNet.sample, dmol.py:234-245."""
    if _is_dmol(sd):
        from . import dmol_ref
        return dmol_ref.dmolnet_sample(sd, h, return_loc=return_loc, t=t)
    if _is_logit(hp):
        return gauss_sample(sd, h, return_loc, t, eps)
    if return_loc:
        x, ls = like_params(sd, h)
    else:
        loc, ls = like_params(sd, h, t)
        x = loc + torch.exp(ls) * (torch.randn_like(loc) if eps is None else eps)
    return torch.clamp(x, min=-1.0, max=1.0), ls.exp()


def forward(sd, hp, x, parents, beta=1, eps=None, drop=(1, 1), u=None):
    """VAE.forward, simple_vae.py:343-352."""
    q_loc, q_ls = encode(sd, x, parents)
    e = torch.randn_like(q_loc) if eps is None else eps
    z = q_loc + q_ls.exp() * e
    h, (p_loc, p_ls) = decode(sd, hp, parents, z=z, drop=drop)
    nll_pp = nll(sd, h, x, hp, u)
    kl_pp = gaussian_kl(q_loc, q_ls, p_loc, p_ls).sum(dim=-1) / np.prod(x.shape[1:])
    return dict(elbo=nll_pp.mean() + beta * kl_pp.mean(), nll=nll_pp.mean(), kl=kl_pp.mean())


def sample(sd, hp, parents, return_loc=True, t=None, eps=None):
    """VAE.sample, simple_vae.py:354-358."""
    h, _ = decode(sd, hp, parents, t=t, eps=eps)
    return like_sample(sd, h, return_loc, t=t, hp=hp)


def abduct(sd, hp, x, parents, cf_parents=None, alpha=0.5, t=None, eps=None):
    """VAE.abduct, simple_vae.py:360-404 (note r_var = a*var_q + (1-a)*var_p here, unlike vae.py)."""
    q_loc, q_ls = encode(sd, x, parents)
    e = torch.randn_like(q_loc) if eps is None else eps
    z = q_loc + q_ls.exp() * e
    if not hp.cond_prior:
        return [z.detach()]
    if cf_parents is None:
        return [dict(z=z, q_loc=q_loc, q_logscale=q_ls)]
    p_loc, p_ls, _ = cond_prior(sd, cf_parents, t)
    q_scale = q_ls.exp()
    u = (z - q_loc) / q_scale
    r_loc = alpha * q_loc + (1 - alpha) * p_loc
    r_scale = (alpha * q_scale.pow(2) + (1 - alpha) * p_ls.exp().pow(2)).sqrt()
    if t is not None:
        r_scale = r_scale * t
    return [r_loc + r_scale * u]


def forward_latents(sd, hp, latents, parents, return_loc=True, t=None):
    """VAE.forward_latents, simple_vae.py:406-415."""
    h, _ = decode(sd, hp, parents, z=latents[0], t=t)
    return like_sample(sd, h, return_loc, t=t, hp=hp)
