"""Functional CPU restatement of the image half of the DSCM (src/pgm/dscm.py) -- ORACLE, test-only.

  expand_parents ...... vae_preprocess, dscm.py:121-132 (minus the hard-coded .cuda())
  ukbb_parents ........ ukbb_preprocess, dscm.py:98-118 (min/max table: datasets.py:89-98)
  counterfactual ...... dscm.py:52-56 (+ notebook cell 9's cond_prior unwrapping / total effect)
  cf_particles ........ dscm.py:43-72 (MC mean / variance over particles)
  lagrangian .......... dscm.py:85-88

``pgm.counterfactual`` (Pyro) is outside the hot path: callers pass ``cf_parents`` directly.
"""
import torch

from . import hvae_ref

UKBB_MIN_MAX = {  # datasets.py:89-98 get_attr_max_min
    "age": (73.0, 44.0),
    "brain_volume": (1629520.0, 841919.0),
    "ventricle_volume": (157075.0, 7613.27001953125),
}
UKBB_LOG_STATS = {  # dscm.py:112-117
    "age": (4.112339973449707, 0.11769197136163712),
    "brain_volume": (13.965583801269531, 0.09537758678197861),
    "ventricle_volume": (10.345998764038086, 0.43127763271331787),
}


def ukbb_parents(pa):
    out = {}
    for k, v in pa.items():
        if k in ("mri_seq", "sex"):
            out[k] = v
            continue
        hi, lo = UKBB_MIN_MAX[k]
        raw = (v + 1) / 2 * (hi - lo) + lo
        mu, sd = UKBB_LOG_STATS[k]
        out[k] = (torch.log(raw.clamp(min=1e-12)) - mu) / sd
    return out


def expand_parents(pa, parents_x, input_res, ukbb=False):
    if ukbb:
        pa = ukbb_parents(pa)
    cols = [pa[k] if pa[k].dim() > 1 else pa[k][..., None] for k in parents_x]
    flat = torch.cat(cols, dim=1)
    return flat[..., None, None].repeat(1, 1, input_res, input_res).float()


def counterfactual(sd, hp, x, parents, cf_parents, t_abduct=1.0, noise=None, te_cf=False, alpha=0.65, t_u=None):
    """One abduction -> action -> prediction pass.  Returns dict(cf_x, rec_loc, rec_scale, u, zs)."""
    zs = hvae_ref.hvae_abduct(sd, hp, x, parents, t=t_abduct, noise=noise)
    if hp.cond_prior:  # utils.py:299 / notebook cell 9 unwrap the dicts
        zs = [z["z"] for z in zs]
    rec_loc, rec_scale = hvae_ref.hvae_forward_latents(sd, hp, zs, parents)
    cf_zs = zs
    if te_cf and hp.cond_prior:
        cf_zs = hvae_ref.hvae_abduct(sd, hp, x, parents, cf_parents=cf_parents, alpha=alpha, t=t_abduct, noise=noise)
    cf_loc, cf_scale = hvae_ref.hvae_forward_latents(sd, hp, cf_zs, cf_parents)
    if t_u is not None:
        cf_scale = cf_scale * t_u
    u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
    cf_x = torch.clamp(cf_loc + cf_scale * u, min=-1, max=1)
    return dict(cf_x=cf_x, rec_loc=rec_loc, rec_scale=rec_scale, cf_loc=cf_loc, cf_scale=cf_scale, u=u, zs=zs)


def cf_particles(sd, hp, x, parents, cf_parents_list, t_abduct=1.0, noises=None):
    """dscm.py:43-72 with P = len(cf_parents_list) particles -> (mean cf_x, var cf_x or None)."""
    P = len(cf_parents_list)
    sx = torch.zeros_like(x)
    sx2 = torch.zeros_like(x)
    for i, cfp in enumerate(cf_parents_list):
        o = counterfactual(sd, hp, x, parents, cfp, t_abduct, None if noises is None else noises[i])
        if P == 1:
            return o["cf_x"], None
        sx = sx + o["cf_x"]
        sx2 = sx2 + o["cf_x"] ** 2
    return sx / P, (sx2 - sx ** 2 / P) / P


def lagrangian(aux_loss, elbo, lmbda, eps, damping):
    """dscm.py:85-88."""
    sg = (eps - elbo).detach()
    return aux_loss - (lmbda - damping * sg) * (eps - elbo)


def dscm_forward(sd, hp, x, parents, cf_parents_list, beta, t_abduct=1.0, noise=None, aux_fn=None, lmbda=None, eps=None,
                 damping=0.0):
    """Image half of DSCM.forward (dscm.py:40-95) as ONE differentiable torch graph over the state dict ``sd``: factual
    ELBO (vae.py:439-458), then per particle abduct -> replay under cf parents -> replay under the observed parents ->
    dscm.py:55-56, particle mean / variance (dscm.py:58-72), and -- when ``aux_fn(cf_x)`` (standing in for the predictor's
    ``elbo_fn.differentiable_loss / B``, pyro-side) and the multiplier are given -- the Lagrangian of dscm.py:85-88.
    ``noise``: a single hvae_ref._Noise shared by all passes in the reference's draw order (factual first)."""
    out = hvae_ref.hvae_forward(sd, hp, x, parents, beta=beta, noise=noise)
    P = len(cf_parents_list)
    sx, sx2, cf_x = torch.zeros_like(x), torch.zeros_like(x), None
    for cfp in cf_parents_list:
        zs = hvae_ref.hvae_abduct(sd, hp, x, parents, t=t_abduct, noise=noise)
        if hp.cond_prior:
            zs = [z["z"] for z in zs]
        cf_loc, cf_scale = hvae_ref.hvae_forward_latents(sd, hp, zs, cfp)
        rec_loc, rec_scale = hvae_ref.hvae_forward_latents(sd, hp, zs, parents)
        u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
        cf_x = torch.clamp(cf_loc + cf_scale * u, min=-1, max=1)
        if P > 1:
            sx = sx + cf_x
            with torch.no_grad():
                sx2 = sx2 + cf_x ** 2
    res = dict(out)
    if P > 1:
        with torch.no_grad():
            res["var_cf_x"] = (sx2 - sx ** 2 / P) / P
        res["cf_x"] = sx / P
    else:
        res["var_cf_x"] = None
        res["cf_x"] = cf_x
    if aux_fn is not None:
        res["aux_loss"] = aux_fn(res["cf_x"])
        res["loss"] = lagrangian(res["aux_loss"], out["elbo"], lmbda, eps, damping)
    return res
