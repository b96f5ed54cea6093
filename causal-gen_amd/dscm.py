"""Image half of the deep structural causal model on MI355X -- the ``src/pgm/dscm.py`` surface.

``DSCM(args, pgm, predictor, vae)`` keeps the reference's constructor, members and ``forward(obs, do, elbo_fn,
cf_particles, t_abduct)`` contract (dscm.py:15-95).  ``pgm`` / ``predictor`` / ``elbo_fn`` stay duck-typed (the Pyro
parent mechanisms are outside the hot path); everything that touches pixels -- the factual ELBO, abduction,
the two latent replays, the pixel-noise step and the particle statistics -- runs through the HIP engine.
``counterfactual()`` is dscm.py:52-56 factored out, with the notebook's cond-prior unwrapping and total-effect switch
(SURVEY 3.4).
"""
from typing import Dict, Optional

import os

import torch
from torch import Tensor, nn

from . import _lib
from .engine import StaticParents, compact_parents

_UKBB_MIN_MAX = {"age": (73.0, 44.0), "brain_volume": (1629520.0, 841919.0), "ventricle_volume": (157075.0, 7613.27001953125)}
_UKBB_LOG_STATS = {"age": (4.112339973449707, 0.11769197136163712), "brain_volume": (13.965583801269531, 0.09537758678197861),
                   "ventricle_volume": (10.345998764038086, 0.43127763271331787)}


def ukbb_preprocess(pa: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """[-1,1]-normalised UKBB parents -> the log-standardised parents the HVAE was trained with (dscm.py:98-118).
    A handful of scalars per sample; plain host-side tensor math."""
    out = {}
    for k, v in pa.items():
        if k in ("mri_seq", "sex"):
            out[k] = v
            continue
        hi, lo = _UKBB_MIN_MAX[k]
        mu, sd = _UKBB_LOG_STATS[k]
        out[k] = (torch.log(((v + 1) / 2 * (hi - lo) + lo).clamp(min=1e-12)) - mu) / sd
    return out


def vae_preprocess(args, pa: Dict[str, Tensor]) -> Tensor:
    """Concatenate parents in ``args.parents_x`` order and expand to [B,ctx,R,R] on the GPU (dscm.py:121-132).  The
    expansion is a stride-0 view: the HVAE lays out [B,1,1,ctx] only (engine.from_parents)."""
    if "ukbb" in getattr(args, "dataset", ""):
        pa = ukbb_preprocess(pa)
    cols = [pa[k] if pa[k].dim() > 1 else pa[k][..., None] for k in args.parents_x]
    flat = torch.cat(cols, dim=1).float().cuda()
    return flat[..., None, None].expand(-1, -1, args.input_res, args.input_res)


def cf_pixels(x, rec_loc, rec_scale, cf_loc, cf_scale, sum_x=None, sum_x2=None):
    """dscm.py:55-63 as one fused launch: u = (x-rec_loc)/clamp(rec_scale,1e-12); clamp(cf_loc + cf_scale*u, -1, 1)."""
    lib = _lib.require_gpu()
    ts = [t.contiguous().float() for t in (x, rec_loc, rec_scale, cf_loc, cf_scale)]
    out = torch.empty_like(ts[0])
    lib.cf_pixels(out.numel(), *[t.data_ptr() for t in ts], out.data_ptr(),
                  sum_x.data_ptr() if sum_x is not None else None, sum_x2.data_ptr() if sum_x2 is not None else None,
                  torch.cuda.current_stream(out.device).cuda_stream)
    return out


@torch.no_grad()
def counterfactual(vae, x, parents, cf_parents, t_abduct=1.0, te_cf=False, alpha=0.65, t_u=None):
    """Abduction -> action -> prediction for one batch (dscm.py:52-56; notebook cell 9 for cond_prior / total effect)."""
    te = bool(te_cf and vae.cond_prior)
    if os.environ.get("CGEN_CF_REUSE", "1") != "0" and hasattr(vae, "abduct_with_reconstruction"):
        # the reconstruction replay would rebuild, from the same latents and parents, the hidden state the abduction pass
        # already holds: one decoder pass less per counterfactual, same bits
        zs, (rec_loc, rec_scale) = vae.abduct_with_reconstruction(x, parents, t=t_abduct)
        if vae.cond_prior:
            zs = [z["z"] for z in zs]
        if te:  # total effect: mediator latents from a second abduction under the counterfactual parents
            zs = vae.abduct(x, parents, cf_parents=cf_parents, alpha=alpha, t=t_abduct)
        cf_loc, cf_scale = vae.forward_latents(zs, cf_parents)
        if t_u is not None:
            cf_scale = cf_scale * t_u
        return cf_pixels(x.to(rec_loc.device), rec_loc, rec_scale, cf_loc, cf_scale)
    zs = vae.abduct(x, parents, t=t_abduct)
    if vae.cond_prior:
        zs = [z["z"] for z in zs]
    if not te and os.environ.get("CGEN_CF_PAIR", "1") != "0" and hasattr(vae, "forward_latents_pair"):
        # the two replays share their latents: two concurrent streams
        (rec_loc, rec_scale), (cf_loc, cf_scale) = vae.forward_latents_pair(zs, parents, cf_parents)
    else:
        rec_loc, rec_scale = vae.forward_latents(zs, parents)
        cf_zs = zs
        if te:
            cf_zs = vae.abduct(x, parents, cf_parents=cf_parents, alpha=alpha, t=t_abduct)
        cf_loc, cf_scale = vae.forward_latents(cf_zs, cf_parents)
    if t_u is not None:
        cf_scale = cf_scale * t_u
    return cf_pixels(x.to(rec_loc.device), rec_loc, rec_scale, cf_loc, cf_scale)


class GraphedCounterfactual:
    """``counterfactual`` captured in a hipGraph per input shape (abduct -> replay x2 -> cf pixels is a fixed sequence of
    ~1200 launches; eager issue is host-bound).  The returned tensor is the graph's static output buffer: it is valid
    until the next call with the same shapes.  Weight updates are picked up (the weight images are refreshed before the
    replay when any parameter version changed); noise comes from the device-side Philox counter, fresh on every replay."""

    def __init__(self, vae, **kw):
        self.vae, self.kw, self.graphs = vae, kw, {}

    @torch.no_grad()
    def __call__(self, x, parents, cf_parents):
        vae = self.vae
        key = (tuple(x.shape), x.dtype, tuple(parents.shape), tuple(cf_parents.shape), compact_parents(parents) is not None,
               compact_parents(cf_parents) is not None)
        ent = self.graphs.get(key)
        if ent is None:
            out = counterfactual(vae, x, parents, cf_parents, **self.kw)  # eager warm-up: sizes the arena, builds tables
            sx, sp, sc = x.clone(), StaticParents(parents), StaticParents(cf_parents)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (a live NCCL watchdog thread must not invalidate the capture: train.CAPTURE_MODE)
                so = counterfactual(vae, sx, sp.t, sc.t, **self.kw)
            self.graphs[key] = (g, sx, sp, sc, so)
            return out
        g, sx, sp, sc, so = ent
        eng = vae.engine()
        eng.stream = torch.cuda.current_stream(eng.device).cuda_stream
        eng.prepare_weights()  # no-op unless a parameter changed since the images were last written
        sx.copy_(x, non_blocking=True)
        sp.load(parents)
        sc.load(cf_parents)
        g.replay()
        return so


class DSCM(nn.Module):
    def __init__(self, args, pgm: nn.Module, predictor: nn.Module, vae: nn.Module):
        super().__init__()
        self.args = args
        self.pgm = pgm
        if pgm is not None:
            self.pgm.requires_grad_(False)
        self.predictor = predictor
        if predictor is not None:
            self.predictor.requires_grad_(False)
        self.vae = vae
        self.lmbda = nn.Parameter(args.lmbda_init * torch.ones(1))
        self.register_buffer("eps", args.elbo_constraint * torch.ones(1))

    def forward(self, obs: Dict[str, Tensor], do: Dict[str, Tensor], elbo_fn=None, cf_particles: int = 1,
                t_abduct: float = 1.0) -> Dict[str, Tensor]:
        """dscm.py:30-95.  With autograd on and a trainable HVAE the image half runs as ONE recorded engine step
        (``HVAE._run_dscm_forward``): factual ELBO, abduction, reconstruction and counterfactual replay share a tape, so
        ``out["loss"].backward()`` reaches the HVAE weights through ``cf_x`` and through the ELBO constraint exactly as the
        reference's autograd graph does (train_cf.py:159-183).  Under ``torch.no_grad()`` (or with a frozen HVAE) the same
        numbers come from the inference calls."""
        pa = {k: v for k, v in obs.items() if k != "x"}
        _pa = vae_preprocess(self.args, {k: v.clone() for k, v in pa.items()})
        cf_pa, cf_list = None, []
        for _ in range(cf_particles):
            cf_pa = self.pgm.counterfactual(obs=pa, intervention=do, num_particles=1)
            cf_list.append(vae_preprocess(self.args, {k: v.clone() for k, v in cf_pa.items()}))
        vae = self.vae
        if torch.is_grad_enabled() and any(p.requires_grad for p in vae.parameters()):
            if not getattr(vae, "_dscm_differentiable", False):
                raise NotImplementedError("DSCM.forward with autograd needs the HVAE image mechanism (its counterfactual passes are "
                                          "recorded on the engine tape); call under torch.no_grad() for inference")
            from .vae import _DSCMFunction

            trig = vae.__dict__.get("_trigger")
            dev = next(vae.parameters()).device
            if trig is None or trig.device != dev:
                trig = vae.__dict__["_trigger"] = torch.zeros(1, device=dev, requires_grad=True)
            elbo, nll, kl, cf_x, var = _DSCMFunction.apply(trig, vae, obs["x"], _pa, tuple(cf_list), self.args.beta, t_abduct)
            vae_out = dict(elbo=elbo, nll=nll, kl=kl)
            var_cf_x = var if cf_particles > 1 else None
            cfs = {"x": cf_x}
        else:
            vae_out = vae(obs["x"], _pa, beta=self.args.beta)
            x = obs["x"].cuda().float()
            sx = torch.zeros_like(x) if cf_particles > 1 else None
            sx2 = torch.zeros_like(x) if cf_particles > 1 else None
            cf_x = None
            with torch.no_grad():
                for _cf_pa in cf_list:
                    zs = vae.abduct(x, parents=_pa, t=t_abduct)
                    if vae.cond_prior:
                        zs = [z["z"] for z in zs]
                    cf_loc, cf_scale = vae.forward_latents(zs, parents=_cf_pa)
                    rec_loc, rec_scale = vae.forward_latents(zs, parents=_pa)
                    cf_x = cf_pixels(x, rec_loc, rec_scale, cf_loc, cf_scale, sx, sx2)
            if cf_particles > 1:
                var_cf_x = (sx2 - sx ** 2 / cf_particles) / cf_particles
                cfs = {"x": sx / cf_particles}
            else:
                var_cf_x = None
                cfs = {"x": cf_x}
        cfs.update(cf_pa)
        nan = sum(int(torch.isnan(v).sum()) for v in list(vae_out.values()) + [cfs["x"]])
        if nan > 0:
            return {"loss": torch.tensor(float("nan"))}
        out = dict(vae_out)
        if elbo_fn is not None and self.predictor is not None:
            aux_loss = elbo_fn.differentiable_loss(self.predictor.model_anticausal, self.predictor.guide_pass, **cfs) / cfs["x"].shape[0]
            with torch.no_grad():
                sg = self.eps - vae_out["elbo"]
            damp = self.args.damping * sg
            out["loss"] = aux_loss - (self.lmbda - damp) * (self.eps - vae_out["elbo"])
            out["aux_loss"] = aux_loss
        out.update({"cfs": cfs, "var_cf_x": var_cf_x})
        return out
