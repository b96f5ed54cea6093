"""Duck-typed stand-ins for the Pyro side of ``DSCM.forward`` (src/pgm/dscm.py:30-95) -- ORACLE / test infrastructure.

The reference's ``DSCM`` takes a parent SCM (``pgm.counterfactual``), the anticausal predictors and a Pyro ELBO; all three
are outside the hot path and need pyro.  These stubs are what ``oracle/make_dscm_golden.py`` hands the REFERENCE's own
``DSCM`` class when it generates ``tests/golden/dscm_*.pt``, and what the tests hand the oracle restatement and the product,
so that all three see the same parents and the same auxiliary loss."""
import torch


class StubPGM(torch.nn.Module):
    """``counterfactual(obs, intervention, num_particles)``: the observed parents with the intervened ones replaced; call i
    of an instance scales the intervention by (1 + 0.25 i), so that the particles of one ``DSCM.forward`` differ."""

    def __init__(self):
        super().__init__()
        self.calls = 0

    def counterfactual(self, obs, intervention, num_particles=1):
        out = {k: v.clone() for k, v in obs.items()}
        for k, v in intervention.items():
            out[k] = v.to(out[k].device) * (1.0 + 0.25 * self.calls)
        self.calls += 1
        return out


class StubPredictor(torch.nn.Module):
    model_anticausal = None
    guide_pass = None


class StubELBO:
    """``differentiable_loss(model, guide, **cfs)``: a fixed smooth functional of the counterfactual image and parents, summed
    over the batch (``DSCM.forward`` divides by B, dscm.py:78-83)."""

    def __init__(self, w):
        self.w = w

    def differentiable_loss(self, model, guide, **cfs):
        x = cfs["x"]
        val = (x * self.w.to(x.device)).sum() + 0.5 * (x ** 2).sum()
        for k in sorted(cfs):
            if k != "x":
                val = val + 0.1 * cfs[k].float().to(x.device).sum()
        return val


CASES = {
    # tag: (tiny-hparams overrides, dataset, parents_x, intervened parents, cf_particles, t_abduct)
    "default_p3": (dict(hps="tiny"), "none", ["p0", "p1", "p2"], ["p1"], 3, 0.9),
    "ukbb_light_p1": (dict(hps="tiny_ukbb", z_max_res=8, context_dim=4), "ukbb192", ["mri_seq", "brain_volume", "ventricle_volume", "sex"],
                      ["brain_volume"], 1, 1.0),
}
CONSTANTS = dict(beta=1.7, lmbda_init=0.8, elbo_constraint=2.0, damping=10.0)


def make_obs(tag, hp, B, gen):
    """Inputs of a case: pixels on the 256 u8 levels, parents as the PGM holds them ([B] or [B,1]; the UKBB ones in [-1, 1])."""
    _, dataset, parents_x, do_keys, _, _ = CASES[tag]
    R, C = hp.input_res, hp.input_channels
    obs = {"x": (torch.randint(0, 256, (B, C, R, R), generator=gen).float() - 127.5) / 127.5}
    for k in parents_x:
        if k in ("mri_seq", "sex"):
            obs[k] = torch.randint(0, 2, (B, 1), generator=gen).float()
        elif "ukbb" in dataset:
            obs[k] = torch.rand(B, 1, generator=gen) * 1.6 - 0.8
        else:
            obs[k] = torch.randn(B, generator=gen)
    do = {}
    for k in do_keys:
        do[k] = (torch.rand(obs[k].shape, generator=gen) * 1.2 - 0.6) if "ukbb" in dataset else torch.randn(obs[k].shape, generator=gen)
    return obs, do
