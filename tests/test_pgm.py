"""Parent-SCM counterfactuals without Pyro (causal-gen_amd/pgm.py; reference: src/pgm/flow_pgm.py:47-108, layers.py:33-43,
107-171).  Pyro cannot be imported in the build image, so there are no reference-made vectors: these are the invariants the
mechanisms must satisfy (DESIGN: parity unpinned for this row)."""
from types import SimpleNamespace

import torch

from causal_gen_amd import pgm


def test_linear_spline_is_a_monotone_bijection_with_identity_tails():
    torch.manual_seed(0)
    sp = pgm.LinearSpline(1, count_bins=4)
    x = torch.linspace(-5, 5, 2001).unsqueeze(-1).double()
    sp = sp.double()
    y = sp(x)
    assert torch.all(y[1:] > y[:-1]), "strictly increasing"
    out = (x.abs() >= 3).squeeze(-1)
    assert torch.equal(y[out], x[out]), "identity outside [-bound, bound]"
    assert (sp.inv(y) - x).abs().max() < 1e-9
    assert (sp(sp.inv(x)) - x).abs().max() < 1e-9
    # continuity (also at the knots and at each bin's interior point): around the steepest step of the coarse grid a 200x
    # finer grid has proportionally smaller steps (a linear rational spline may be very steep, never discontinuous)
    i = int((y[1:] - y[:-1]).squeeze(-1).argmax())
    fine = torch.linspace(float(x[i]), float(x[i + 1]), 201).unsqueeze(-1).double()
    yf = sp(fine)
    assert (yf[1:] - yf[:-1]).max() < 0.05 * (y[i + 1] - y[i]).item() + 1e-9


def test_conditional_affine_and_normalisation_round_trip():
    torch.manual_seed(1)
    aff = pgm.ConditionalAffine(pgm.DenseNN(2, [8, 8], [1, 1], torch.nn.LeakyReLU(0.1)))
    ctx, eps = torch.randn(7, 2), torch.randn(7, 1)
    assert (aff.inv(aff(eps, ctx), ctx) - eps).abs().max() < 1e-5
    y = pgm.normalize_fwd(torch.randn(9, 1) * 2)
    assert y.abs().max() < 1 and (pgm.normalize_fwd(pgm.normalize_inv(y)) - y).abs().max() < 1e-6


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)


def test_flow_pgm_counterfactuals():
    m = pgm.FlowPGM(SimpleNamespace(widths=[16, 16]))
    _randomise(m, 2)
    obs = m.sample(6, torch.Generator().manual_seed(3))
    assert set(obs) == set(m.variables)
    # null intervention: abduction -> prediction reproduces the observation
    same = m.counterfactual(obs, {})
    for k in obs:
        assert (same[k] - obs[k]).abs().max() < 1e-4, k
    # do(age): age takes the value; sex / mri_seq (non-descendants) keep theirs; brain and ventricle volume move, with the
    # SAME exogenous noise as observed
    do = {"age": obs["age"] + 0.7}
    cf = m.counterfactual(obs, do)
    assert torch.equal(cf["age"], do["age"]) and torch.equal(cf["sex"], obs["sex"]) and torch.equal(cf["mri_seq"], obs["mri_seq"])
    assert (cf["brain_volume"] - obs["brain_volume"]).abs().max() > 1e-3
    eps = m.infer_exogeneous(obs)
    eps_cf = m.infer_exogeneous(cf)
    for k in ("brain_volume_base", "ventricle_volume_base"):
        assert (eps[k] - eps_cf[k]).abs().max() < 1e-4, k
    # do(ventricle_volume) (a leaf) changes nothing else
    cf2 = m.counterfactual(obs, {"ventricle_volume": torch.zeros(6, 1)})
    for k in ("sex", "mri_seq", "age", "brain_volume"):
        assert (cf2[k] - obs[k]).abs().max() < 1e-4
    # particles average identical deterministic counterfactuals
    cf3 = m.counterfactual(obs, do, num_particles=3)
    assert (cf3["brain_volume"] - cf["brain_volume"]).abs().max() < 1e-5


def test_morphomnist_pgm_counterfactuals_stay_in_range():
    m = pgm.MorphoMNISTPGM(SimpleNamespace(widths=[8, 8]))
    _randomise(m, 4)
    obs = m.sample(5, torch.Generator().manual_seed(5))
    assert obs["digit"].shape == (5, 10) and obs["thickness"].abs().max() < 1
    same = m.counterfactual(obs, {})
    for k in obs:
        assert (same[k] - obs[k]).abs().max() < 1e-4, k
    cf = m.counterfactual(obs, {"thickness": (obs["thickness"] * 0.5)})
    assert torch.equal(cf["digit"], obs["digit"]) and cf["intensity"].abs().max() < 1
    assert (cf["intensity"] - obs["intensity"]).abs().max() > 1e-4


def test_gumbel_max_abduction():
    g = torch.Generator().manual_seed(6)
    logits = torch.log_softmax(torch.randn(64, 5, generator=g), -1)
    k = torch.randint(0, 5, (64, 1), generator=g)
    for _ in range(3):
        # exact posterior (top-down truncated Gumbels): the abducted noise always reproduces the observed class
        eps = pgm.gumbel_max_abduct_exact(k, logits, generator=g)
        assert torch.equal(pgm.gumbel_max_forward(eps, logits), k)
        # the reference's formula (layers.py:139-160), restated as is: the winner's perturbed logit is its fresh Gumbel draw
        # and every other class is truncated below (that draw - logit_k) -- above the winner when logit_k < 0, which is
        # why the reference pins `finding` to its observed value unless it or its parent is intervened on
        # (flow_pgm.py:96-105); the structure is what is checked here
        eps_r = pgm.gumbel_max_abduct(k, logits, generator=g)
        y = eps_r + logits
        win = y.gather(-1, k)
        top = win - logits.gather(-1, k)
        others = y.masked_fill(torch.nn.functional.one_hot(k.squeeze(-1), 5).bool(), float("-inf"))
        assert (others <= top + 1e-6).all() and torch.isfinite(eps_r).all()


def test_dscm_vae_preprocess_accepts_the_pgm_output():
    from causal_gen_amd import dscm

    m = pgm.FlowPGM(SimpleNamespace(widths=[8, 8]))
    obs = m.sample(3, torch.Generator().manual_seed(7))
    cf = m.counterfactual(obs, {"sex": 1 - obs["sex"]})
    pa = dscm.ukbb_preprocess({k: v.clone() for k, v in cf.items()})
    assert set(pa) == set(cf) and all(torch.isfinite(v).all() for v in pa.values())
