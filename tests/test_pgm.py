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


def test_linear_spline_hand_computed_values():
    """Zero raw parameters: 4 equal bins of width 1.5 on [-3, 3], interior derivatives d = 1e-3 + softplus(0) = 0.694147...,
    tails 1, lambda = 0.5.  Middle bin [0, 1.5] (d_k = d_{k+1}): w_b = 1, s = 1, w_c = d, y_c = 0.75, so
    y(theta) = (0 * (.5 - theta) + d * .75 * theta) / ((.5 - theta) + d * theta) for theta <= .5 -- evaluated by hand below.
    First bin [-3, -1.5] (d_0 = 1 at the tail): w_b = sqrt(1 / d), w_c = .5 + .5 w_b d, y_c = (-1.5 - .75 w_b) / (.5 + .5 w_b) ."""
    import math

    sp = pgm.LinearSpline(1, count_bins=4).double()
    with torch.no_grad():
        for p in sp.parameters():
            p.zero_()
    d = 1e-3 + math.log(2.0)
    # middle bin, theta = 0.2 (x = 0.3) and theta = 0.8 (x = 1.2)
    th = 0.2
    y_l = (d * 0.75 * th) / ((0.5 - th) + d * th)
    th = 0.8
    y_r = (d * 0.75 * (1 - th) + 1.5 * (th - 0.5)) / (d * (1 - th) + (th - 0.5))
    # first bin, theta = 0.4 (x = -2.4)
    wb = math.sqrt(1.0 / d)
    wc = 0.5 * 1.0 + 0.5 * wb * d
    yc = (0.5 * -3.0 + 0.5 * wb * -1.5) / (0.5 + 0.5 * wb)
    th = 0.4
    y_f = (-3.0 * (0.5 - th) + wc * yc * th) / ((0.5 - th) + wc * th)
    x = torch.tensor([[0.3], [1.2], [-2.4], [0.0], [1.5], [4.0]], dtype=torch.float64)
    y = sp(x).squeeze(-1).tolist()
    for got, want in zip(y, [y_l, y_r, y_f, 0.0, 1.5, 4.0]):
        assert abs(got - want) < 1e-12, (got, want)
    assert (sp.inv(sp(x)) - x).abs().max() < 1e-12


def test_conditional_affine_hand_computed_value():
    """layers.py:33-43 with a one-layer-deep net made explicit: loc = 0.3 + 0.5 c, log_scale = -0.2 + 0.25 c."""
    import math

    nn_ = pgm.DenseNN(1, [1], [1, 1], torch.nn.Identity())
    with torch.no_grad():
        nn_.layers[0].weight.fill_(1.0); nn_.layers[0].bias.zero_()
        nn_.layers[1].weight.copy_(torch.tensor([[0.5], [0.25]])); nn_.layers[1].bias.copy_(torch.tensor([0.3, -0.2]))
    aff = pgm.ConditionalAffine(nn_)
    c, eps = torch.tensor([[2.0]]), torch.tensor([[-1.5]])
    want = (0.3 + 0.5 * 2.0) + math.exp(-0.2 + 0.25 * 2.0) * -1.5
    assert abs(aff(eps, c).item() - want) < 1e-6
    assert abs(aff.inv(torch.tensor([[want]]), c).item() - -1.5) < 1e-6
    # the [-1, 1] normalisation: 2 sigmoid(0.7) - 1 = tanh(0.35)
    assert abs(pgm.normalize_fwd(torch.tensor(0.7)).item() - math.tanh(0.35)) < 1e-7


def test_colour_mnist_pgm_counterfactuals():
    """flow_pgm.py:451-530: two categorical roots, no exogenous noise: intervened variables take the value, the rest keep theirs."""
    m = pgm.ColourMNISTPGM(SimpleNamespace())
    assert sorted(k for k, _ in m.named_parameters()) == ["colour_logits", "digit_logits"]
    obs = m.sample(7, torch.Generator().manual_seed(8))
    assert obs["digit"].shape == (7, 10) and obs["colour"].shape == (7, 10)
    assert torch.equal(obs["digit"].sum(-1), torch.ones(7)) and torch.equal(obs["colour"].sum(-1), torch.ones(7))
    same = m.counterfactual(obs, {})
    assert all(torch.equal(same[k], obs[k]) for k in obs)
    do = {"colour": obs["colour"].roll(1, 0)}
    cf = m.counterfactual(obs, do, num_particles=2)
    assert torch.equal(cf["colour"], do["colour"]) and torch.equal(cf["digit"], obs["digit"])
    assert m.infer_exogeneous(obs) == {}


def test_chest_pgm_counterfactuals():
    """flow_pgm.py:533-710: age spline (8 bins), finding | age by Gumbel-max, race / sex roots."""
    m = pgm.ChestPGM(SimpleNamespace())
    keys = sorted(m.state_dict())
    assert "age_flow_components.0.unnormalized_widths" in keys and "finding_transform_GumbelMax.context_nn.layers.2.weight" in keys
    assert m.age_flow_components[0].unnormalized_widths.shape == (1, 8) and m.race_logits.shape == (1, 3)
    _randomise(m, 9)
    obs = m.sample(16, torch.Generator().manual_seed(10))
    assert set(obs) == set(m.variables) and obs["race"].shape == (16, 3) and obs["finding"].shape == (16, 1)
    m.generator = torch.Generator().manual_seed(11)
    # null intervention: everything is reproduced (finding through the reference's pin, flow_pgm.py:96-105)
    same = m.counterfactual(obs, {})
    for k in obs:
        assert (same[k] - obs[k]).abs().max() < 1e-4, k
    # do(sex) / do(race): roots without descendants: nothing else moves
    cf = m.counterfactual(obs, {"sex": 1 - obs["sex"], "race": obs["race"].roll(1, 1)})
    assert torch.equal(cf["sex"], 1 - obs["sex"]) and (cf["age"] - obs["age"]).abs().max() < 1e-4 and torch.equal(cf["finding"], obs["finding"])
    # do(age): finding is re-predicted from the abducted Gumbel noise under the new logits, values stay binary
    cf = m.counterfactual(obs, {"age": obs["age"] + 1.0})
    assert torch.equal(cf["age"], obs["age"] + 1.0) and set(cf["finding"].flatten().tolist()) <= {0.0, 1.0}
    # ... and with the exact posterior an intervention that leaves the logits unchanged reproduces the observed class
    m.exact_gumbel_posterior = True
    cf = m.counterfactual(obs, {"age": obs["age"].clone()})
    assert torch.equal(cf["finding"], obs["finding"])
    # do(finding) overrides the mechanism
    cf = m.counterfactual(obs, {"finding": 1 - obs["finding"]})
    assert torch.equal(cf["finding"], 1 - obs["finding"]) and (cf["age"] - obs["age"]).abs().max() < 1e-4
    eps = m.infer_exogeneous(obs)
    assert eps["age_base"].shape == (16, 1) and eps["finding_base"].shape == (16, 2)


def test_reference_pgm_checkpoints_load_without_their_predictors():
    """The reference keeps its anticausal predictors (encoder_*) on the PGM module; load_reference_state_dict drops exactly those."""
    m = pgm.MorphoMNISTPGM(SimpleNamespace(widths=[8, 8]))
    sd = {k: v.clone() + 1 for k, v in m.state_dict().items()}
    sd["encoder_t.cnn.0.weight"] = torch.zeros(3)
    sd["encoder_y.fc.2.bias"] = torch.zeros(10)
    dropped = m.load_reference_state_dict(sd)
    assert dropped == ["encoder_t.cnn.0.weight", "encoder_y.fc.2.bias"]
    assert torch.equal(m.digit_logits, sd["digit_logits"])
    import pytest

    with pytest.raises(RuntimeError):
        m.load_reference_state_dict({"nonsense": torch.zeros(1)})
