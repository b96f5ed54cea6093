"""DSCM.forward on the HIP path against oracle/dscm_ref.py (a differentiable torch-CPU restatement of dscm.py:40-95 over
the oracle HVAE): particle mean / variance of the counterfactual, ELBO, Lagrangian loss -- and d loss / d theta through the
counterfactual branch (train_cf.py:159-183 fine-tunes the HVAE through abduct -> two replays -> dscm.py:55-56), which the
reference gets from autograd and this build from one reverse sweep over a tape shared by all passes."""
from types import SimpleNamespace

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


class StubPGM(torch.nn.Module):
    """pgm.counterfactual is pyro-side (outside the hot path): returns the observed parents with `do` applied."""

    def counterfactual(self, obs, intervention, num_particles=1):
        return {k: intervention.get(k, v) for k, v in obs.items()}


class StubPredictor(torch.nn.Module):
    model_anticausal = None
    guide_pass = None


class StubELBO:
    """Stands in for TraceStorage_ELBO.differentiable_loss(model, guide, **cfs): a fixed smooth functional of cf_x, summed
    over the batch (DSCM.forward divides by B)."""

    def __init__(self, w):
        self.w = w

    def differentiable_loss(self, model, guide, **cfs):
        x = cfs["x"]
        return ((x * self.w.to(x.device)).sum() + 0.5 * (x ** 2).sum())


def _build(name, dtype="f32"):
    from causal_gen_amd import vae
    from causal_gen_amd.hps import Hparams

    fx = load_golden(name)
    hpd = dict(fx["hp"])
    m = vae.HVAE(Hparams(**hpd))
    m.load_state_dict(fx["state_dict"])
    m.compute_dtype = dtype
    return fx, hpd, m.cuda().eval()


@pytest.mark.parametrize("name,particles", [("tiny_default_c1.pt", 3), ("tiny_light_c1.pt", 1), ("tiny_default_c3.pt", 2)])
def test_dscm_forward_matches_oracle_values_and_gradients(name, particles):
    fx, hpd, m = _build(name)
    _dscm_case(fx, hpd, m, name, particles)


def test_dscm_forward_at_morphomnist_size():
    """The same comparison at a BASELINE preset's size: the morphomnist HVAE (32x32, 20 + 20 blocks, 12 parents; exogenous prior,
    as DSCM.forward requires -- dscm.py:52-54 replays plain z lists) with perturbed weights, two particles, values and every
    parameter gradient against oracle/dscm_ref.py run live on the box's CPU."""
    import bench
    from causal_gen_amd import vae
    from causal_gen_amd.hps import setup_hparams
    from oracle import fullsize_recipe as R

    hp = setup_hparams("morphomnist", cond_prior=False)
    torch.manual_seed(7)
    m = vae.HVAE(hp)
    m.apply(R.init_bias)
    R.perturb(m)
    x, pa = R.inputs(hp, 2)
    fx = {"x": x, "pa": pa, "cf_pa": pa.roll(1, 0) * 0.5, "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()}}
    m.compute_dtype = "f32"
    _dscm_case(fx, dict(vars(hp)), m.cuda().eval(), "morphomnist", 2)


def _dscm_case(fx, hpd, m, name, particles):
    from causal_gen_amd import dscm
    from oracle import dscm_ref, hvae_ref

    hp = SimpleNamespace(**hpd)
    x, pa, cf = fx["x"], fx["pa"], fx["cf_pa"]
    B, ctx = x.shape[0], pa.shape[1]
    names = [f"p{i}" for i in range(ctx)]
    beta, t_ab, lmbda0, eps_c, damping = 1.7, 0.9, 0.8, 2.0, 10.0
    g = torch.Generator().manual_seed(5)
    w = torch.randn(x.shape, generator=g) * 0.3

    # ---- oracle: one differentiable graph, noise drawn once and recorded
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
    lm = torch.tensor([lmbda0], requires_grad=True)
    noise = hvae_ref._Noise(None)
    torch.manual_seed(17)
    # particle p intervenes on parent 0 with a different value each time (the stub PGM is deterministic: emulate by lists)
    cf_list = [torch.cat([cf[:, :1] * (1.0 + 0.5 * p), pa[:, 1:]], 1) for p in range(particles)]
    ref = dscm_ref.dscm_forward(sd, hp, x, pa, cf_list, beta, t_abduct=t_ab, noise=noise,
                                aux_fn=lambda c: ((c * w).sum() + 0.5 * (c ** 2).sum()) / B, lmbda=lm, eps=torch.tensor([eps_c]),
                                damping=damping)
    ref["loss"].sum().backward()

    # ---- HIP path through the reference's DSCM surface
    class SeqPGM(StubPGM):
        def __init__(self):
            super().__init__()
            self.i = 0

        def counterfactual(self, obs, intervention, num_particles=1):
            out = dict(obs)
            out["p0"] = cf[:, 0, 0, 0].cuda() * (1.0 + 0.5 * self.i)
            self.i += 1
            return out

    args = SimpleNamespace(**{**hpd, "parents_x": names, "dataset": "none", "lmbda_init": lmbda0, "elbo_constraint": eps_c, "damping": damping})
    args.beta = beta
    model = dscm.DSCM(args, SeqPGM(), StubPredictor(), m).cuda()
    for p in m.parameters():
        p.requires_grad_(True)
    obs = {"x": x.cuda()}
    obs.update({n: pa[:, i, 0, 0].cuda() for i, n in enumerate(names)})
    m.noise = [e.clone() for e in noise.drawn]
    out = model(obs, {"p0": None}, StubELBO(w), cf_particles=particles, t_abduct=t_ab)
    assert not m.noise, "every recorded draw must have been consumed, in the reference's order"
    for k in ("elbo", "nll", "kl"):
        assert abs(float(out[k]) - float(ref[k])) <= 1e-4 * abs(float(ref[k])), (k, float(out[k]), float(ref[k]))
    ok = torch.ones_like(x, dtype=torch.bool)
    assert (out["cfs"]["x"].detach().cpu() - ref["cf_x"].detach())[ok].abs().max() < 1e-3
    if particles > 1:
        assert (out["var_cf_x"].cpu() - ref["var_cf_x"]).abs().max() < 1e-3
    else:
        assert out["var_cf_x"] is None
    assert abs(float(out["aux_loss"]) - float(ref["aux_loss"])) <= 2e-4 * abs(float(ref["aux_loss"])) + 1e-5
    assert abs(float(out["loss"]) - float(ref["loss"])) <= 2e-4 * abs(float(ref["loss"])) + 1e-5
    out["loss"].sum().backward()
    torch.cuda.synchronize()
    assert abs(float(model.lmbda.grad) - float(lm.grad)) <= 1e-4 * abs(float(lm.grad)) + 1e-6
    worst, n_checked = 0.0, 0
    for n_, p in m.named_parameters():
        rg = sd[n_].grad
        if rg is None or float(rg.abs().max()) == 0.0:
            continue
        assert p.grad is not None, n_
        d = float((p.grad.cpu() - rg).abs().max()) / float(rg.abs().max())
        l2 = float((p.grad.cpu() - rg).norm()) / float(rg.norm())
        worst = max(worst, d)
        n_checked += 1
        # max-norm 5e-3 (the counterfactual step clamps to [-1, 1] and the RGB decode clamps three times per pixel: a value
        # within f32 rounding of a clamp edge takes the other branch of the subgradient), relative L2 2e-3
        assert d < 5e-3 and l2 < 2e-3, (n_, d, l2)
    assert n_checked > 20
    print(name, "particles", particles, "worst grad err rel-to-max", worst, "over", n_checked, "tensors")


def test_cf_branch_alone_reaches_the_weights():
    """loss = aux(cf_x) only (no ELBO term): the gradient that arrives in the encoder / posterior / decoder weights is the
    one that flowed through abduction and both replays."""
    from causal_gen_amd import vae as hvae_mod
    from oracle import dscm_ref, hvae_ref

    fx, hpd, m = _build("tiny_default_c1.pt")
    hp = SimpleNamespace(**hpd)
    x, pa, cf = fx["x"], fx["pa"], fx["cf_pa"]
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
    noise = hvae_ref._Noise(None)
    torch.manual_seed(3)
    ref = dscm_ref.dscm_forward(sd, hp, x, pa, [cf], 1.0, noise=noise)
    (ref["cf_x"] ** 2).sum().backward()
    m.noise = [e.clone() for e in noise.drawn]
    trig = torch.zeros(1, device="cuda", requires_grad=True)
    elbo, nll, kl, cf_x, _ = hvae_mod._DSCMFunction.apply(trig, m, x.cuda(), pa.cuda(), (cf.cuda(),), 1.0, 1.0)
    (cf_x ** 2).sum().backward()
    torch.cuda.synchronize()
    for n_ in ("encoder.stem.weight", "decoder.blocks.0.posterior.conv.1.weight", "decoder.blocks.1.z_proj.weight",
               "likelihood.x_logscale.weight"):
        rg = sd[n_].grad
        got = dict(m.named_parameters())[n_].grad.cpu()
        assert float((got - rg).abs().max()) < 2e-3 * float(rg.abs().max()), n_


def test_backward_after_a_later_pass_fails_loudly():
    """A forward pass whose tape was recycled by a later inference call must not return silently empty gradients."""
    fx, hpd, m = _build("tiny_light_c1.pt")
    for p in m.parameters():
        p.requires_grad_(True)
    out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=1.0)
    m.abduct(fx["x"].cuda(), fx["pa"].cuda())
    with pytest.raises(RuntimeError, match="recycled"):
        out["elbo"].backward()


def test_dscm_forward_without_grad_keeps_the_inference_path():
    from causal_gen_amd import dscm

    fx, hpd, m = _build("tiny_default_c1.pt")
    x, pa, cf = fx["x"].cuda(), fx["pa"].cuda(), fx["cf_pa"].cuda()
    args = SimpleNamespace(**hpd, parents_x=["a", "b", "c"], dataset="none", lmbda_init=1.0, elbo_constraint=2.0, damping=10.0)
    model = dscm.DSCM(args, StubPGM(), None, m)
    obs = {"x": x, "a": pa[:, 0, 0, 0], "b": pa[:, 1, 0, 0], "c": pa[:, 2:3, 0, 0]}
    with torch.no_grad():
        out = model(obs, {"a": cf[:, 0, 0, 0]}, None, cf_particles=2)
    assert out["cfs"]["x"].shape == x.shape and torch.isfinite(out["cfs"]["x"]).all()


def test_dscm_forward_with_the_pyro_free_parent_scm():
    """DSCM.forward end to end on the GPU box with causal-gen_amd/pgm.FlowPGM as `pgm` (no pyro): a null intervention gives the
    observation back; do(sex) changes brain volume, hence the image; the loss is differentiable into the HVAE."""
    from causal_gen_amd import dscm, pgm, vae
    from causal_gen_amd.hps import Hparams
    from oracle import hparams as ohp

    hp = vars(ohp.tiny_hparams(hps="tiny_ukbb", z_max_res=8, context_dim=4))
    torch.manual_seed(0)
    m = vae.HVAE(Hparams(**hp))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.1 if p.dim() < 4 else 0.5 / (p[0].numel() ** 0.5)))
    m = m.cuda().eval()
    scm = pgm.FlowPGM(SimpleNamespace(widths=[8, 8]))
    with torch.no_grad():
        for p in scm.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    scm = scm.cuda()
    obs = scm.sample(3, torch.Generator().manual_seed(2))
    x = ((torch.randint(0, 256, (3, 1, 16, 16), generator=g).float() - 127.5) / 127.5).cuda()
    args = SimpleNamespace(**hp, parents_x=["mri_seq", "brain_volume", "ventricle_volume", "sex"], dataset="ukbb", lmbda_init=1.0,
                           elbo_constraint=2.0, damping=10.0)
    model = dscm.DSCM(args, scm, StubPredictor(), m).cuda()
    o = dict(obs, x=x)
    with torch.no_grad():
        same = model(o, {}, None)
        flip = model(o, {"sex": 1 - obs["sex"]}, None)
    assert (same["cfs"]["x"] - x).abs().max() < 1e-4
    assert (flip["cfs"]["x"] - x).abs().mean() > 1e-4
    assert torch.equal(flip["cfs"]["sex"], 1 - obs["sex"]) and (flip["cfs"]["brain_volume"] - obs["brain_volume"]).abs().max() > 1e-4
    for p in m.parameters():
        p.requires_grad_(True)
    out = model(o, {"sex": 1 - obs["sex"]}, StubELBO(torch.ones(3, 1, 16, 16)), cf_particles=2)
    out["loss"].sum().backward()
    torch.cuda.synchronize()
    assert m.encoder.stem.weight.grad is not None and float(m.encoder.stem.weight.grad.abs().max()) > 0


@pytest.mark.parametrize("tag", ["default_p3", "ukbb_light_p1"])
def test_dscm_forward_against_the_references_own_dscm_forward(tag):
    """The product's ``DSCM.forward`` against tests/golden/dscm_*.pt: outputs of the REFERENCE's ``DSCM`` class itself (src/pgm/dscm.py:15-95;
    oracle/make_dscm_golden.py runs it with pyro / torchvision / imageio / seaborn stubbed and the stand-in pgm / predictor / ELBO of
    oracle/dscm_stubs.py).  Same obs / do dicts through the same surface: ELBO, NLL, KL, counterfactual particle mean and variance,
    auxiliary loss, damped Lagrangian, and the gradients of the loss w.r.t. the HVAE weights and the multiplier."""
    from causal_gen_amd import dscm, vae
    from causal_gen_amd.hps import Hparams
    from oracle import dscm_stubs as S

    fx = load_golden("dscm_%s.pt" % tag)
    hpd, c = dict(fx["hp"]), fx["constants"]
    m = vae.HVAE(Hparams(**hpd))
    m.load_state_dict(fx["state_dict"])
    m.compute_dtype = "f32"
    m = m.cuda().eval()
    for p in m.parameters():
        p.requires_grad_(True)
    args = SimpleNamespace(**{**hpd, "parents_x": fx["parents_x"], "dataset": fx["dataset"], "lmbda_init": c["lmbda_init"],
                              "elbo_constraint": c["elbo_constraint"], "damping": c["damping"]})
    args.beta = c["beta"]
    model = dscm.DSCM(args, S.StubPGM(), S.StubPredictor(), m).cuda()
    obs = {k: v.cuda() for k, v in fx["obs"].items()}
    do = {k: v.cuda() for k, v in fx["do"].items()}
    pre = dscm.vae_preprocess(args, {k: v.clone() for k, v in obs.items() if k != "x"})
    # (the UKBB log-standardisation runs where the parents live -- here the GPU: one f32 ulp of log(1.6e6) ~ 14, divided by sigma ~ 0.1)
    assert torch.allclose(pre[:, :, 0, 0].cpu(), fx["vae_parents"], rtol=0, atol=3e-5)
    m.noise = [e.clone() for e in fx["eps"]]
    out = model(obs, do, S.StubELBO(fx["w"]), cf_particles=fx["particles"], t_abduct=fx["t_abduct"])
    assert not m.noise, "every draw of the reference must be consumed, in its order"
    for k in ("elbo", "nll", "kl"):
        assert abs(float(out[k].detach()) - float(fx["out"][k])) <= 1e-4 * abs(float(fx["out"][k])), (k, float(out[k].detach()), float(fx["out"][k]))
    for k in ("aux_loss", "loss"):
        assert abs(float(out[k].detach()) - float(fx["out"][k])) <= 2e-4 * abs(float(fx["out"][k])) + 1e-5, (k, float(out[k].detach()), float(fx["out"][k]))
    assert float((out["cfs"]["x"].detach().cpu() - fx["cf_x"]).abs().max()) < 1e-3
    for k, v in fx["cf_parents"].items():
        assert torch.equal(out["cfs"][k].cpu(), v)
    if fx["particles"] > 1:
        assert float((out["var_cf_x"].cpu() - fx["var_cf_x"]).abs().max()) < 1e-3
    else:
        assert out["var_cf_x"] is None
    out["loss"].sum().backward()
    torch.cuda.synchronize()
    assert abs(float(model.lmbda.grad) - float(fx["lmbda_grad"])) <= 1e-4 * abs(float(fx["lmbda_grad"])) + 1e-6
    named, n = dict(m.named_parameters()), 0
    for name, g in fx["grads"].items():
        if float(g.abs().max()) == 0.0:
            continue
        got = named[name].grad
        assert got is not None, name
        d = float((got.cpu() - g).abs().max()) / float(g.abs().max())
        l2 = float((got.cpu() - g).norm()) / float(g.norm())
        assert d < 5e-3 and l2 < 2e-3, (name, d, l2)
        n += 1
    assert n > 50
