"""The oracle (oracle/*.py) against the fixtures generated from the imported reference
(oracle/make_golden.py -> tests/golden).  CPU only."""
import glob
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import dmol_ref, dscm_ref, hvae_ref, train_ref
from oracle import hparams as ohp

TINY = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "tiny_*.pt")))
TOL = dict(rtol=2e-5, atol=2e-6)


def _sd(fx, grad=False):
    return {k: v.clone().requires_grad_(grad and v.is_floating_point()) for k, v in fx["state_dict"].items()}


@pytest.mark.parametrize("name", TINY)
def test_forward_and_grads(name):
    fx = load_golden(name)
    hp = SimpleNamespace(**fx["hp"])
    sd = _sd(fx, grad=True)
    f = fx["fwd"]
    out = hvae_ref.hvae_forward(sd, hp, fx["x"], fx["pa"], beta=f["beta"], noise=f["eps"], want_stats=True)
    for k in ("elbo", "nll", "kl"):
        torch.testing.assert_close(out[k].detach(), f[k], **TOL)
    torch.testing.assert_close(out["_h"].detach(), f["h"], **TOL)
    assert len(out["_kl_maps"]) == len(f["kl_maps"])
    for a, b in zip(out["_kl_maps"], f["kl_maps"]):
        torch.testing.assert_close(a.detach(), b, **TOL)
    out["elbo"].backward()
    for n, g in f["grads"].items():
        torch.testing.assert_close(sd[n].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m: f"{n}: {m}")


@pytest.mark.parametrize("name", TINY)
def test_variants_and_cf(name):
    fx = load_golden(name)
    hp = SimpleNamespace(**fx["hp"])
    sd = _sd(fx)
    x, pa, cf_pa = fx["x"], fx["pa"], fx["cf_pa"]
    with torch.no_grad():
        if "fwd_drop" in fx:
            d = fx["fwd_drop"]
            o = hvae_ref.hvae_forward(sd, hp, x, pa, beta=1.0, noise=d["eps"], drop=d["drop"])
            for k in ("elbo", "nll", "kl"):
                torch.testing.assert_close(o[k], d[k], **TOL)
        d = fx["fwd_freebits"]
        hp_fb = SimpleNamespace(**{**fx["hp"], "kl_free_bits": d["free_bits"]})
        o = hvae_ref.hvae_forward(sd, hp_fb, x, pa, beta=1.0, noise=d["eps"])
        for k in ("elbo", "nll", "kl"):
            torch.testing.assert_close(o[k], d[k], **TOL)
        # abduct -> forward_latents -> cf pixels (dscm.py:52-56)
        ab = fx["abduct"]
        o = dscm_ref.counterfactual(sd, hp, x, pa, cf_pa, t_abduct=ab["t"], noise=ab["eps"])
        for a, b in zip(o["zs"], ab["zs"]):
            torch.testing.assert_close(a, b, **TOL)
        for k in ("rec_loc", "rec_scale", "cf_loc", "cf_scale", "cf_x"):
            torch.testing.assert_close(o[k], fx["cf"][k], rtol=1e-4, atol=1e-5)
        # partial latents
        pl = fx["partial_latents"]
        loc, sc = hvae_ref.hvae_forward_latents(sd, hp, ab["zs"][: pl["n"]], cf_pa, t=pl["t"], noise=pl["eps"])
        torch.testing.assert_close(loc, pl["loc"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sc, pl["scale"], rtol=1e-4, atol=1e-5)
        if "mediator" in fx:
            md = fx["mediator"]
            zstar = hvae_ref.hvae_abduct(sd, hp, x, pa, cf_parents=cf_pa, alpha=md["alpha"], t=md["t"], noise=md["eps"])
            for a, b in zip(zstar, md["zstar"]):
                torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        sm = fx["sample"]
        loc, sc = hvae_ref.hvae_sample(sd, hp, pa, t=sm["t"], noise=sm["eps"])
        torch.testing.assert_close(loc, sm["loc"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sc, sm["scale"], rtol=1e-4, atol=1e-5)


def test_op_vectors():
    fx = load_golden("ops.pt")
    g = fx["gaussian_kl"]
    torch.testing.assert_close(hvae_ref.gaussian_kl(g["q_loc"], g["q_logscale"], g["p_loc"], g["p_logscale"]), g["kl"],
                               rtol=1e-6, atol=1e-6)
    for C in (1, 3):
        d = fx[f"dgauss_c{C}"]
        hp = ohp.tiny_hparams(input_channels=C)
        sd = {"likelihood." + k: v for k, v in d["state_dict"].items()}
        h = d["h"].clone().requires_grad_(True)
        nll = hvae_ref.dgauss_nll(sd, hp, h, d["x"])
        torch.testing.assert_close(nll.detach(), d["nll"], **TOL)
        (gh,) = torch.autograd.grad(nll.sum(), h)
        torch.testing.assert_close(gh, d["grad_h"], rtol=1e-4, atol=1e-6)
        loc, ls = hvae_ref.dgauss_params(sd, hp, d["h"], d["x"])
        torch.testing.assert_close(loc, d["loc"], **TOL)
        torch.testing.assert_close(ls, d["logscale"], **TOL)
        sx, ss = hvae_ref.dgauss_sample(sd, hp, d["h"])
        torch.testing.assert_close(sx, d["sample_x"], **TOL)
        torch.testing.assert_close(ss, d["sample_scale"], **TOL)
    d = fx["dmol"]
    l = d["l"].clone().requires_grad_(True)
    loss = dmol_ref.dmol_nll(d["x"], l)
    torch.testing.assert_close(loss.detach(), d["loss"], **TOL)
    (gl,) = torch.autograd.grad(loss.sum(), l)
    torch.testing.assert_close(gl, d["grad_l"], rtol=1e-4, atol=1e-7)
    for mask in ("soft", "hard", "top3"):
        mx, ms = dmol_ref.dmol_mean(d["l"], mask)
        torch.testing.assert_close(mx, d[f"mean_{mask}"], **TOL)
        torch.testing.assert_close(ms, d[f"scale_{mask}"], **TOL)


@pytest.mark.parametrize("name", ["morphomnist", "cmnist"])
def test_anchor_init_and_forward(name):
    """Full-size presets: the oracle's init consumes the RNG in the reference's order (sum|theta| equal)
    and reproduces (elbo, nll, kl) at the seeded input."""
    row = load_golden("anchors.pt")[name]
    hp = ohp.make_hparams(name)
    torch.manual_seed(7)
    sd = hvae_ref.init_state_dict(hp)
    assert list(sd.keys()) == row["keys"]
    assert [tuple(v.shape) for v in sd.values()] == row["shapes"]
    assert hvae_ref.count_params(sd) == row["n_params"]
    assert abs(float(sum(v.abs().double().sum() for v in sd.values())) - row["abs_sum"]) < 1e-6 * row["abs_sum"]
    g = torch.Generator().manual_seed(123)
    R, C = hp.input_res, hp.input_channels
    x = (torch.randint(0, 256, (2, C, R, R), generator=g).float() - 127.5) / 127.5
    pa = torch.randn(2, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, R, R)
    torch.manual_seed(11)
    with torch.no_grad():
        o = hvae_ref.hvae_forward(sd, hp, x, pa, beta=hp.beta)
    for k in ("elbo", "nll", "kl"):
        assert abs(float(o[k]) - row[k]) <= 2e-5 * abs(row[k]) + 1e-7, (k, float(o[k]), row[k])


def test_anchor_ukbb192_init():
    row = load_golden("anchors.pt")["ukbb192"]
    hp = ohp.make_hparams("ukbb192")
    torch.manual_seed(7)
    sd = hvae_ref.init_state_dict(hp)
    assert list(sd.keys()) == row["keys"]
    assert hvae_ref.count_params(sd) == row["n_params"] == 17371122
    assert abs(float(sum(v.abs().double().sum() for v in sd.values())) - row["abs_sum"]) < 1e-6 * row["abs_sum"]


def test_train_steps_adamw_ema():
    fx = load_golden("train_steps.pt")
    hp = SimpleNamespace(**fx["hp"])
    tr = train_ref.RefTrainer(fx["p0"], hp)
    n = max(fx["snaps"])
    for s in range(n):
        gn = tr.apply_grads({k: fx["grads"][k][s] for k in fx["p0"]})
        assert abs(gn - fx["norms"][s]) <= 1e-5 * fx["norms"][s]
        if s + 1 in fx["snaps"]:
            snap = fx["snaps"][s + 1]
            for k in fx["p0"]:
                torch.testing.assert_close(tr.sd[k].detach(), snap["params"][k], rtol=1e-5, atol=1e-7)
                torch.testing.assert_close(tr.ema[k], snap["ema"][k], rtol=1e-5, atol=1e-7)
            assert abs(tr.lr() - snap["lr"]) < 1e-12
    assert tr.skipped == 1


@pytest.mark.parametrize("name,n_params", [("simple_vae_c1.pt", 234690), ("simple_vae_c1x.pt", 208274), ("simple_vae_c3.pt", 236358),
                                           ("simple_vae_dmol3.pt", 237956), ("simple_vae_gauss1.pt", 234690)])
def test_simple_vae_config1(name, n_params):
    """Config 1: the oracle's restatement of simple_vae.py against the reference's own outputs (conditional-prior
    preset; the same with the exogenous prior; RGB input)."""
    from oracle import simple_ref

    fx = load_golden(name)
    assert fx["n_params"] == n_params
    hp = SimpleNamespace(**fx["hp"])
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx["state_dict"].items()}
    x, pa, cf_pa, eps = fx["x"], fx["pa"], fx["cf_pa"], fx["eps"]
    f = fx["fwd"]
    u = fx.get("u")  # pinned dequantisation noise (GaussNet only)
    out = simple_ref.forward(sd, hp, x, pa, beta=f["beta"], eps=eps, u=u)
    for k in ("elbo", "nll", "kl"):
        torch.testing.assert_close(out[k].detach(), f[k], **TOL)
    out["elbo"].backward()
    for n, g in f["grads"].items():
        torch.testing.assert_close(sd[n].grad, g, rtol=1e-4, atol=1e-6, msg=lambda m: f"{n}: {m}")
    with torch.no_grad():
        if "fwd_drop" in fx:
            d = fx["fwd_drop"]
            o = simple_ref.forward(sd, hp, x, pa, beta=1.0, eps=eps, drop=d["drop"], u=u)
            for k in ("elbo", "nll", "kl"):
                torch.testing.assert_close(o[k], d[k], **TOL)
        ab = fx["abduct"]
        q = simple_ref.abduct(sd, hp, x, pa, t=ab["t"], eps=eps)[0]
        if not hp.cond_prior:
            q = dict(z=q)
        torch.testing.assert_close(q["z"], ab["z"], **TOL)
        zs = simple_ref.abduct(sd, hp, x, pa, cf_parents=cf_pa, alpha=ab["alpha"], t=ab["t"], eps=eps)[0]
        torch.testing.assert_close(zs, ab["zstar"], **TOL)
        rl, rs = simple_ref.forward_latents(sd, hp, [q["z"]], pa, t=ab["t"])
        cl, cs = simple_ref.forward_latents(sd, hp, [zs], cf_pa, t=ab["t"])
        torch.testing.assert_close(rl, ab["rec_loc"], **TOL)
        torch.testing.assert_close(cs, ab["cf_scale"], **TOL)
        cf = torch.clamp(cl + cs * ((x - rl) / rs.clamp(min=1e-12)), min=-1, max=1)
        torch.testing.assert_close(cf, ab["cf_x"], rtol=1e-4, atol=1e-4)
        s = fx["sample"]
        sx, ss = simple_ref.sample(sd, hp, pa, t=s["t"], eps=eps)
        torch.testing.assert_close(sx, s["x"], **TOL)
        torch.testing.assert_close(ss, s["scale"], **TOL)


@pytest.mark.parametrize("tag", ["default_p3", "ukbb_light_p1"])
def test_dscm_forward_pinned_by_the_reference(tag):
    """oracle/dscm_ref.py against tests/golden/dscm_*.pt -- outputs of the REFERENCE's own ``DSCM.forward`` (src/pgm/dscm.py:30-95,
    run by oracle/make_dscm_golden.py with pyro / torchvision / imageio / seaborn stubbed in sys.modules and the duck-typed
    pgm / predictor / ELBO of oracle/dscm_stubs.py): parent preprocessing incl. the UKBB log-standardisation (dscm.py:98-132), factual
    ELBO, particle mean and variance (dscm.py:58-72), auxiliary loss, damped Lagrangian (dscm.py:85-88) and d loss / d theta,
    d loss / d lambda through the counterfactual branch."""
    from types import SimpleNamespace

    from oracle import dscm_ref, dscm_stubs as S, hvae_ref

    fx = load_golden("dscm_%s.pt" % tag)
    hp = SimpleNamespace(**fx["hp"])
    obs, do, c = fx["obs"], fx["do"], fx["constants"]
    x, B = obs["x"], obs["x"].shape[0]
    ukbb = "ukbb" in fx["dataset"]
    pa = {k: v for k, v in obs.items() if k != "x"}
    parents = dscm_ref.expand_parents({k: v.clone() for k, v in pa.items()}, fx["parents_x"], hp.input_res, ukbb=ukbb)
    assert torch.allclose(parents[:, :, 0, 0], fx["vae_parents"], rtol=0, atol=1e-6)  # vae_preprocess / ukbb_preprocess
    pgm = S.StubPGM()
    cf_dicts = [pgm.counterfactual(obs=pa, intervention=do, num_particles=1) for _ in range(fx["particles"])]
    cf_list = [dscm_ref.expand_parents({k: v.clone() for k, v in d.items()}, fx["parents_x"], hp.input_res, ukbb=ukbb) for d in cf_dicts]
    for k, v in fx["cf_parents"].items():
        assert torch.equal(cf_dicts[-1][k], v)  # dscm.py:74 keeps the LAST particle's parents
    sd = {k: v.clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
    lm = torch.tensor([c["lmbda_init"]], requires_grad=True)
    elbo_fn = S.StubELBO(fx["w"])
    noise = hvae_ref._Noise([e.clone() for e in fx["eps"]])
    out = dscm_ref.dscm_forward(sd, hp, x, parents, cf_list, c["beta"], t_abduct=fx["t_abduct"], noise=noise,
                                aux_fn=lambda cx: elbo_fn.differentiable_loss(None, None, x=cx, **cf_dicts[-1]) / B, lmbda=lm,
                                eps=torch.tensor([c["elbo_constraint"]]), damping=c["damping"])
    assert not noise.source, "every draw of the reference must be consumed, in its order"
    for k in ("elbo", "nll", "kl", "loss", "aux_loss"):
        assert abs(float(out[k].detach()) - float(fx["out"][k])) <= 2e-5 * abs(float(fx["out"][k])) + 1e-6, (k, float(out[k].detach()), float(fx["out"][k]))
    assert float((out["cf_x"] - fx["cf_x"]).abs().max()) < 2e-5
    if fx["particles"] > 1:
        assert float((out["var_cf_x"] - fx["var_cf_x"]).abs().max()) < 2e-5
    else:
        assert out["var_cf_x"] is None and fx["var_cf_x"] is None
    out["loss"].sum().backward()
    assert abs(float(lm.grad) - float(fx["lmbda_grad"])) <= 2e-5 * abs(float(fx["lmbda_grad"])) + 1e-6
    n = 0
    for name, g in fx["grads"].items():
        if float(g.abs().max()) == 0.0:
            continue
        d = float((sd[name].grad - g).abs().max()) / float(g.abs().max())
        assert d < 2e-3, (name, d)
        n += 1
    assert n > 50


def test_dmol_low_bit_branch():
    """dmol.py:52-60, 88-102 (low_bit=True: 5-bit pixels) -- oracle vs the reference-made vectors (oracle/make_dmol_lowbit_golden.py),
    values and gradient; the 8-bit branch on the same inputs gives other numbers."""
    from oracle import dmol_ref

    d = load_golden("dmol_lowbit.pt")
    l = d["l"].clone().requires_grad_(True)
    loss = dmol_ref.dmol_nll(d["x"], l, low_bit=True)
    torch.testing.assert_close(loss.detach(), d["loss"], rtol=1e-5, atol=1e-6)
    (g,) = torch.autograd.grad(loss.sum(), l)
    torch.testing.assert_close(g, d["grad_l"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dmol_ref.dmol_nll(d["x"], d["l"]), d["loss_8bit"], rtol=1e-5, atol=1e-6)
    assert float((d["loss"] - d["loss_8bit"]).abs().min()) > 1.0
