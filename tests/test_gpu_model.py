"""HIP path vs the golden fixtures made from the imported reference (tests/golden) and vs the oracle.
Tolerances are the north-star's: ELBO / NLL / KL 1e-4 relative, counterfactual pixels 1e-3 absolute (f32 path)."""
import copy
import glob
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

TINY = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "tiny_*.pt")))


def build(fx, dtype="f32"):
    from causal_gen_amd import dmol, vae
    from causal_gen_amd.hps import Hparams

    args = Hparams(**fx["hp"])
    m = vae.HVAE(args)
    if fx["likelihood"] == "dmol":
        m.likelihood = dmol.DmolNet(args)
    m.load_state_dict(fx["state_dict"])
    m.compute_dtype = dtype
    return m.cuda().eval(), args


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)


@pytest.mark.parametrize("name", TINY)
def test_forward_elbo_and_grads(name):
    fx = load_golden(name)
    m, _ = build(fx)
    f = fx["fwd"]
    m.noise = [e.clone() for e in f["eps"]]
    out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=f["beta"])
    for k in ("elbo", "nll", "kl"):
        assert rel(out[k], f[k]) < 1e-4, (k, float(out[k]), float(f[k]))
    out["elbo"].backward()
    torch.cuda.synchronize()
    got = {n: p.grad for n, p in m.named_parameters()}
    worst = 0.0
    for n, g in f["grads"].items():
        assert got[n] is not None, n
        d = (got[n].cpu() - g).abs().max().item()
        scale = g.abs().max().item() + 1e-8
        worst = max(worst, d / scale)
        # measured on MI355X (f32 MFMA 16x16x4, f32 accumulation in a different order than ATen's CPU kernels, fast exp / tanh):
        # 3e-5 .. 3.7e-4 of the tensor's largest entry over the seven fixtures; the bound leaves a factor ~2
        assert d / scale < 7e-4, (n, d, scale)
    for n, p in m.named_parameters():
        if n not in f["grads"]:
            assert p.grad is None, n
    print(name, "worst grad rel-to-max err", worst)


@pytest.mark.parametrize("name", TINY)
def test_abduct_replay_counterfactual(name):
    fx = load_golden(name)
    m, args = build(fx)
    x, pa, cf_pa = fx["x"].cuda(), fx["pa"].cuda(), fx["cf_pa"].cuda()
    ab = fx["abduct"]
    m.noise = [e.clone() for e in ab["eps"]]
    zs = m.abduct(x, pa, t=ab["t"])
    if args.cond_prior:
        for z, ql, qs in zip(zs, ab["q_loc"], ab["q_logscale"]):
            torch.testing.assert_close(z["q_loc"].cpu().contiguous(), ql, rtol=1e-3, atol=1e-4)
            torch.testing.assert_close(z["q_logscale"].cpu().contiguous(), qs, rtol=1e-3, atol=1e-4)
        zs = [z["z"] for z in zs]
    assert len(zs) == len(ab["zs"])
    for a, b in zip(zs, ab["zs"]):
        assert a.shape == b.shape
        torch.testing.assert_close(a.cpu().contiguous(), b, rtol=1e-3, atol=2e-4)
    m.noise = None
    rec_loc, rec_scale = m.forward_latents(zs, pa)
    cf_loc, cf_scale = m.forward_latents(zs, cf_pa)
    from causal_gen_amd.dscm import cf_pixels

    cf_x = cf_pixels(x, rec_loc, rec_scale, cf_loc, cf_scale)
    c = fx["cf"]
    assert (rec_loc.cpu() - c["rec_loc"]).abs().max() < 1e-3
    assert (cf_loc.cpu() - c["cf_loc"]).abs().max() < 1e-3
    torch.testing.assert_close(rec_scale.cpu(), c["rec_scale"], rtol=2e-3, atol=1e-6)
    # cf pixels: 1e-3 abs wherever the abducted pixel noise u is not astronomically amplified by a tiny scale
    ok = c["rec_scale"] > 1e-3
    assert ((cf_x.cpu() - c["cf_x"]).abs()[ok]).max() < 1e-3
    # partial latents + temperature: injected prior noise
    pl = fx["partial_latents"]
    m.noise = [e.clone() for e in pl["eps"]]
    loc, sc = m.forward_latents(zs[: pl["n"]], cf_pa, t=pl["t"])
    assert (loc.cpu() - pl["loc"]).abs().max() < 1e-3
    if "mediator" in fx:
        md = fx["mediator"]
        m.noise = [e.clone() for e in md["eps"]]
        zstar = m.abduct(x, pa, cf_parents=cf_pa, alpha=md["alpha"], t=md["t"])
        for a, b in zip(zstar, md["zstar"]):
            torch.testing.assert_close(a.cpu().contiguous(), b, rtol=1e-3, atol=3e-4)
    sm = fx["sample"]
    m.noise = [e.clone() for e in sm["eps"]]
    loc, sc = m.sample(pa, t=sm["t"])
    assert (loc.cpu() - sm["loc"]).abs().max() < 1e-3


def test_train_mode_drop_cond():
    fx = load_golden("tiny_condprior_morpho_c1.pt")
    m, _ = build(fx)
    d = fx["fwd_drop"]
    m.train()
    m.decoder.drop_cond = lambda: d["drop"]
    m.noise = [e.clone() for e in d["eps"]]
    with torch.no_grad():
        out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=1.0)
    for k in ("elbo", "nll", "kl"):
        assert rel(out[k], d[k]) < 1e-4, k


def test_deepcopy_and_state_dict_roundtrip():
    fx = load_golden("tiny_default_c1.pt")
    m, _ = build(fx)
    f = fx["fwd"]
    m.noise = [e.clone() for e in f["eps"]]
    with torch.no_grad():
        a = m(fx["x"].cuda(), fx["pa"].cuda(), beta=f["beta"])["elbo"].item()
    m2 = copy.deepcopy(m)
    assert list(m2.state_dict().keys()) == list(fx["state_dict"].keys())
    m2.noise = [e.clone() for e in f["eps"]]
    with torch.no_grad():
        b = m2(fx["x"].cuda(), fx["pa"].cuda(), beta=f["beta"])["elbo"].item()
    assert a == b
    for k, v in m.state_dict().items():
        torch.testing.assert_close(v.cpu(), fx["state_dict"][k])


def test_philox_noise_statistics_and_determinism():
    fx = load_golden("tiny_light_c1.pt")
    m, _ = build(fx)
    x, pa = fx["x"].cuda(), fx["pa"].cuda()
    with torch.no_grad():
        torch.manual_seed(5)
        m.__dict__["_eng"] = None
        a = [m(x, pa)["elbo"].item() for _ in range(3)]
        torch.manual_seed(5)
        m.__dict__["_eng"] = None
        b = [m(x, pa)["elbo"].item() for _ in range(3)]
    assert a == b and len(set(a)) == 3  # same seed => same stream; fresh noise every call
    from causal_gen_amd import _lib

    lib = _lib.load()
    out = torch.empty(1 << 20, device="cuda")
    rng = torch.tensor([123, 0], dtype=torch.int64, device="cuda")
    lib.philox_normal(out.data_ptr(), out.numel(), rng.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
    assert abs(out.mean().item()) < 5e-3 and abs(out.std().item() - 1) < 5e-3
    assert abs((out ** 4).mean().item() - 3) < 0.05


def test_bf16_path_close_to_fixture():
    """bf16 storage/MFMA: report-and-bound the deviation (not a parity claim)."""
    fx = load_golden("tiny_light_c1.pt")
    m, _ = build(fx, dtype="f16")
    f = fx["fwd"]
    m.noise = [e.clone() for e in f["eps"]]
    out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=f["beta"])
    assert rel(out["nll"], f["nll"]) < 3e-2 and rel(out["elbo"], f["elbo"]) < 3e-2
    out["elbo"].backward()
    g = dict(m.named_parameters())["likelihood.x_loc.weight"].grad.cpu()
    ref = f["grads"]["likelihood.x_loc.weight"]
    assert (g - ref).abs().max() / ref.abs().max() < 0.1


def test_sample_return_loc_false_adds_scaled_noise():
    """DGaussNet.sample(return_loc=False) (vae.py:416-421): x = clamp(loc + scale * N(0,1)); latents replayed so that the
    only randomness is the pixel noise."""
    fx = load_golden("tiny_light_c1.pt")
    m, _ = build(fx)
    x, pa = fx["x"].cuda(), fx["pa"].cuda()
    zs = m.abduct(x, pa)
    loc, scale = m.forward_latents(zs, pa)
    eng = m.engine()
    eng.begin()
    eng.recording = False
    eng.prepare_weights()
    pnt = eng.from_nchw(pa.float())
    lat = [eng.from_nchw(z.float()) for z in zs]
    h, _ = m._decode(eng, pnt, latents=lat)
    xs, ss = m._sample_likelihood(eng, h, return_loc=False)
    torch.testing.assert_close(ss, scale)
    assert float(xs.abs().max()) <= 1.0 and not torch.equal(xs, loc)
    inside = (loc.abs() + 4 * scale) < 1.0  # pixels the clamp cannot touch
    if int(inside.sum()) > 200:
        u = ((xs - loc) / scale)[inside]
        assert abs(float(u.mean())) < 0.2 and abs(float(u.std()) - 1.0) < 0.2


@pytest.mark.parametrize("name", ["tiny_light_c1.pt", "tiny_condprior_morpho_c1.pt", "tiny_default_c3.pt"])
def test_free_bits_forward_and_grads(name):
    """kl_free_bits > 0 (vae.py:443-449): values against the reference's own outputs (golden), gradients against autograd
    through the oracle with the same eps."""
    from oracle import hvae_ref

    if name not in TINY:
        pytest.skip("fixture not generated")
    fx = load_golden(name)
    d = fx["fwd_freebits"]
    m, _ = build(fx)
    m.free_bits = d["free_bits"]
    m.noise = [e.clone() for e in d["eps"]]
    out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=1.0)
    for k in ("elbo", "nll", "kl"):
        assert rel(out[k], d[k]) < 1e-4, (k, float(out[k]), float(d[k]))
    out["elbo"].backward()
    torch.cuda.synchronize()
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx["state_dict"].items()}
    hp = SimpleNamespace(**{**fx["hp"], "kl_free_bits": d["free_bits"]})
    o = hvae_ref.hvae_forward(sd, hp, fx["x"], fx["pa"], beta=1.0, noise=[e.clone() for e in d["eps"]])
    o["elbo"].backward()
    worst = 0.0
    for n, p in m.named_parameters():
        ref = sd[n].grad
        if ref is None or ref.abs().max() == 0:
            continue
        assert p.grad is not None, n
        err = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-8)
        worst = max(worst, err)
        assert err < 2e-3, (n, err)
    print(name, "free-bits worst grad rel-to-max err", worst)


def test_u8_pixels_are_normalised_on_the_device():
    """trainer.py:17 fused into the load: feeding the raw u8 batch equals feeding (x - 127.5) / 127.5."""
    from causal_gen_amd.train import preprocess_batch

    fx = load_golden("tiny_default_c3.pt")
    m, args = build(fx)
    g = torch.Generator().manual_seed(9)
    xu = torch.randint(0, 256, tuple(fx["x"].shape), generator=g, dtype=torch.uint8)
    xf = (xu.float() - 127.5) / 127.5
    pa = fx["pa"].cuda()
    outs = []
    for x in (xf.cuda(), xu.cuda()):
        m.noise = [e.clone() for e in fx["fwd"]["eps"]]
        with torch.no_grad():
            outs.append(m(x, pa, beta=1.0))
    for k in ("elbo", "nll", "kl"):
        assert rel(outs[1][k], outs[0][k]) < 1e-5, (k, float(outs[1][k]), float(outs[0][k]))
    b = preprocess_batch(SimpleNamespace(device="cuda", input_res=xu.shape[-1]), {"x": xu, "pa": fx["pa"][:, :, 0, 0]}, expand_pa=True)
    torch.testing.assert_close(b["x"].cpu(), xf, rtol=0, atol=2e-7)
    assert tuple(b["pa"].shape) == tuple(fx["pa"].shape) and b["x"].shape == xf.shape


def test_concurrent_replays_equal_sequential_replays_on_first_use(monkeypatch):
    """dscm.counterfactual runs its two replays as two concurrent streams (HVAE.forward_latents_pair).  On a FRESH engine the
    decoder biases' NHWC images are created lazily inside the side-stream section and cached for everybody: the main
    stream once read them before the conversion had run (tools/fuzz_model.py).  Sequential and concurrent replays must give
    the same pixels bit for bit, first call included."""
    from causal_gen_amd import dscm

    name = "tiny_light_c1.pt"
    if name not in TINY:
        pytest.skip("fixture not generated")
    fx = load_golden(name)
    outs = []
    for pair in ("0", "1", "1"):
        monkeypatch.setenv("CGEN_CF_PAIR", pair)
        m, _ = build(fx, "f16")  # fresh model => fresh engine => empty caches
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        eng = m.engine()
        eng.rng_ptr()
        eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
        with torch.no_grad():
            zs = m.abduct(x, pa)
            zs = zs[: max(1, len(zs) // 2)]  # the remaining blocks sample from the prior: the two replays must draw the
            if pair == "0":                  # noise two consecutive forward_latents calls would have drawn
                a, b = m.forward_latents(zs, pa), m.forward_latents(zs, pa.roll(1, 0))
            else:
                a, b = m.forward_latents_pair(zs, pa, pa.roll(1, 0))
            after = m.engine().rng.clone()
        torch.cuda.synchronize()
        outs.append((a[0].clone(), a[1].clone(), b[0].clone(), b[1].clone(), after.float()))
    for o in outs[1:]:
        for u, v in zip(outs[0], o):
            assert torch.equal(u, v)


@pytest.mark.parametrize("name,te", [("tiny_light_c1.pt", False), ("tiny_condprior_morpho_c1.pt", False),
                                     ("tiny_condprior_morpho_c1.pt", True), ("tiny_default_c3.pt", False)])
def test_counterfactual_reusing_the_abduction_pass_gives_the_same_bits(name, te, monkeypatch):
    """dscm.counterfactual takes the reconstruction from the abduction pass (HVAE.abduct_with_reconstruction) instead of
    replaying the latents under the observed parents: same pixels and same Philox state afterwards as the reference's
    three-call sequence abduct -> forward_latents(parents) -> forward_latents(cf_parents), bit for bit."""
    from causal_gen_amd import dscm

    if name not in TINY:
        pytest.skip("fixture not generated")
    fx = load_golden(name)
    outs = []
    for reuse in ("0", "1"):
        monkeypatch.setenv("CGEN_CF_REUSE", reuse)
        monkeypatch.setenv("CGEN_CF_PAIR", "0")
        for dt in ("f32", "f16"):
            m, _ = build(fx, dt)
            x, pa = fx["x"].cuda(), fx["pa"].cuda()
            eng = m.engine()
            eng.rng_ptr()
            eng.rng.copy_(torch.tensor([33, 0], dtype=torch.int64, device=eng.rng.device))
            with torch.no_grad():
                cf = dscm.counterfactual(m, x, pa, pa.roll(1, 0), t_abduct=0.7, te_cf=te, alpha=0.4)
            torch.cuda.synchronize()
            outs.append((reuse, dt, cf.clone(), eng.rng.clone()))
    for (r0, d0, c0, g0), (r1, d1, c1, g1) in zip(outs[:2], outs[2:]):
        assert d0 == d1 and torch.equal(c0, c1) and torch.equal(g0, g1), (d0,)


def test_two_stream_inference_sections_are_bit_identical_to_one_stream():
    """non-recording passes with encoder activations (abduct, eval forward) run the decoder's prior / posterior sections on
    two streams by default (Engine.infer_branch, CGEN_INFER_BRANCH).  Fork / join only order launches: the latents of an abduction pass,
    the eval-mode ELBO and the counterfactual loop under hipGraph capture must be bit-identical with the sections on one stream."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from causal_gen_amd import dscm

    m, hp = bench.build_model("ukbb192", "f16")
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
    x, pa = bench.synth_batch("ukbb192", hp, 4, "cuda", 3)
    cf_pa = pa.roll(1, 0)
    eng = m.engine()
    eng.rng_ptr()
    assert eng.infer_branch, "two-stream inference sections are the default"
    res = {}
    for two in (True, False):
        eng.infer_branch = two
        with torch.no_grad():
            eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
            zs = m.abduct(x, pa)
            eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
            out = m(x, pa, beta=1.0)
            eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
            cf = dscm.counterfactual(m, x, pa, cf_pa, t_abduct=1.0)
            gcf = dscm.GraphedCounterfactual(m, t_abduct=1.0)
            gcf(x, pa, cf_pa)  # eager warm-up + capture (with the sections as `two` says)
            eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
            cfg = gcf(x, pa, cf_pa).clone()  # replay
        torch.cuda.synchronize()
        res[two] = ([z.clone() for z in zs], [out[k].clone() for k in ("elbo", "nll", "kl")], cf.clone(), cfg)
    eng.infer_branch = True
    assert len(res[True][0]) == len(res[False][0]) > 30
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][2], res[False][2])
    assert torch.equal(res[True][3], res[False][3])
