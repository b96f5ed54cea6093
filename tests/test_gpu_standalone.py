"""The reference's module classes are importable and callable on their own: ``Block(...)(x)``, ``Encoder(args)(x)``,
``DGaussNet(args)`` forward / nll / sample run on the HIP engine (inference only) and match the oracle's restatement of
vae.py:73-84, 112-134, 352-422 on the module's own state_dict."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sd(mod, prefix):
    return {prefix + k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}


@pytest.mark.parametrize("version,d,cin,cout", [("light", None, 32, 32), ("light", 2, 32, 48), (None, None, 16, 16), (None, 1.5, 24, 16),
                                                 ("light", None, 48, 32)])
def test_block_standalone(version, d, cin, cout):
    from causal_gen_amd import vae
    from oracle import hvae_ref

    torch.manual_seed(1)
    blk = vae.Block(cin, cin // 4, cout, down_rate=d, version=version).cuda()
    x = torch.randn(3, cin, 12, 12)
    y = blk(x.cuda())
    ref = hvae_ref._block(_sd(blk, "b."), "b", x, version == "light", 3, True, d)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError, match="inference-only"):
        blk(x.cuda().requires_grad_(True))
    blk.compute_dtype = "f16"
    yb = blk(x.cuda())
    assert float((yb.cpu() - ref).abs().max()) <= 0.05 * float(ref.abs().max())


def test_encoder_standalone():
    from causal_gen_amd import vae
    from causal_gen_amd.hps import setup_hparams
    from oracle import hvae_ref

    hp = setup_hparams("morphomnist")
    hp.vr = None  # (vae.py:428: HVAE.__init__ sets args.vr before it builds the Encoder)
    torch.manual_seed(2)
    enc = vae.Encoder(hp).cuda()
    x = torch.randn(4, hp.input_channels, hp.input_res, hp.input_res)
    acts = enc(x.cuda())
    ref = hvae_ref.encode(_sd(enc, "encoder."), hp, x)
    assert sorted(acts) == sorted(ref)
    for r in ref:
        torch.testing.assert_close(acts[r].cpu(), ref[r], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("C", [1, 3])
def test_dgaussnet_standalone(C):
    from causal_gen_amd import vae
    from causal_gen_amd.hps import setup_hparams
    from oracle import hvae_ref

    hp = setup_hparams("morphomnist", input_channels=C)
    torch.manual_seed(3)
    lk = vae.DGaussNet(hp).cuda()
    with torch.no_grad():
        for p in lk.parameters():
            p.add_(0.3 * torch.randn_like(p))
    sd = _sd(lk, "likelihood.")
    g = torch.Generator().manual_seed(4)
    h = torch.randn(5, hp.widths[0], 16, 16, generator=g)
    x = (torch.randint(0, 256, (5, C, 16, 16), generator=g).float() - 127.5) / 127.5
    x[0, :, :2] = -1.0
    x[1, :, :2] = 1.0
    loc, ls = lk(h.cuda())
    rloc, rls = hvae_ref.dgauss_params(sd, hp, h)
    torch.testing.assert_close(loc.cpu(), rloc, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ls.cpu(), rls, rtol=1e-4, atol=1e-5)
    loc, ls = lk(h.cuda(), x.cuda(), t=0.7)
    rloc, rls = hvae_ref.dgauss_params(sd, hp, h, x, t=0.7)
    torch.testing.assert_close(loc.cpu(), rloc, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ls.cpu(), rls, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(lk.nll(h.cuda(), x.cuda()).cpu(), hvae_ref.dgauss_nll(sd, hp, h, x), rtol=1e-4, atol=1e-6)
    xs, sc = lk.sample(h.cuda())
    rx, rs = hvae_ref.dgauss_sample(sd, hp, h)
    torch.testing.assert_close(xs.cpu(), rx, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sc.cpu(), rs, rtol=1e-4, atol=1e-6)
    xr, _ = lk.sample(h.cuda(), return_loc=False)
    assert torch.isfinite(xr).all() and float(xr.abs().max()) <= 1.0 and not torch.equal(xr, xs)


@pytest.mark.parametrize("C", [1, 3])
def test_dgauss_sampling_is_the_reference_formula_on_the_kernels_own_noise(C):
    """DGaussNet.sample(h, return_loc=False) (vae.py:413-422): x = clamp(loc + exp(logscale) * eps).  The kernel draws eps from
    Philox at the NCHW element index; `cgen_philox_normal` reproduces that stream on the same state, and the oracle's restatement of
    the reference formula fed with it must give the same pixels -- sampling parity, not just finite / in-range / reproducible."""
    from causal_gen_amd import _lib, vae
    from causal_gen_amd.hps import setup_hparams
    from oracle import hvae_ref

    hp = setup_hparams("morphomnist", input_channels=C)
    torch.manual_seed(7)
    lk = vae.DGaussNet(hp).cuda()
    with torch.no_grad():
        for p in lk.parameters():
            p.add_(0.3 * torch.randn_like(p))
    sd = _sd(lk, "likelihood.")
    h = torch.randn(4, hp.widths[0], 12, 12, generator=torch.Generator().manual_seed(8))
    xs, sc = lk.sample(h.cuda(), return_loc=False)
    eng = vae._SA_ENGINES[lk]
    eps = torch.empty(xs.numel(), device="cuda")
    _lib.load().philox_normal(eps.data_ptr(), eps.numel(), eng.rng_ptr(), 978, torch.cuda.current_stream().cuda_stream)
    eps = eps.view_as(xs).cpu()
    rx, rs = hvae_ref.dgauss_sample(sd, hp, h, return_loc=False, noise=lambda like: eps)
    torch.testing.assert_close(sc.cpu(), rs, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(xs.cpu(), rx, rtol=1e-4, atol=2e-5)
    assert float((xs.cpu() - hvae_ref.dgauss_sample(sd, hp, h)[0]).abs().max()) > 1e-3  # (the noise did something)
    xs2, _ = lk.sample(h.cuda(), return_loc=False)
    assert not torch.equal(xs, xs2)  # a fresh draw per call


@pytest.mark.parametrize("name", ["morphomnist", "ukbb192"])
def test_decoder_standalone(name):
    """Decoder.forward (vae.py:222-301) on its own: posterior pass on an encoder's activations (h and every block's KL map,
    z when abducting), replay of those latents, and prior sampling -- against the oracle's decode on the module's own
    state_dict, with the eps the oracle drew injected (``decoder.noise``)."""
    from causal_gen_amd import vae
    from causal_gen_amd.hps import setup_hparams
    from oracle import hvae_ref

    hp = setup_hparams(name)
    hp.vr = "light" if "ukbb" in hp.hps else None
    torch.manual_seed(4)
    enc, dec = vae.Encoder(hp).cuda(), vae.Decoder(hp).cuda()
    with torch.no_grad():
        for p in dec.parameters():  # (the prior heads are zero-initialised: give the KL something to measure)
            p.add_(0.02 * torch.randn_like(p))
    dec.eval()
    B = 2
    g = torch.Generator().manual_seed(5)
    x = (torch.randint(0, 256, (B, hp.input_channels, hp.input_res, hp.input_res), generator=g).float() - 127.5) / 127.5
    pc = torch.randn(B, hp.context_dim, generator=g)
    pa = pc[..., None, None].repeat(1, 1, hp.input_res, hp.input_res)
    sd = {**_sd(enc, "encoder."), **_sd(dec, "decoder.")}
    acts_ref = hvae_ref.encode(sd, hp, x)
    noise = hvae_ref._Noise()
    h_ref, stats_ref = hvae_ref.decode(sd, hp, pa, acts=acts_ref, abduct=True, noise=noise)
    acts = enc(x.cuda())
    dec.noise = [e.clone() for e in noise.drawn]
    h, stats = dec(pa.cuda(), x=acts, abduct=True)
    assert not dec.noise and len(stats) == len(stats_ref) > 0
    torch.testing.assert_close(h.cpu(), h_ref, rtol=2e-3, atol=2e-3)
    for s, r in zip(stats, stats_ref):
        torch.testing.assert_close(s["kl"].cpu(), r["kl"], rtol=2e-3, atol=2e-4)
        zr = r["z"]["z"] if isinstance(r["z"], dict) else r["z"]
        zs = s["z"]["z"] if isinstance(s["z"], dict) else s["z"]
        torch.testing.assert_close(zs.cpu(), zr, rtol=2e-3, atol=2e-3)
    # replay of the abducted latents under the same parents reproduces h; the broadcast parent vector is accepted as well
    zs = [(s["z"]["z"] if isinstance(s["z"], dict) else s["z"]) for s in stats]
    lat = [zs.pop(0) if b.stochastic else None for b in dec.blocks]
    h2, st2 = dec(pc.cuda(), latents=lat)
    assert st2 == []
    torch.testing.assert_close(h2, h, rtol=1e-3, atol=1e-3)
    h3, _ = dec(pa.cuda())  # prior sampling
    assert torch.isfinite(h3).all() and not torch.equal(h3, h)


def test_dmol_python_surface_against_reference_vectors():
    """The module-level names of src/dmol.py (loss dmol.py:24-118, means :164-215, sampling :121-161) are importable and callable
    on channels-last tensors; values and the loss gradient against the reference-made ``ops.pt``."""
    import os

    from causal_gen_amd import dmol
    from oracle import dmol_ref

    d = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ops.pt"))["dmol"]
    l = d["l"].cuda().requires_grad_(True)
    x = d["x"].cuda()
    loss = dmol.discretized_mix_logistic_loss(x, l)
    assert loss.shape == (3,)
    torch.testing.assert_close(loss.detach().cpu(), d["loss"], rtol=1e-4, atol=1e-6)  # north-star: DMoL nats/dim within 1e-4 rel
    loss.sum().backward()
    torch.testing.assert_close(l.grad.cpu(), d["grad_l"], rtol=2e-3, atol=2e-6)
    # a non-trivial upstream gradient scales per sample
    l2 = d["l"].cuda().requires_grad_(True)
    wts = torch.tensor([0.5, -2.0, 3.0], device="cuda")
    (dmol.discretized_mix_logistic_loss(x, l2) * wts).sum().backward()
    torch.testing.assert_close(l2.grad.cpu(), d["grad_l"] * wts.cpu()[:, None, None, None], rtol=2e-3, atol=6e-6)
    for mask in ("soft", "hard", "top3"):
        m, s = dmol.mean_discretized_mix_logistic(d["l"].cuda(), 10, mask=mask, return_scale=True)
        assert m.shape == (3, 7, 7, 3)
        torch.testing.assert_close(m.cpu(), d[f"mean_{mask}"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(s.cpu(), d[f"scale_{mask}"], rtol=1e-4, atol=1e-6)
    assert dmol.mean_discretized_mix_logistic(d["l"].cuda(), 10).shape == (3, 7, 7, 3)
    a, sa = dmol.sample_from_discretized_mix_logistic(d["l"].cuda(), 10, return_scale=True, t=0.7)
    b = dmol.sample_from_discretized_mix_logistic(d["l"].cuda(), 10)
    assert a.shape == (3, 7, 7, 3) and torch.isfinite(a).all() and a.abs().max() <= 1 and (sa > 0).all() and not torch.equal(a, b)
    with pytest.raises(Exception):
        dmol.discretized_mix_logistic_loss(d["x"], d["l"])  # CPU tensors: no fallback


def test_dmolnet_standalone():
    """DmolNet.forward / nll / sample (dmol.py:218-245) on their own against the oracle on the module's own state_dict."""
    from causal_gen_amd import dmol
    from causal_gen_amd.hps import setup_hparams
    from oracle import dmol_ref

    hp = setup_hparams("cmnist")
    torch.manual_seed(5)
    net = dmol.DmolNet(hp).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(3.0)
    sd = _sd(net, "likelihood.")
    g = torch.Generator().manual_seed(6)
    h = torch.randn(3, hp.widths[0], 9, 9, generator=g)
    x = (torch.randint(0, 256, (3, 3, 9, 9), generator=g).float() - 127.5) / 127.5
    l = net(h.cuda())
    assert l.shape == (3, 9, 9, 100)
    torch.testing.assert_close(l.cpu(), dmol_ref.dmolnet_logits(sd, h), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(net.nll(h.cuda(), x.cuda()).cpu(), dmol_ref.dmolnet_nll(sd, h, x), rtol=1e-4, atol=1e-6)
    for mask in ("soft", "hard", "top2"):
        net.mask = mask
        xs, sc = net.sample(h.cuda())
        rx, rs = dmol_ref.dmolnet_sample(sd, h, mask=mask)
        torch.testing.assert_close(xs.cpu(), rx, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(sc.cpu(), rs, rtol=1e-4, atol=1e-6)
    xs, sc = net.sample(h.cuda(), return_loc=False, t=0.5)
    assert xs.shape == (3, 3, 9, 9) and xs.abs().max() <= 1 and torch.isfinite(sc).all()


@pytest.mark.parametrize("name,cond_prior", [("morphomnist", True), ("ukbb192", False)])
def test_decoder_block_halves_standalone(name, cond_prior):
    """DecoderBlock.forward_prior / forward_posterior (vae.py:170-192) on their own against the oracle's Block on the module's own
    state_dict (virtual cat[z, pa] / cat[z, pa, x]; temperature on the logscales)."""
    import math

    from causal_gen_amd import vae
    from causal_gen_amd.hps import setup_hparams
    from oracle import hvae_ref

    hp = setup_hparams(name)
    hp.vr = "light" if "ukbb" in hp.hps else None
    hp.cond_prior = cond_prior
    torch.manual_seed(7)
    w, res = 64, 12
    blk = vae.DecoderBlock(hp, w, w, res).cuda()
    g = torch.Generator().manual_seed(8)
    z = torch.randn(2, w, res, res, generator=g)
    xa = torch.randn(2, w, res, res, generator=g)
    pc = torch.randn(2, hp.context_dim, generator=g)
    pa = pc[..., None, None].repeat(1, 1, res, res)
    light = hp.vr == "light"
    sd = _sd(blk, "b.")
    ref_p = hvae_ref._block(sd, "b.prior", torch.cat([z, pa], 1) if cond_prior else z, light, 3, False, None)
    zd = hp.z_dim
    p_loc, p_ls, p_feat = blk.forward_prior(z.cuda(), pa.cuda() if cond_prior else None, t=0.8)
    torch.testing.assert_close(p_loc.cpu(), ref_p[:, :zd], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(p_ls.cpu(), ref_p[:, zd:2 * zd] + math.log(0.8), rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(p_feat.cpu(), ref_p[:, 2 * zd:], rtol=2e-4, atol=2e-5)
    ref_q = hvae_ref._block(sd, "b.posterior", torch.cat([z, pa, xa], 1), light, 3, False, None)
    q_loc, q_ls = blk.forward_posterior(z.cuda(), xa.cuda(), pc.cuda())  # (the broadcast parent vector is accepted as well)
    torch.testing.assert_close(q_loc.cpu(), ref_q[:, :zd], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(q_ls.cpu(), ref_q[:, zd:], rtol=2e-4, atol=2e-5)
    with pytest.raises(RuntimeError, match="inference-only"):
        blk.forward_prior(z.cuda().requires_grad_(True), pa.cuda())
