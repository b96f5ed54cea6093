"""Generate tests/golden/*.pt by running the REFERENCE itself (import from /root/reference/src).

Run in the build container only (the reference does not exist on the GPU box):

    python3 -B oracle/make_golden.py

Fixtures are data: inputs, injected noise and the reference's outputs.  No reference source text is
stored.  The reference is imported with bytecode writing disabled so nothing is written under
/root/reference.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/src"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import vae as ref_vae  # noqa: E402  (reference)
import dmol as ref_dmol  # noqa: E402  (reference)
from hps import Hparams  # noqa: E402  (reference)

from oracle import hparams as ohp  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_args(hp):
    a = Hparams()
    a.update(dict(vars(hp)))
    return a


class EpsTap:
    """Replace vae.sample_gaussian so every draw is recorded (SURVEY probe C.6)."""

    def __enter__(self):
        self.eps = []
        self._orig = ref_vae.sample_gaussian

        def tapped(loc, logscale):
            e = torch.randn_like(loc)
            self.eps.append(e.clone())
            return loc + logscale.exp() * e

        ref_vae.sample_gaussian = tapped
        return self

    def __exit__(self, *a):
        ref_vae.sample_gaussian = self._orig


def randomise(model, gen):
    """At reference init the prior head is x0 and biases are zero, so most paths are dead: perturb."""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 4 and "decoder.bias" not in n:
                fan = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen) * (0.7 / np.sqrt(fan)))
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.1)


def make_inputs(hp, B, gen):
    R, C = hp.input_res, hp.input_channels
    x = (torch.randint(0, 256, (B, C, R, R), generator=gen).float() - 127.5) / 127.5
    pa = torch.randn(B, hp.context_dim, generator=gen)[..., None, None].repeat(1, 1, R, R)
    cf = torch.randn(B, hp.context_dim, generator=gen)[..., None, None].repeat(1, 1, R, R)
    return x, pa, cf


def tiny_fixture(tag, hp, likelihood="dgauss", B=3, seed=0):
    gen = torch.Generator().manual_seed(1000 + seed)
    torch.manual_seed(seed)
    a = ref_args(hp)
    m = ref_vae.HVAE(a)
    if likelihood == "dmol":
        m.likelihood = ref_dmol.DmolNet(a)
    randomise(m, gen)
    m.eval()
    x, pa, cf_pa = make_inputs(hp, B, gen)
    fx = dict(hp=dict(vars(hp)), likelihood=likelihood, x=x, pa=pa, cf_pa=cf_pa,
              state_dict={k: v.clone() for k, v in m.state_dict().items()})

    # ---- HVAE.forward (eval mode => drop_cond inactive) + grads of every parameter
    beta = 2.5
    torch.manual_seed(11)
    with EpsTap() as tap:
        out = m(x, pa, beta=beta)
    out["elbo"].backward()
    fx["fwd"] = dict(beta=beta, eps=tap.eps, elbo=out["elbo"].detach(), nll=out["nll"].detach(), kl=out["kl"].detach(),
                     grads={n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    m.zero_grad()
    # per-layer KL maps and final h via a second identical pass on the pieces
    with torch.no_grad(), EpsTap():
        torch.manual_seed(11)
        acts = m.encoder(x)
        h, stats = m.decoder(parents=pa, x=acts)
    fx["fwd"]["h"] = h
    fx["fwd"]["kl_maps"] = [s["kl"] for s in stats]
    fx["fwd"]["acts"] = {int(k): v for k, v in acts.items()}

    # ---- training-mode forward with a fixed conditioning-dropout draw (morphomnist only)
    if "morphomnist" in hp.hps and hp.cond_prior:
        m.train()
        m.decoder.drop_cond = lambda: (0, 1)
        torch.manual_seed(12)
        with torch.no_grad(), EpsTap() as tap:
            o2 = m(x, pa, beta=1.0)
        fx["fwd_drop"] = dict(drop=(0, 1), eps=tap.eps, **{k: v.detach() for k, v in o2.items()})
        m.eval()

    # ---- free-bits variant
    m.free_bits = 0.05
    torch.manual_seed(13)
    with torch.no_grad(), EpsTap() as tap:
        o3 = m(x, pa, beta=1.0)
    fx["fwd_freebits"] = dict(free_bits=0.05, eps=tap.eps, **{k: v.detach() for k, v in o3.items()})
    m.free_bits = 0.0

    with torch.no_grad():
        # ---- abduct (t=0.9) -> forward_latents x2 -> dscm.py:55-56
        torch.manual_seed(21)
        with EpsTap() as tap:
            zs = m.abduct(x, pa, t=0.9)
        ab = dict(t=0.9, eps=tap.eps)
        if hp.cond_prior:
            ab["q_loc"] = [z["q_loc"] for z in zs]
            ab["q_logscale"] = [z["q_logscale"] for z in zs]
            zs = [z["z"] for z in zs]
        ab["zs"] = zs
        fx["abduct"] = ab
        rec_loc, rec_scale = m.forward_latents(zs, pa)
        cf_loc, cf_scale = m.forward_latents(zs, cf_pa)
        u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
        fx["cf"] = dict(rec_loc=rec_loc, rec_scale=rec_scale, cf_loc=cf_loc, cf_scale=cf_scale, u=u,
                        cf_x=torch.clamp(cf_loc + cf_scale * u, min=-1, max=1))
        # partial latents: only the first two given -> the rest sampled from the prior
        torch.manual_seed(22)
        with EpsTap() as tap:
            pl_loc, pl_scale = m.forward_latents(zs[:2], cf_pa, t=0.7)
        fx["partial_latents"] = dict(n=2, t=0.7, eps=tap.eps, loc=pl_loc, scale=pl_scale)
        # ---- mediator z* (cond_prior only), vae.py:480-513
        if hp.cond_prior:
            torch.manual_seed(23)
            with EpsTap() as tap:
                zstar = m.abduct(x, pa, cf_parents=cf_pa, alpha=0.65, t=0.8)
            te_loc, te_scale = m.forward_latents(zstar, cf_pa)
            fx["mediator"] = dict(alpha=0.65, t=0.8, eps=tap.eps, zstar=zstar, loc=te_loc, scale=te_scale)
        # ---- prior sample
        torch.manual_seed(24)
        with EpsTap() as tap:
            s_loc, s_scale = m.sample(pa, t=0.85)
        fx["sample"] = dict(t=0.85, eps=tap.eps, loc=s_loc, scale=s_scale)
    path = os.path.join(OUT, f"tiny_{tag}.pt")
    torch.save(fx, path)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB  elbo={float(fx['fwd']['elbo']):.6f} "
          f"nll={float(fx['fwd']['nll']):.6f} kl={float(fx['fwd']['kl']):.6f}")


def op_vectors():
    gen = torch.Generator().manual_seed(5)
    fx = {}
    # gaussian KL / reparam incl. extreme logscales
    sh = (4, 6, 5, 5)
    q_loc, p_loc = torch.randn(sh, generator=gen), torch.randn(sh, generator=gen)
    q_ls = torch.randn(sh, generator=gen) * 1.5 - 1.0
    p_ls = torch.randn(sh, generator=gen) * 1.5 - 0.5
    q_ls[0, 0, 0, :] = torch.tensor([-9.0, -6.0, 0.0, 2.5, 4.0])
    p_ls[0, 0, 1, :] = torch.tensor([-9.0, -6.0, 0.0, 2.5, 4.0])
    fx["gaussian_kl"] = dict(q_loc=q_loc, q_logscale=q_ls, p_loc=p_loc, p_logscale=p_ls,
                             kl=ref_vae.gaussian_kl(q_loc, q_ls, p_loc, p_ls))
    # DGauss nll from (h) through the module, C=1 and C=3, pixels on the u8 grid incl. +-1 edge bins
    for C in (1, 3):
        hp = ohp.tiny_hparams(input_channels=C)
        a = ref_args(hp)
        torch.manual_seed(3 + C)
        lik = ref_vae.DGaussNet(a)
        with torch.no_grad():
            for p in lik.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.5)
        B, R = 4, 9
        h = torch.randn(B, hp.widths[0], R, R, generator=gen)
        x = (torch.randint(0, 256, (B, C, R, R), generator=gen).float() - 127.5) / 127.5
        x[0, :, 0, 0], x[0, :, 0, 1] = -1.0, 1.0
        h[1] *= 6.0  # drives logscale to the -9 clamp and the cdf clamps
        h.requires_grad_(True)
        nll = lik.nll(h, x)
        (gh,) = torch.autograd.grad(nll.sum(), h)
        with torch.no_grad():
            loc, ls = lik.forward(h, x)
            s_x, s_scale = lik.sample(h)
        fx[f"dgauss_c{C}"] = dict(state_dict={k: v.clone() for k, v in lik.state_dict().items()}, h=h.detach(), x=x,
                                   nll=nll.detach(), grad_h=gh, loc=loc, logscale=ls, sample_x=s_x, sample_scale=s_scale)
    # DMoL: loss, grad, soft/hard/top3 means
    B, R = 3, 7
    l = torch.randn(B, R, R, 100, generator=gen) * 1.5
    l[0, 0, 0, 10:] *= 6.0  # extreme params -> mid-bin fallback branch
    x = (torch.randint(0, 256, (B, R, R, 3), generator=gen).float() - 127.5) / 127.5
    x[0, 0, 1], x[0, 0, 2] = -1.0, 1.0
    l.requires_grad_(True)
    loss = ref_dmol.discretized_mix_logistic_loss(x, l)
    (gl,) = torch.autograd.grad(loss.sum(), l)
    d = dict(l=l.detach(), x=x, loss=loss.detach(), grad_l=gl)
    with torch.no_grad():
        for mask in ("soft", "hard", "top3"):
            mx, ms = ref_dmol.mean_discretized_mix_logistic(l.detach().clone(), 10, mask=mask, return_scale=True)
            d[f"mean_{mask}"], d[f"scale_{mask}"] = mx, ms
    fx["dmol"] = d
    path = os.path.join(OUT, "ops.pt")
    torch.save(fx, path)
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def anchors():
    """full-size presets, seed-7 default init + bias zeroing -> sum|theta| and (elbo,nll,kl)."""
    rows = {}
    for name in ("morphomnist", "cmnist", "ukbb192"):
        hp = ohp.make_hparams(name)
        a = ref_args(hp)
        torch.manual_seed(7)
        m = ref_vae.HVAE(a)

        def init_bias(mod):
            if type(mod) == torch.nn.Conv2d:
                torch.nn.init.zeros_(mod.bias)

        m.apply(init_bias)
        m.eval()
        g = torch.Generator().manual_seed(123)
        R, C = hp.input_res, hp.input_channels
        x = (torch.randint(0, 256, (2, C, R, R), generator=g).float() - 127.5) / 127.5
        pa = torch.randn(2, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, R, R)
        torch.manual_seed(11)
        with torch.no_grad():
            o = m(x, pa, beta=hp.beta)
        rows[name] = dict(abs_sum=float(sum(p.abs().double().sum() for p in m.parameters())),
                          n_params=sum(p.numel() for p in m.parameters()),
                          keys=list(m.state_dict().keys()),
                          shapes=[tuple(v.shape) for v in m.state_dict().values()],
                          elbo=float(o["elbo"]), nll=float(o["nll"]), kl=float(o["kl"]))
        print(name, {k: v for k, v in rows[name].items() if k not in ("keys", "shapes")})
    torch.save(rows, os.path.join(OUT, "anchors.pt"))


def train_steps():
    """(params0, grad sequence) -> params / EMA after k steps, using torch.optim.AdamW + LambdaLR and the
    reference's EMA class (utils.py imported with the absent ``imageio`` stubbed)."""
    sys.modules.setdefault("imageio", types.ModuleType("imageio"))
    import utils as ref_utils  # reference

    gen = torch.Generator().manual_seed(9)
    net = torch.nn.ParameterDict({"a": torch.nn.Parameter(torch.randn(5, 3, generator=gen)),
                                  "b": torch.nn.Parameter(torch.randn(7, generator=gen))})
    p0 = {k: v.detach().clone() for k, v in net.items()}
    hp = ohp.tiny_hparams(wd=0.05)
    opt = torch.optim.AdamW(net.parameters(), lr=hp.lr, weight_decay=hp.wd, betas=hp.betas)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=ref_utils.linear_warmup(hp.lr_warmup_steps))
    ema = ref_utils.EMA(net, beta=hp.ema_rate)
    n_steps = 150
    grads = {k: torch.randn(n_steps, *v.shape, generator=gen) for k, v in p0.items()}
    grads["a"][5] *= 100.0  # clipped: 350 <= norm < 500 (see norms)
    grads["a"][9] *= 1e4    # skipped (norm >= 500)
    snaps, norms = {}, []
    for s in range(n_steps):
        for k, p in net.items():
            p.grad = grads[k][s].clone()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), hp.grad_clip)
        norms.append(float(gn))
        if gn < hp.grad_skip:
            opt.step(); sch.step(); ema.update()
        if s + 1 in (1, 2, 10, 101, 102, 103, 150):
            snaps[s + 1] = dict(params={k: v.detach().clone() for k, v in net.items()},
                                ema={k: v.detach().clone() for k, v in ema.ema_model.items()},
                                lr=sch.get_last_lr()[0])
    torch.save(dict(p0=p0, grads=grads, norms=norms, snaps=snaps, hp=dict(vars(hp))), os.path.join(OUT, "train_steps.pt"))
    print("train_steps: norms[5]=%.1f norms[9]=%.1f" % (norms[5], norms[9]))


def simple_vae_fixture(B=6, seed=11, tag="c1", C=1, cond_prior=True, x_like="diag_dgauss"):
    """Config 1: the reference's ``simple_vae.VAE`` at the morphomnist preset with --cond_prior
    --context_dim 12 (234 690 parameters), parents [B,12] = two uniform(-1,1) scalars + one-hot(10).
    tag "c1x": the same with the exogenous N(0,I) prior (no --cond_prior); tag "c3": RGB input (independent channels); tag "dmol3": RGB with dmol.DmolNet; tag "gauss1": the logit-space GaussNet (dequantisation noise pinned)."""
    import simple_vae as ref_simple  # noqa: E402  (reference)

    gen = torch.Generator().manual_seed(seed)
    a = Hparams()
    a.update(dict(hps="morphomnist", input_res=32, input_channels=C, z_dim=16, context_dim=12, cond_prior=cond_prior,
                  widths=[16, 32, 64, 128, 256], x_like=x_like, std_init=0.0, kl_free_bits=0.0))
    torch.manual_seed(seed)
    m = ref_simple.VAE(a).eval()
    n_params = sum(p.numel() for p in m.parameters())
    init_abs_sum = float(sum(p.detach().double().abs().sum() for p in m.parameters()))  # default init under manual_seed(seed)
    with torch.no_grad():  # the prior heads are zero-initialised: perturb so every path is alive
        for n, p in m.named_parameters():
            if "prior.z_" in n or "likelihood.x_logscale" in n:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    x = (torch.randint(0, 256, (B, C, 32, 32), generator=gen).float() - 127.5) / 127.5
    x[0, 0, 0, 0], x[0, 0, 0, 1] = -1.0, 1.0

    def parents():
        sc = torch.rand(B, 2, generator=gen) * 2 - 1
        oh = torch.nn.functional.one_hot(torch.randint(0, 10, (B,), generator=gen), 10).float()
        return torch.cat([sc, oh], dim=1)

    pa, cf_pa = parents(), parents()
    eps = torch.randn(B, 16, generator=gen)
    orig = ref_simple.sample_gaussian
    ref_simple.sample_gaussian = lambda loc, ls: loc + ls.exp() * eps
    u_deq = torch.rand(B, C, 32, 32, generator=gen) if x_like.endswith("_gauss") else None
    orig_rand = torch.rand_like
    if u_deq is not None:  # GaussNet.nll dequantises with torch.rand_like(x) (simple_vae.py:222): pin it
        torch.rand_like = lambda t_, **k: u_deq.clone()
    try:
        for p in m.parameters():
            p.grad = None
        out = m(x, pa, beta=2.0)
        out["elbo"].backward()
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        fx = dict(hp=dict(hps="morphomnist", input_res=32, input_channels=C, z_dim=16, context_dim=12, cond_prior=cond_prior,
                          widths=[16, 32, 64, 128, 256], x_like=x_like, std_init=0.0, hidden_dim=128),
                  n_params=n_params, init_seed=seed, init_abs_sum=init_abs_sum, state_dict={k: v.detach().clone() for k, v in m.state_dict().items()},
                  x=x, pa=pa, cf_pa=cf_pa, eps=eps,
                  fwd=dict(beta=2.0, grads=grads, **{k: v.detach() for k, v in out.items()}))
        with torch.no_grad():
            # training-mode forward with a fixed conditioning-dropout draw (simple_vae.py:286-293)
            if cond_prior:
                m.train()
                m.decoder.drop_cond = lambda: (0, 1)
                o2 = m(x, pa, beta=1.0)
                fx["fwd_drop"] = dict(drop=(0, 1), **{k: v.detach() for k, v in o2.items()})
                m.eval()
                del m.decoder.drop_cond
            # 4-D parents are accepted too (takes [:, :, 0, 0]; simple_vae.py:64-65)
            o3 = m(x, pa[..., None, None].repeat(1, 1, 32, 32), beta=2.0)
            assert torch.allclose(o3["elbo"], out["elbo"])
            # abduct (t = 0.9) -> mediator z* -> forward_latents x2 -> dscm.py:55-56
            q = m.abduct(x, pa, t=0.9)[0]
            zstar = m.abduct(x, pa, cf_parents=cf_pa, alpha=0.3, t=0.9)[0]
            if not cond_prior:  # exogenous prior: abduct returns the latent itself, with or without cf_parents
                q = dict(z=q, q_loc=None, q_logscale=None)
            rec_loc, rec_scale = m.forward_latents([q["z"]], pa, t=0.9)
            cf_loc, cf_scale = m.forward_latents([zstar], cf_pa, t=0.9)
            u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
            cf_x = torch.clamp(cf_loc + cf_scale * u, min=-1, max=1)
            fx["abduct"] = dict(t=0.9, alpha=0.3, z=q["z"], q_loc=q["q_loc"], q_logscale=q["q_logscale"], zstar=zstar,
                                rec_loc=rec_loc, rec_scale=rec_scale, cf_loc=cf_loc, cf_scale=cf_scale, cf_x=cf_x)
            # prior sample with injected eps
            s_loc, s_scale = m.sample(pa, t=0.8)
            fx["sample"] = dict(t=0.8, x=s_loc, scale=s_scale)
    finally:
        ref_simple.sample_gaussian = orig
        torch.rand_like = orig_rand
    if u_deq is not None:
        fx["u"] = u_deq
    path = os.path.join(OUT, "simple_vae_%s.pt" % tag)
    torch.save(fx, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", n_params, "parameters")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "simple":
        simple_vae_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "simple_dmol3":
        simple_vae_fixture(seed=14, tag="dmol3", C=3, cond_prior=True, x_like="diag_dmol")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "simple_gauss1":
        simple_vae_fixture(seed=15, tag="gauss1", C=1, cond_prior=True, x_like="diag_gauss")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "simple_c3":
        simple_vae_fixture(seed=13, tag="c3", C=3, cond_prior=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "simple_c1x":
        simple_vae_fixture(seed=12, tag="c1x", C=1, cond_prior=False)
        sys.exit(0)
    T = ohp.tiny_hparams
    tiny_fixture("default_c1", T(hps="tiny"))
    tiny_fixture("default_c3", T(hps="tiny", input_channels=3, context_dim=5), seed=1)
    tiny_fixture("light_c1", T(hps="tiny_ukbb", z_max_res=8), seed=2)
    tiny_fixture("condprior_morpho_c1", T(hps="tiny_morphomnist", cond_prior=True, context_dim=4), seed=3)
    tiny_fixture("dmol_c3", T(hps="tiny", input_channels=3, context_dim=2), likelihood="dmol", seed=4)
    tiny_fixture("light_pad_c1", T(hps="tiny_ukbb", input_res=28, enc_arch="28b1d2,14b1d2,8b1d8,1b1",
                                   dec_arch="1b1,8b1,14b1,28b1", widths=[8, 8, 16, 32], z_max_res=14), seed=5)
    tiny_fixture("qcorr_c1", T(hps="tiny", q_correction=True), seed=6)
    op_vectors()
    anchors()
    train_steps()
    simple_vae_fixture()
    simple_vae_fixture(seed=12, tag="c1x", C=1, cond_prior=False)
    simple_vae_fixture(seed=13, tag="c3", C=3, cond_prior=True)
    simple_vae_fixture(seed=14, tag="dmol3", C=3, cond_prior=True, x_like="diag_dmol")
    simple_vae_fixture(seed=15, tag="gauss1", C=1, cond_prior=True, x_like="diag_gauss")
