"""Debug: where does the counterfactual branch's gradient stop?  (tiny_default_c1, cf-only loss)"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import load_golden
from causal_gen_amd import vae
from causal_gen_amd.hps import Hparams
from oracle import dscm_ref, hvae_ref

fx = load_golden("tiny_default_c1.pt")
hpd = dict(fx["hp"])
m = vae.HVAE(Hparams(**hpd)); m.load_state_dict(fx["state_dict"]); m = m.cuda().eval()
x, pa, cf = fx["x"], fx["pa"], fx["cf_pa"]
sd = {k: v.detach().clone().requires_grad_(True) for k, v in fx["state_dict"].items()}
noise = hvae_ref._Noise(None)
torch.manual_seed(3)
ref = dscm_ref.dscm_forward(sd, SimpleNamespace(**hpd), x, pa, [cf], 1.0, noise=noise)
(ref["cf_x"] ** 2).sum().backward()
m.noise = [e.clone() for e in noise.drawn]
eng = m.engine()
orig = eng._bw_conv
def traced(site, segs, act, out, res1, res2):
    g = eng.grads.get(id(out.base))
    print("bw_conv %-40s out.base=%x has_grad=%s ivs=%s" % (site.name, id(out.base), g is not None, None if g is None else g[1]))
    return orig(site, segs, act, out, res1, res2)
trig = torch.zeros(1, device="cuda", requires_grad=True)
elbo, nll, kl, cf_x, _ = vae._DSCMFunction.apply(trig, m, x.cuda(), pa.cuda(), (cf.cuda(),), 1.0, 1.0)
print("tape entries", len(eng.tape), "cf_x err", float((cf_x.detach().cpu() - ref["cf_x"].detach()).abs().max()))
eng.tape = [((traced if fn == orig else fn), a) for fn, a in eng.tape]
(cf_x ** 2).sum().backward()
torch.cuda.synchronize()
passes, xin = m.__dict__["_saved_cf"]
for nm, t in zip(("rec", "cf"), passes[0]):
    e = eng.grads.get(id(t.base))
    print(nm, "seed grad entry", None if e is None else (e[1],))
for n_, p in m.named_parameters():
    rg = sd[n_].grad
    if rg is None: continue
    got = p.grad
    print("%-45s ref %.3e got %s" % (n_, float(rg.norm()), "None" if got is None else "%.3e err %.2e" % (float(got.norm()), float((got.cpu()-rg).abs().max()/(rg.abs().max()+1e-30)))))
