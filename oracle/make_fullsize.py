"""Full-size golden anchors with LIVE weights, made by running the REFERENCE itself (/root/reference/src).

    python3 -B oracle/make_fullsize.py        (build container only; writes tests/golden/fullsize.pt)

anchors.pt pins the default init, where the prior heads are x0 and KL ~ 5e-3 (SURVEY probe C.7).  Here every BASELINE
preset (morphomnist, cmnist + DmolNet, ukbb192, mimic-shape 224^2) is perturbed by a SEEDED recipe -- nothing but the
recipe, the inputs' seeds and the reference's outputs is stored (17 M weights would be 70 MB) -- so a test can rebuild
the same parameters anywhere:

    torch.manual_seed(7); m = HVAE(args) [; m.likelihood = DmolNet(args)]; m.apply(init_bias); perturb(m, seed 5)

Stored per preset: (elbo, nll, kl) at beta = preset beta with the eps sequence of torch.manual_seed(11) (shapes listed so
the test regenerates it), a strided sample + norm of the gradient of a dozen named parameters, and a strided sample of the
counterfactual pixels (abduct at torch.manual_seed(21) -> two replays -> dscm.py:55-56) under rolled parents.
"""
import math
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dmol as ref_dmol  # noqa: E402  (reference)
import vae as ref_vae  # noqa: E402  (reference)
from hps import Hparams  # noqa: E402  (reference)

from oracle import fullsize_recipe as R  # noqa: E402
from oracle import hparams as ohp  # noqa: E402


class EpsTap:
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


def main():
    rows = {}
    for name, B, dmol in R.CASES:
        hp = ohp.make_hparams(name)
        a = Hparams()
        a.update(dict(vars(hp)))
        torch.manual_seed(7)
        m = ref_vae.HVAE(a)
        if dmol:
            m.likelihood = ref_dmol.DmolNet(a)
        m.apply(R.init_bias)
        R.perturb(m)
        m.eval()
        x, pa = R.inputs(hp, B)
        torch.manual_seed(11)
        with EpsTap() as tap:
            out = m(x, pa, beta=hp.beta)
        out["elbo"].backward()
        shapes = [tuple(e.shape) for e in tap.eps]
        regen = R.eps_sequence(11, shapes)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(regen, tap.eps)), "eps recipe does not reproduce randn_like"
        named = dict(m.named_parameters())
        grads = {}
        for n in R.grad_names(list(named)):
            g = named[n].grad
            if g is None:  # (the last decoder block's z_feat_proj is never used, vae.py:299-300)
                continue
            grads[n] = dict(norm=float(g.double().norm()), sample=R.sample(g).clone())
        row = dict(B=B, dmol=dmol, beta=float(hp.beta), elbo=float(out["elbo"]), nll=float(out["nll"]), kl=float(out["kl"]),
                   eps_shapes=shapes, grads=grads, abs_sum=float(sum(p.detach().abs().double().sum() for p in m.parameters())))
        with torch.no_grad():
            cf_pa = pa.roll(1, 0) if B > 1 else pa.flip(1)
            torch.manual_seed(21)
            with EpsTap() as tap2:
                zs = m.abduct(x, pa, t=1.0)
            if hp.cond_prior:
                zs = [z["z"] for z in zs]
            shapes2 = [tuple(e.shape) for e in tap2.eps]
            rec_loc, rec_scale = m.forward_latents(zs, pa)
            cf_loc, cf_scale = m.forward_latents(zs, cf_pa)
            u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
            cf_x = torch.clamp(cf_loc + cf_scale * u, min=-1, max=1)
            row["cf"] = dict(eps_shapes=shapes2, cf_x=R.sample_img(cf_x).clone(), rec_loc=R.sample_img(rec_loc).clone(),
                             rec_scale=R.sample_img(rec_scale).clone(), cf_loc=R.sample_img(cf_loc).clone(),
                             moved=float((cf_x - x).abs().mean()))
        rows[R.key(name, dmol)] = row
        print(R.key(name, dmol), {k: v for k, v in row.items() if k in ("B", "elbo", "nll", "kl", "abs_sum")}, "cf moved", row["cf"]["moved"],
              "grad norms", {n.split(".", 2)[-1][-28:]: round(g["norm"], 5) for n, g in list(grads.items())[:4]}, flush=True)
    path = os.path.join(ROOT, "tests", "golden", "fullsize.pt")
    torch.save(rows, path)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
