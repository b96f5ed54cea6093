"""Debug: ukbb192 f32 decoder.bias.4 gradient vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import fullsize_recipe as R, hparams as ohp, hvae_ref

name, B = "ukbb192", 2
m, hp = bench.build_model(name, "f32"); R.perturb(m); m = m.cuda().eval()
x, pa = R.inputs(hp, B)
shapes = [(B, b.z_dim, b.res, b.res) for b in m.decoder.blocks if b.stochastic]
eps = R.eps_sequence(11, shapes)
for p in m.parameters(): p.requires_grad_(True)
for trial in range(2):
    if trial == 1:
        os.environ["CGEN_WGRAD_FLUSH_FRAC"] = ""
        m.__dict__["_eng"] = None
        for p in m.parameters(): p.grad = None
    m.noise = [e.clone() for e in eps]
    out = m(x.cuda(), pa.cuda(), beta=5.0); out["elbo"].backward(); torch.cuda.synchronize()
    if trial == 0:
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
        ref = hvae_ref.hvae_forward(sd, ohp.make_hparams(name), x, pa, beta=5.0, noise=hvae_ref._Noise([e.clone() for e in eps]))
        ref["elbo"].backward()
    for i in range(5):
        n_ = "decoder.bias.%d" % i
        g, rg = dict(m.named_parameters())[n_].grad.cpu(), sd[n_].grad
        d = (g - rg).abs()
        bad = (d > 2e-3 * rg.abs().max()).nonzero()
        print("trial", trial, n_, tuple(g.shape), "max err rel", float(d.max() / rg.abs().max()), "n bad", bad.shape[0],
              "first bad idx", bad[:6].tolist(), "ratio at worst", float(g.flatten()[d.argmax()] / rg.flatten()[d.argmax()]))
        if bad.shape[0]:
            ys = bad[:, 2].unique().tolist(); xs = bad[:, 3].unique().tolist(); cs = bad[:, 1].unique().tolist()
            print("   bad rows", ys[:30], "cols", xs[:30], "chans", cs[:30])
