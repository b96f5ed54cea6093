import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
from conftest import load_golden
from oracle import fullsize_recipe as R
import test_gpu_fullsize as T
name, dmol = "ukbb192", False
row = load_golden("fullsize.pt")[R.key(name, dmol)]
B = [b for n, b, d in R.CASES if n == name][0]
x, pa = R.inputs(T._model(name, dmol, "f32")[1], B)
eps = R.eps_sequence(11, row["eps_shapes"])
outs = {}
for tag, dtype, small in (("f32", "f32", 1), ("f16_small", "f16", 1), ("f16_nosmall", "f16", 0)):
    m, hp = T._model(name, dmol, dtype)
    m.train()
    for p in m.parameters():
        p.requires_grad_(True)
    m.noise = [e.clone() for e in eps]
    eng = m.engine()
    eng.blk3_small = small
    ot = m(x.cuda(), pa.cuda(), beta=row["beta"])
    vals = {k: float(ot[k].detach()) for k in ("elbo", "nll", "kl")}
    # per-layer KL sums if the model keeps them
    kls = getattr(m, "_last_kl_per_layer", None)
    print(tag, vals)
    ot["elbo"].backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    outs[tag] = (vals, g)
    del m
    torch.cuda.empty_cache()
ref = outs["f32"][1]
for tag in ("f16_small", "f16_nosmall"):
    worst = []
    for n, gr in ref.items():
        d = (outs[tag][1][n] - gr).norm() / (gr.norm() + 1e-12)
        worst.append((float(d), n))
    worst.sort(reverse=True)
    print(tag, "worst relative L2 gradient errors vs f32:", [(round(a, 4), b) for a, b in worst[:6]])
