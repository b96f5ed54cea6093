import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
from conftest import load_golden
from oracle import fullsize_recipe as R
import test_gpu_fullsize as T
name, dmol = "ukbb192", False
row = load_golden("fullsize.pt")[R.key(name, dmol)]
B = [b for n, b, d in R.CASES if n == name][0]
mb, hp = T._model(name, dmol, "f16")
x, pa = R.inputs(hp, B)
eps = R.eps_sequence(11, row["eps_shapes"])
mb.train()
for p in mb.parameters():
    p.requires_grad_(True)
for it in range(3):
    mb.noise = [e.clone() for e in eps]
    ot = mb(x.cuda(), pa.cuda(), beta=row["beta"])
    vals = {k: float(ot[k].detach()) for k in ("elbo", "nll", "kl")}
    print(it, vals, "elbo - (nll + beta kl) =", vals["elbo"] - (vals["nll"] + row["beta"] * vals["kl"]), "ref", row["elbo"], row["nll"], row["kl"])
    ot["elbo"].backward()
    torch.cuda.synchronize()
