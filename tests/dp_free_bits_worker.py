"""Worker of test_free_bits_under_data_parallelism (launched twice by torch.distributed.run, gloo, both ranks on cuda:0).
Each rank computes the single-process result on the full batch first, then its half of the batch with the cross-rank
exchange of the per-channel KL sums, and the averaged gradients must equal the full-batch gradients."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_golden  # noqa: E402
from test_gpu_model import build  # noqa: E402


def run(fx, fb, x, pa, eps):
    m, _ = build(fx)
    m.free_bits = fb
    m.noise = [e.clone() for e in eps]
    out = m(x.cuda(), pa.cuda(), beta=1.0)
    out["elbo"].backward()
    torch.cuda.synchronize()
    return {k: float(out[k]) for k in ("elbo", "nll", "kl")}, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def main():
    fx = load_golden("tiny_condprior_morpho_c1.pt")
    d = fx["fwd_freebits"]
    idx = torch.tensor([0, 1, 2, 0])
    x, pa, eps = fx["x"][idx], fx["pa"][idx], [e[idx] for e in d["eps"]]
    ref, gref = run(fx, d["free_bits"], x, pa, eps)  # no process group yet: the single-process path
    torch.distributed.init_process_group("gloo")
    r, w = torch.distributed.get_rank(), torch.distributed.get_world_size()
    sl = slice(r * 2, r * 2 + 2)
    out, g = run(fx, d["free_bits"], x[sl], pa[sl], [e[sl] for e in eps])
    nll = torch.tensor([out["nll"]], dtype=torch.float64)
    torch.distributed.all_reduce(nll)
    assert abs(out["kl"] - ref["kl"]) <= 1e-5 * abs(ref["kl"]), (out, ref)  # the floored KL is a global-batch statistic
    assert abs(float(nll) / w - ref["nll"]) <= 1e-5 * abs(ref["nll"]), (float(nll) / w, ref)
    worst = 0.0
    for n, gr in gref.items():
        gg = g[n].clone()
        torch.distributed.all_reduce(gg)
        gg /= w
        if float(gr.abs().max()) == 0:
            continue
        err = float((gg - gr).abs().max()) / float(gr.abs().max())
        worst = max(worst, err)
        assert err < 2e-4, (n, err)
    torch.distributed.barrier()
    if r == 0:
        print("DP_FREE_BITS_OK worst grad err %.2e" % worst)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
