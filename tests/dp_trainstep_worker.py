"""Worker of test_dp_two_ranks_reproduce_the_single_process_step (launched twice by torch.distributed.run, gloo, both ranks on
cuda:0).  Mean-of-gradients == gradient-of-the-mean on the REAL path, in two halves (a captured step cannot take injected noise,
and the in-kernel Philox draws are indexed by the LOCAL sample number):
  (a) default graph-captured TrainStep, Philox noise: both ranks hold the SAME two samples, so the rank-averaged gradient must be
      the gradient of that half batch -- three optimiser steps against a single process on those two samples (catches a wrong
      averaging factor, a lost scalar exchange, a broken graph split);
  (b) eager TrainStep, injected per-sample eps: the ranks hold DIFFERENT halves of a batch of four -- three steps against a single
      process on all four samples (catches a rank whose gradient is dropped or double-counted)."""
import os
import sys
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def run(idx, pg, use_graph, inject, steps=3):
    from causal_gen_amd.train import TrainStep
    from test_gpu_train import setup

    fx, hpd, m = setup()
    m.compute_dtype = "f32"
    m.train()
    torch.manual_seed(1234)  # the engine's Philox seed
    ts = TrainStep(m, SimpleNamespace(**hpd), ema=False, use_graph=use_graph, process_group=pg)
    x, pa = fx["x"][idx].cuda(), fx["pa"][idx].cuda()
    outs = []
    for _ in range(steps):
        if inject:
            m.noise = [e[idx].clone() for e in fx["fwd"]["eps"]]
        outs.append([float(v) for v in ts.step(x, pa).cpu()])
    torch.cuda.synchronize()
    return outs, {k: v.detach().clone() for k, v in m.state_dict().items()}


def compare(tag, got, ref, ptol, stol):
    (out, sd), (ref_out, ref_sd) = got, ref
    worst = 0.0
    for k, v in ref_sd.items():
        d = float((sd[k] - v).abs().max()) / (float(v.abs().max()) + 1e-12)
        worst = max(worst, d)
        assert d <= ptol, (tag, k, d)
    for a, b in zip(out, ref_out):
        for u, v in zip(a, b):
            assert abs(u - v) <= stol * abs(v) + 1e-7, (tag, out, ref_out)
    return worst


def main():
    half, full = torch.tensor([0, 1]), torch.tensor([0, 1, 2, 0])
    ref_a = run(half, None, True, False)
    ref_b = run(full, None, False, True)
    torch.distributed.init_process_group("gloo")
    r, w = torch.distributed.get_rank(), torch.distributed.get_world_size()
    pg = torch.distributed.group.WORLD
    wa = compare("graph / same halves", run(half, pg, True, False), ref_a, 1e-6, 1e-6)
    wb = compare("eager / different halves", run(full[r * 2:r * 2 + 2], pg, False, True), ref_b, 2e-4, 2e-5)
    torch.distributed.barrier()
    if r == 0:
        print("DP_TRAINSTEP_OK worst param err: graph %.2e, eager %.2e" % (wa, wb))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
