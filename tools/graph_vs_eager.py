#!/usr/bin/env python3
"""ukbb192 (or another preset) bf16: N optimiser steps eager vs hipGraph replay -> parameters must be bit-identical."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from causal_gen_amd.train import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
outs = []
for use_graph in (False, True):
    m, hp = bench.build_model(cfg, "f16")
    m = m.cuda()
    torch.manual_seed(123)
    ts = TrainStep(m, hp, ema=True, use_graph=use_graph)
    x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1)
    for _ in range(4):
        o = ts.step(x, pa)
    torch.cuda.synchronize()
    outs.append(([float(v) for v in o.cpu()], {k: v.detach().clone() for k, v in m.state_dict().items()}, ts.stats()))
(o0, s0, t0), (o1, s1, t1) = outs
bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
print("eager", o0, t0["opt_steps"], "| graph", o1, t1["opt_steps"], "| differing tensors %d of %d %s" % (len(bad), len(s0), bad[:3]))
sys.exit(1 if bad or o0 != o1 else 0)
