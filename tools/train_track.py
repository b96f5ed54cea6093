#!/usr/bin/env python3
"""Does loss-scaled f16 TRAINING track the f32 parity path?  Two models from one init take the same sequence of synthetic
batches (a fresh batch every step, fixed generator) through `TrainStep` (AdamW + warm-up + clip / skip + EMA, hipGraph
replay), one in f16 and one in f32; prints both ELBO curves, their relative distance, skipped steps and the final
distance of the parameters.

usage: tools/train_track.py [config] [batch] [steps]      (default: ukbb192 8 300)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from causal_gen_amd.train import TrainStep

cfg = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
every = max(1, steps // 15)

runs = {}
for dt in ("f32", "f16"):
    torch.manual_seed(0)
    m, hp = bench.build_model(cfg, dt)
    m = m.cuda().train()
    if getattr(m, "cond_prior", False):
        m.decoder.__dict__["drop_cond"] = lambda: (1, 1)
    ts = TrainStep(m, hp, ema=True, use_graph=True)
    eng = m.engine()
    eng.rng_ptr()
    eng.rng.copy_(torch.tensor([77, 0], dtype=torch.int64, device=eng.rng.device))  # same Philox noise in both runs
    curve = []
    for it in range(steps):
        x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1000 + it)
        out3 = ts.step(x, pa)
        if it % every == 0 or it == steps - 1:
            curve.append((it, [float(v) for v in out3.tolist()]))
    torch.cuda.synchronize()
    st = ts.stats()
    runs[dt] = (curve, st, torch.cat([p.detach().flatten().float() for p in m.parameters()]).cpu(), getattr(eng, "loss_scale", 1.0))
    del ts, m
    torch.cuda.empty_cache()

c32, c16 = runs["f32"][0], runs["f16"][0]
print("%s B=%d, %d optimiser steps, fresh synthetic batch per step; f16 loss scale 2^%d" % (cfg, B, steps, round(__import__("math").log2(runs["f16"][3]))))
print(" step |   elbo f32    elbo f16   rel diff |    kl f32      kl f16")
worst = 0.0
for (it, a), (_, b) in zip(c32, c16):
    rel = abs(a[0] - b[0]) / abs(a[0])
    worst = max(worst, rel)
    print("%5d | %10.5f  %10.5f  %.2e | %10.3e  %10.3e" % (it, a[0], b[0], rel, a[2], b[2]))
p32, p16 = runs["f32"][2], runs["f16"][2]
print("largest ELBO distance along the run %.2e; parameters after %d steps: relative L2 distance %.3e; skipped steps f32 %d / f16 %d; "
      "last grad norm f32 %.3f / f16 %.3f" % (worst, steps, float((p32 - p16).norm() / p32.norm()), runs["f32"][1]["n_skipped"],
                                             runs["f16"][1]["n_skipped"], runs["f32"][1]["grad_norm"], runs["f16"][1]["grad_norm"]))
