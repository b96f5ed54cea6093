#!/usr/bin/env python3
"""Full-size model through the f16 throughput path and the f32 parity path: ELBO and per-parameter gradient agreement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
res = {}
for dt in ("f32", "f16"):
    m, hp = bench.build_model(cfg, dt)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
    x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1)
    eng = m.engine()
    eng.rng_ptr()
    eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
    out = m(x, pa, beta=1.0)
    out["elbo"].backward()
    torch.cuda.synchronize()
    res[dt] = ({k: float(out[k]) for k in ("elbo", "nll", "kl")}, {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None})
    if dt == "f16":  # where do the loss-scaled activation gradients sit in binary16's range?
        mx, n_sub, n_zero, n_all = 0.0, 0, 0, 0
        for g, _, _ in eng.grads.values():
            v = eng.to_nchw(g).abs()
            mx = max(mx, float(v.max()))
            n_all += v.numel()
            n_zero += int((v == 0).sum())
            n_sub += int(((v > 0) & (v < 6.1e-5)).sum())
        print("f16 activation gradients: loss scale 2^%d, largest |g| %.3g (overflow at 65504), %.2f %% of the non-zero elements subnormal (< 6.1e-5), "
              "%.1f %% exactly zero (ReLU masks included)" % (round(__import__("math").log2(eng.loss_scale)), mx, 100.0 * n_sub / max(1, n_all - n_zero), 100.0 * n_zero / n_all))
    del m
print("f32 ", res["f32"][0])
print("f16", res["f16"][0])
errs = []
for n, gf in res["f32"][1].items():
    gb = res["f16"][1][n]
    den = float(gf.norm())
    if den == 0:
        continue
    errs.append((float((gb - gf).norm()) / den, float((gb * gf).sum()) / (den * float(gb.norm()) + 1e-30), n, gf.numel()))
errs.sort(reverse=True)
print("params %d; relative L2 error of the f16 gradient: max %.4f median %.4f; min cosine %.5f" % (len(errs), errs[0][0], errs[len(errs) // 2][0], min(e[1] for e in errs)))
for e in errs[:8]:
    print("   %.4f cos %.5f %s (%d)" % e)
