#!/usr/bin/env python3
"""Does the backward pass modify memory the forward pass wrote?  Snapshot the arena after forward, diff after backward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
m, hp = bench.build_model(cfg, "f16")
m = m.cuda().train()
B = 8 if hp.input_res > 64 else 64
x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1)
eng = m.engine()
for it in range(int(os.environ.get("N", "4"))):
    m.zero_grad()
    out = m(x, pa, beta=1.0)
    torch.cuda.synchronize()
    ci, off = eng.arena.ci, eng.arena.off
    snaps = [eng.arena.chunks[k][: (off if k == ci else eng.arena.chunks[k].numel())].clone() for k in range(ci + 1)]
    out["elbo"].backward()
    torch.cuda.synchronize()
    tot = 0
    for k, s in enumerate(snaps):
        cur = eng.arena.chunks[k][: s.numel()]
        d = (cur != s).nonzero().flatten()
        tot += d.numel()
        if d.numel():
            # contiguous runs
            dd = d.cpu()
            brk = (dd[1:] - dd[:-1] > 4096).nonzero().flatten() + 1
            starts = torch.cat([dd[:1], dd[brk]]); ends = torch.cat([dd[brk - 1], dd[-1:]])
            print("iter %d chunk %d: %d bytes of forward memory changed during backward, in %d regions; first regions (offset, span): %s" % (
                it, k, d.numel(), len(starts), [(int(a), int(b - a + 1)) for a, b in zip(starts[:6], ends[:6])]))
    if tot == 0:
        print("iter %d: forward memory untouched by the backward pass (%d bytes checked)" % (it, sum(s.numel() for s in snaps)))
