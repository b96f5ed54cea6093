#!/usr/bin/env python3
"""Which engine call sites issue a given C-ABI entry point during one ukbb192 forward+backward (eager).
usage: python tools/launch_sites.py axpby [config]"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

name = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "ukbb192"
m, hp = bench.build_model(cfg, "f16")
m = m.cuda().train()
B = 32 if hp.input_res > 64 else 256
x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1)
out = m(x, pa, beta=1.0)
out["elbo"].backward()  # warm-up (arena sizing)
eng = m.engine()
orig = getattr(eng.lib, name)
sites = collections.Counter()


def spy(*a):
    fr = [f for f in traceback.extract_stack()[:-1] if "causal" in f.filename]
    sites[" <- ".join("%s:%d(%s)" % (os.path.basename(f.filename), f.lineno, f.name) for f in fr[-3:][::-1])] += 1
    return orig(*a)


setattr(eng.lib, name, spy)
m.zero_grad()
out = m(x, pa, beta=1.0)
out["elbo"].backward()
torch.cuda.synchronize()
for k, v in sites.most_common(20):
    print("%4d  %s" % (v, k))
