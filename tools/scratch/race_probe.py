#!/usr/bin/env python3
"""Does a conv give bit-identical results when another stream keeps the GPU busy?  (hunting timing-dependent results)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from causal_gen_amd.engine import ConvSite, Engine

SHAPES = [(8, 1, [512], 128, 3), (8, 1, [128], 512, 3), (8, 1, [512], 128, 1), (8, 1, [128], 544, 1), (8, 1, [512, 4, 512], 128, 1),
          (8, 6, [192], 48, 3), (8, 6, [48], 192, 3), (8, 12, [160], 40, 3), (8, 24, [128], 32, 3), (8, 24, [32], 128, 3), (32, 1, [512], 128, 3)]
convs = [torch.nn.Conv2d(sum(s[2]), s[3], s[4], padding=s[4] // 2) for s in SHAPES]
eng = Engine("cuda", "f16")
holder = torch.nn.ModuleList(convs).cuda()
sites = [ConvSite(f"c{i}", c, s[2], [True] * len(s[2]), i) for i, (c, s) in enumerate(zip(holder, SHAPES))]
eng.bind(holder, sites)
side = torch.cuda.Stream()
big = torch.randn(64 << 20, device="cuda")
for site, (N, R, segc, Co, ks) in zip(sites, SHAPES):
    eng.begin()
    eng.prepare_weights(force=True)
    xs = []
    g = torch.Generator().manual_seed(1)
    for c in segc:
        t = torch.randn(N, c, R, R, generator=g).cuda()
        xs.append(eng.from_nchw(t, rg=False) if hasattr(eng, "from_nchw") else None)
    y = eng.conv(site, xs, 1)
    torch.cuda.synchronize()
    nb = y.n * y.sn * y.es
    def snap():
        torch.cuda.synchronize()
        for ch in eng.arena.chunks:
            off = y.ptr - ch.data_ptr()
            if 0 <= off and off + nb <= ch.numel():
                return ch[off:off + nb].clone()
    ref = snap()
    bad = 0
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(3):
                big.mul_(1.0001)
        eng.conv(site, xs, 1, out=y)
        if not torch.equal(snap(), ref):
            bad += 1
    print("N%d res%d ci%s co%d ks%d: %d / 40 runs differ under a busy side stream" % (N, R, segc, Co, ks, bad))
