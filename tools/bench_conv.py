#!/usr/bin/env python3
"""Micro-benchmark of single conv shapes through the C ABI (kernel-only time over many back-to-back launches).
usage: python tools/bench_conv.py [dtype] [kind] ; kind in fwd|wgrad|all"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from causal_gen_amd import _lib
from causal_gen_amd.engine import ConvSite, Engine

ONLY = os.environ.get("ONLY")
SHAPES = [  # (N, res, seg channels, Co, ks)   ukbb192 trunk shapes at batch 32
    (32, 192, [32], 8, 3), (32, 192, [8], 32, 3), (32, 96, [64], 16, 3), (32, 96, [16], 64, 3),
    (32, 48, [96], 24, 3), (32, 48, [24], 96, 3), (32, 24, [128], 32, 3), (32, 24, [32], 128, 3),
    (32, 12, [160], 40, 3), (32, 12, [40], 160, 3), (32, 6, [192], 48, 3), (32, 48, [96, 4, 96], 24, 3),
    (32, 48, [16, 96], 96, 1), (32, 96, [16, 4], 64, 1), (32, 24, [128, 4, 128], 32, 3), (32, 12, [160, 4, 160], 40, 3),
]


def main():
    global SHAPES
    if os.environ.get("NB"):
        SHAPES = [(int(os.environ["NB"]),) + sh[1:] for sh in SHAPES]
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
    kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    iters = int(os.environ.get("ITERS", "20"))
    convs = [torch.nn.Conv2d(sum(s[2]), s[3], s[4], padding=s[4] // 2) for s in SHAPES]
    eng = Engine("cuda", dtype)
    eng.wgrad_streams, eng.wgrad_batch = 0, False  # single launches are timed here
    holder = torch.nn.ModuleList(convs).cuda()
    sites = [ConvSite(f"c{i}", c, s[2], [True] * len(s[2]), i) for i, (c, s) in enumerate(zip(holder, SHAPES))]
    eng.bind(holder, sites)
    es = eng.es
    for idx, (site, (N, R, segc, Co, ks)) in enumerate(zip(sites, SHAPES)):
        if ONLY is not None and str(idx) not in ONLY.split(','):
            continue
        eng.begin()
        eng.prepare_weights(force=True)
        xs = [eng.new(N, R, R, c) for c in segc]
        for x in xs:
            eng.lib.philox_normal  # noqa
            if x.c % 8:  # ragged width: zero padding to 8 channels, as Engine.input does for the parents
                eng.fill(eng._padded(x), 0.0)
                x.cpad = (x.c + 7) // 8 * 8
                x._cv = None
            eng.fill(x, 0.5)
        torch.cuda.synchronize()
        flops = 2.0 * sum(segc) * ks * ks * Co * N * R * R
        bytes_alg = N * R * R * (sum(segc) + Co) * es
        res = {}
        def timed(fn):
            """GPU time per launch: the launches are captured in a hipGraph so the host is out of the picture."""
            fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.stream = torch.cuda.current_stream().cuda_stream
                for _ in range(iters):
                    fn()
            eng.stream = torch.cuda.current_stream().cuda_stream
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters

        if kind in ("fwd", "all"):
            y = eng.conv(site, xs, 1)
            res["fwd"] = timed(lambda: eng.conv(site, xs, 1, out=y))
            if os.environ.get("STAMPS"):
                st = torch.zeros(64, dtype=torch.int64, device="cuda")
                os.environ["CGEN_WS_STAMPS"] = hex(st.data_ptr())
                eng.conv(site, xs, 1, out=y)
                torch.cuda.synchronize()
                del os.environ["CGEN_WS_STAMPS"]
                v = st.cpu().tolist()
                n = v[63]
                d = [v[i + 1] - v[i] for i in range(n - 1)]
                if n:
                    print("   ws stamps(cycles) per tile [issue, wait, act, mfma, reduce+epilogue]: %s" % (
                        [d[k * 5:k * 5 + 5] for k in range((n - 1) // 5)]))
            if os.environ.get("PXSTAMPS"):
                # three back-to-back launches, each stamping {entry, tile landed, MFMAs done, exit} per workgroup (100 MHz clock)
                bufs = [torch.zeros(8 * 8192, dtype=torch.int64, device="cuda") for _ in range(3)]
                for b in bufs:
                    os.environ["CGEN_PX_STAMPS"] = hex(b.data_ptr())
                    eng.conv(site, xs, 1, out=y)
                del os.environ["CGEN_PX_STAMPS"]
                torch.cuda.synchronize()
                vs = [b.view(-1, 8).cpu() for b in bufs]
                vs = [v[v[:, 0] > 0] for v in vs]
                if len(vs[0]):
                    t0 = int(vs[1][:, 0].min())
                    us = lambda t: (float(t) - t0) / 100.0
                    v = vs[1]
                    print("   px stamps (us, launch 2 of 3; %d WGs): prev kernel last exit %.2f | first entry 0 last entry %.2f | landed median %.2f | "
                          "mfma done median %.2f | exit median %.2f last %.2f | next kernel first entry %.2f ; per-WG median: wait %.2f mfma %.2f epi %.2f"
                          % (len(v), us(vs[0][:, 3].max()), us(v[:, 0].max()), us(v[:, 1].median()), us(v[:, 2].median()), us(v[:, 3].median()),
                             us(v[:, 3].max()), us(vs[2][:, 0].min()), float((v[:, 1] - v[:, 0]).median()) / 100, float((v[:, 2] - v[:, 1]).median()) / 100,
                             float((v[:, 3] - v[:, 2]).median()) / 100))
                    md = lambda a, b: float((v[:, a] - v[:, b]).median()) / 100
                    if int(v[0, 5]) == 0:
                        print("      (smallp) entry->K loop %.2f K loop %.2f barrier %.2f reduce+epilogue %.2f" % (md(4, 0), md(1, 4), md(2, 1), md(3, 2)))
                    else:
                        print("      entry->weights issued %.2f ->lane consts %.2f ->tile+epilogue requests issued %.2f ->own DMAs landed+act %.2f ->barrier %.2f"
                              % (md(4, 0), md(5, 4), md(6, 5), md(7, 6), md(1, 7)))
        if kind in ("wgrad", "all"):
            g = eng.new(N, R, R, Co)
            eng.fill(g, 0.25)
            def wg():
                eng._wg_events = []
                eng._wgrad(site, xs, 1, g)

            res["wgrad"] = timed(wg)
            if os.environ.get("STAMPS"):
                st = torch.zeros(64, dtype=torch.int64, device="cuda")
                os.environ["CGEN_WG2_STAMPS"] = hex(st.data_ptr())
                wg()
                torch.cuda.synchronize()
                del os.environ["CGEN_WG2_STAMPS"]
                v = st.cpu().tolist()
                n = v[63]
                d = [v[i + 1] - v[i] for i in range(n - 1)]
                # stamps of workgroup 0: entry, setup done, then per tile [barrier (tile landed), next tile's DMA issued,
                # activation pass + barrier, MFMAs], ..., partial slab written
                print("   stamps(cycles): setup %d | per tile [wait+barrier, issue next, act, mfma]: %s | tail %d" % (
                    d[0], [d[1 + k * 4:1 + k * 4 + 4] for k in range(min(5, (n - 3) // 4))], d[-1]))
        for k, us in res.items():
            print("%-5s %s N%d res%-3d ci%-12s co%-3d ks%d : %8.1f us  %7.1f TF/s  %6.2f TB/s(alg)" % (
                k, dtype, N, R, str(segc), Co, ks, us, flops / us / 1e6, bytes_alg / us / 1e6))


if __name__ == "__main__":
    main()
