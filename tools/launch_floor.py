#!/usr/bin/env python3
"""What does a DEPENDENT kernel cost in a replayed hipGraph on this box, before it does any work?  A chain of N launches of
(a) the smallest torch kernel (fill of 1 element), (b) cgen_axpby on 64 elements, (c) cgen_axpby on 1 MB, captured in one
stream and replayed: microseconds per node."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from causal_gen_amd import _lib
from causal_gen_amd._lib import View, NULL_VIEW, F16

lib = _lib.require_gpu()
N = 2000
small = torch.zeros(64, dtype=torch.float16, device="cuda")
big = torch.zeros(512 * 1024, dtype=torch.float16, device="cuda")
one = torch.zeros(1, device="cuda")


def chain(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(s.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N):
                fn(s.cuda_stream)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 / N * 1e6


def v(t, c):
    n = t.numel() // c
    return View(t.data_ptr(), n * c, n * c, c, c, 0)


print("torch fill_(1 element)          : %.2f us per dependent node" % chain(lambda st: one.fill_(1.0)))
print("cgen_axpby in place, 64 elements : %.2f us per dependent node" % chain(lambda st: lib.axpby(F16, 1, 1, 8, v(small, 8), v(small, 8), 1.0, 1.0, 1 << 30, 0, st)))
print("cgen_axpby in place, 1 MB        : %.2f us per dependent node" % chain(lambda st: lib.axpby(F16, 1, 1, 512 * 128, v(big, 8), v(big, 8), 1.0, 1.0, 1 << 30, 0, st)))
