#!/usr/bin/env python3
"""Is reparam_kl_bwd stable while another kernel shares the CUs?  Neighbours: none / torch elementwise / torch bf16 matmul
(rocBLAS, MFMA) / the capped weight-gradient kernel.  Same inputs every time; outputs must be bit-identical."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from causal_gen_amd import _lib
from causal_gen_amd._lib import BF16, NULL_VIEW, View

lib = _lib.require_gpu()
n, h, w, c = 8, 24, 24, 16
g = torch.Generator().manual_seed(0)


def t(scale=1.0, ch=c):
    return (torch.randn(n, h, w, ch, generator=g) * scale).to(torch.bfloat16).cuda()


q = t(0.05, 32); pr = t(0.05, 32); z = t(1.0); gz = t(1e-7)
coef = torch.tensor([3.39e-6], device="cuda")


def view(x, c0, cc):
    return View(x.data_ptr() + 2 * c0, h * w * x.shape[3], w * x.shape[3], x.shape[3], cc, 0)


main = torch.cuda.current_stream()
side = torch.cuda.Stream()
A = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
Bm = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
big = torch.randn(32 << 20, device="cuda")


def run_reparam(out_q, out_p):
    lib.reparam_kl_bwd(BF16, n, h, w, c, view(q, 0, 16), view(q, 16, 16), view(pr, 0, 16), view(pr, 16, 16), view(z, 0, 16), 0.0,
                       view(gz, 0, 16), coef.data_ptr(), 0, None, view(out_q, 0, 16), view(out_q, 16, 16), view(out_p, 0, 16),
                       view(out_p, 16, 16), 0, 0, main.cuda_stream)


def trial(name, neighbour, reps=300):
    outs = []
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        neighbour()
    for i in range(reps):
        oq = torch.zeros(n, h, w, 32, dtype=torch.bfloat16, device="cuda")
        op = torch.zeros(n, h, w, 32, dtype=torch.bfloat16, device="cuda")
        run_reparam(oq, op)
        outs.append((oq, op))
        if i % 20 == 19:
            with torch.cuda.stream(side):
                neighbour()
    torch.cuda.synchronize()
    ref = outs[0]
    bad = sum(1 for o in outs if not (torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1])))
    print("%-28s: %d of %d executions differ from the first" % (name, bad, reps), flush=True)


trial("no neighbour", lambda: None)
trial("elementwise neighbour", lambda: [big.mul_(1.0001) for _ in range(20)])
trial("bf16 matmul neighbour", lambda: [torch.mm(A, Bm) for _ in range(6)])

# the weight-gradient kernel as the neighbour: a few 3x3 problems at 96x96, capped grid
from causal_gen_amd.engine import ConvSite, Engine
eng = Engine("cuda", "bf16")
convs = torch.nn.ModuleList([torch.nn.Conv2d(64, 16, 3, padding=1) for _ in range(6)]).cuda()
sites = [ConvSite(f"c{i}", cv, [64], [True], i) for i, cv in enumerate(convs)]
eng.bind(convs, sites)
eng.begin()
eng.prepare_weights(force=True)
xs = [eng.new(32, 96, 96, 64) for _ in sites]
gs = [eng.new(32, 96, 96, 16) for _ in sites]
for a in xs + gs:
    eng.fill(a, 0.25)
args = []
for site, x, gg in zip(sites, xs, gs):
    eng._wg_deferred = []
    eng.wgrad_flush_frac = []
    eng._wgrad(site, [x], 1, gg)
    args.append(eng._wg_deferred[0][0])
nargs = len(args)
arr = (_lib.WgradArgs * nargs)(*args)
nbytes, nl = C.c_int64(0), C.c_int32(0)
elig = (C.c_int32 * nargs)()
lib.conv2d_wgrad_batch_plan(arr, nargs, None, 0, C.byref(nbytes), None, 0, C.byref(nl), elig)
host = (C.c_char * nbytes.value)()
launches = (_lib.WgradBatchLaunch * max(nl.value, 1))()
lib.conv2d_wgrad_batch_plan(arr, nargs, host, nbytes.value, C.byref(nbytes), launches, nl.value, C.byref(nl), elig)
blob = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for cap in (304,):
    trial("wgrad neighbour, cap %d" % cap, lambda: [lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl.value, cap, side.cuda_stream) for _ in range(3)])
