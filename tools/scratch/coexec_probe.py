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
from causal_gen_amd._lib import F16, NULL_VIEW, View

lib = _lib.require_gpu()
n, h, w, c = 8, 24, 24, 16
g = torch.Generator().manual_seed(0)


def t(scale=1.0, ch=c):
    return (torch.randn(n, h, w, ch, generator=g) * scale).to(torch.float16).cuda()


q = t(0.05, 32); pr = t(0.05, 32); z = t(1.0); gz = t(1e-7)
coef = torch.tensor([3.39e-6], device="cuda")


def view(x, c0, cc):
    return View(x.data_ptr() + 2 * c0, h * w * x.shape[3], w * x.shape[3], x.shape[3], cc, 0)


main = torch.cuda.current_stream()
side = torch.cuda.Stream()
A = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
Bm = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
big = torch.randn(32 << 20, device="cuda")


def run_reparam(out_q, out_p):
    lib.reparam_kl_bwd(F16, n, h, w, c, view(q, 0, 16), view(q, 16, 16), view(pr, 0, 16), view(pr, 16, 16), view(z, 0, 16), 0.0,
                       view(gz, 0, 16), coef.data_ptr(), 0, None, view(out_q, 0, 16), view(out_q, 16, 16), view(out_p, 0, 16),
                       view(out_p, 16, 16), 0, 0, main.cuda_stream)


def trial(name, neighbour, reps=300):
    outs = []
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        neighbour()
    for i in range(reps):
        oq = torch.zeros(n, h, w, 32, dtype=torch.float16, device="cuda")
        op = torch.zeros(n, h, w, 32, dtype=torch.float16, device="cuda")
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
eng = Engine("cuda", "f16")
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

# ---- conv kernels as victims next to the weight-gradient neighbour (address arithmetic, epilogue, MFMA of their own)
def conv_trial(name, N, R, ci, co, ks, reps=600):
    cv = torch.nn.ModuleList([torch.nn.Conv2d(ci, co, ks, padding=ks // 2)]).cuda()
    e2 = Engine("cuda", "f16")
    st = [ConvSite("v0", cv[0], [ci], [True], 0)]
    e2.bind(cv, st)
    e2.begin()
    e2.prepare_weights(force=True)
    xin = e2.new(N, R, R, ci)
    e2.fill(xin, 0.37)
    gg = torch.Generator().manual_seed(1)
    xt = torch.randn(N, ci, R, R, generator=gg).cuda()
    xin = e2.from_nchw(xt)
    outs = [e2.conv(st[0], [xin], 1) for _ in range(8)]
    torch.cuda.synchronize()
    nb = outs[0].n * outs[0].sn * outs[0].es

    def snap(y):
        for ch in e2.arena.chunks:
            off = y.ptr - ch.data_ptr()
            if 0 <= off and off + nb <= ch.numel():
                return ch[off:off + nb]
    ref = snap(outs[0]).clone()
    bad = 0
    neighbour = lambda: [lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl.value, 304, side.cuda_stream) for _ in range(3)]
    for i in range(reps):
        if i % 10 == 0:
            neighbour()
        y = outs[i % 8]
        e2.conv(st[0], [xin], 1, out=y)
        if i % 8 == 7:
            torch.cuda.synchronize()
            bad += sum(1 for o in outs if not torch.equal(snap(o), ref))
    print("%-28s: %d of %d conv executions differ next to the wgrad neighbour" % (name, bad, reps), flush=True)


conv_trial("px 32->128 3x3 @24x24 N32", 32, 24, 32, 128, 3)
conv_trial("ws 128->32 3x3 @24x24 N32", 32, 24, 128, 32, 3)
conv_trial("smallp 160->40 3x3 @12x12", 32, 12, 160, 40, 3)
