"""The fused default Block (cgen_block4) against the four launches it replaces, per Block shape of mimic224 / the MNIST presets:
forward and data gradient (weight gradients ablated), back-to-back launches timed with events.
    python tools/bench_blk4.py [reps] [shape indices, comma separated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_gen_amd import _lib
from causal_gen_amd.engine import ConvSite, Engine

# (N, R, segments, differentiable, bottleneck, out, residual)
SHAPES = [
    (32, 224, [32], [1], 8, 32, True),
    (32, 112, [64], [1], 16, 64, True),
    (32, 112, [64, 6, 64], [1, 0, 1], 16, 32, False),
    (32, 56, [96], [1], 24, 96, True),
    (32, 56, [96, 6, 96], [1, 0, 1], 24, 32, False),
    (32, 28, [128], [1], 32, 128, True),
    (32, 28, [128], [1], 32, 160, False),
    (32, 14, [160], [1], 40, 160, True),
    (32, 8, [192], [1], 48, 192, True),
    (256, 32, [16], [1], 4, 16, True),
    (256, 16, [32], [1], 8, 32, True),
    (256, 8, [64], [1], 16, 64, True),
    (256, 4, [128], [1], 32, 128, True),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else range(len(SHAPES))


def run(shape, fuse):
    N, R, segc, segrg, b, co, with_res = shape
    ci = sum(segc)
    cs = [torch.nn.Conv2d(ci, b, 1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, co, 1)]
    eng = Engine("cuda", "f16")
    eng.wgrad_flush_frac = []
    eng._ablate = "wg"
    holder = torch.nn.ModuleList(cs).cuda()
    sites = [ConvSite("c0", holder[0], segc, [bool(r) for r in segrg], 0)] + [ConvSite(f"c{k}", holder[k], [b], [True], k) for k in (1, 2, 3)]
    for r in range(4):
        sites[r].blk4 = (r, sites)
    eng.blk4_on = 1
    eng.bind(holder, sites)
    eng.blk4_on = fuse
    xs = [torch.randn(N, c, R, R).cuda() for c in segc]
    resid = torch.randn(N, co, R, R).cuda() if with_res else None
    gout = torch.randn(N, co, R, R).cuda()

    def fwd():
        y = eng.block4(sites, xts, res1=rt) if fuse else None
        if y is None:
            h = eng.conv(sites[0], xts, _lib.ACT_GELU)
            h = eng.conv(sites[1], [h], _lib.ACT_GELU)
            h = eng.conv(sites[2], [h], _lib.ACT_GELU)
            y = eng.conv(sites[3], [h], _lib.ACT_GELU, res1=rt)
        return y

    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # forward alone
    eng.begin(); eng.prepare_weights(force=True)
    xts = [eng.from_nchw(x, rg=bool(r)) for x, r in zip(xs, segrg)]
    rt = eng.from_nchw(resid) if resid is not None else None
    for it in range(reps):  # (also grows the arena to its final size: chunk allocations must not land in the timed loop)
        fwd()
    torch.cuda.synchronize()
    eng.begin()
    xts = [eng.from_nchw(x, rg=bool(r)) for x, r in zip(xs, segrg)]
    rt = eng.from_nchw(resid) if resid is not None else None
    torch.cuda.synchronize()
    e0.record()
    for it in range(reps):
        fwd()
    e1.record()
    torch.cuda.synchronize()
    out["fwd"] = 1e3 * e0.elapsed_time(e1) / reps
    # backward: record `reps` independent Blocks, then one backward pass over the tape (twice: the first pass sizes the arena)
    for warm in (1, 0):
      eng.begin()
      xts = [eng.from_nchw(x, rg=bool(r)) for x, r in zip(xs, segrg)]
      rt = eng.from_nchw(resid) if resid is not None else None
      gt = eng.from_nchw(gout)
      eng.recording = True
      ys = [fwd() for it in range(reps)]
      for y in ys:
          gy = eng.seed_grad(y)
          eng.lib.axpby(eng.dt, N, R, R, gt.cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
      eng.recording = False
      torch.cuda.synchronize()
      e0.record()
      eng.backward()
      e1.record()
      torch.cuda.synchronize()
      out["bwd"] = 1e3 * e0.elapsed_time(e1) / reps
    return out


only = os.environ.get("B4_ONLY")  # "0" / "1": run only the four-launch / only the fused path (for counter passes: one kernel family per run)
for i in sel:
    s = SHAPES[i]
    if only is not None:
        r = run(s, int(only))
        print(i, s, "fused" if int(only) else "four launches", r, flush=True)
        continue
    a, b_ = run(s, 0), run(s, 1)
    N, R, segc, _, b, co, wr = s
    gb = N * R * R * 2 * (sum(segc) + co * (2 if wr else 1) + 3 * b) / 1e9  # forward algorithmic bytes (inputs, output, residual, three bottlenecks)
    print("%2d  n %3d  %3dx%-3d %-14s b %2d -> %3d  | fwd  four %7.1f us  fused %7.1f us (%.2fx, %.0f GB/s alg.) | bwd  four %7.1f us  fused %7.1f us (%.2fx)" % (
        i, N, R, R, segc, b, co, a["fwd"], b_["fwd"], a["fwd"] / b_["fwd"], gb / (b_["fwd"] * 1e-6), a["bwd"], b_["bwd"], a["bwd"] / b_["bwd"]), flush=True)
