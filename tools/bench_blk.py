"""Per-shape timing of the fused light-Block kernel (cgen_block3) against the two-launch path (forward and data gradient),
batch 32.  usage: python tools/bench_blk.py [reps]   (BLK_ONLY=<res> one resolution)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_gen_amd.engine import ConvSite, Engine

SHAPES = [(32, 192, [32], 8, 32), (32, 192, [32], 8, 64), (32, 96, [64], 16, 64), (32, 96, [64], 16, 96), (32, 96, [64, 4, 64], 16, 32),
          (32, 48, [96], 24, 96), (32, 48, [96], 24, 128), (32, 48, [96, 4, 96], 24, 32), (32, 24, [128], 32, 128), (32, 24, [128], 32, 160),
          (32, 24, [128, 4, 128], 32, 32)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = os.environ.get("BLK_ONLY")
for (N, R, segc, b, co) in SHAPES:
    if only and str(R) != only:
        continue
    row = []
    for fuse in (0, 2):
        ci = sum(segc)
        c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, co, 3, padding=1)
        eng = Engine("cuda", "f16")
        eng.blk3_on, eng.blk3_minres = 2, 8
        eng.blk3_res, eng.blk3_res3 = [], []
        eng.wgrad_flush_frac = []
        holder = torch.nn.ModuleList([c1, c2]).cuda()
        rgs = [c >= 8 for c in segc]  # (the narrow segment stands for the parents: no gradient)
        s1, s2 = ConvSite("c1", holder[0], segc, rgs, 0), ConvSite("c2", holder[1], [b], [True], 1)
        s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
        eng.bind(holder, [s1, s2])
        eng.blk3_on = fuse
        for p in holder.parameters():
            p.requires_grad_(False)  # time the data path only
        xs = [torch.randn(N, c, R, R).cuda() for c in segc]
        res = torch.randn(N, co, R, R).cuda() if co == ci else None
        gout = torch.randn(N, co, R, R).cuda()
        tf, tb = [], []
        for it in range(reps + 3):
            eng.begin(); eng.prepare_weights(force=(it == 0)); eng.recording = True
            xts = [eng.from_nchw(x, rg=r) for x, r in zip(xs, rgs)]
            rt = eng.from_nchw(res) if res is not None else None
            go = eng.from_nchw(gout)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            y = eng.block2(s1, s2, xts, 1, res1=rt)
            e[1].record()
            gy = eng.seed_grad(y)
            eng.lib.axpby(eng.dt, N, R, R, go.cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
            eng.recording = False
            e[2].record()
            eng.backward()
            e[3].record()
            torch.cuda.synchronize()
            if it >= 3:
                tf.append(e[0].elapsed_time(e[1])); tb.append(e[2].elapsed_time(e[3]))
        row.append((min(tf) * 1e3, min(tb) * 1e3))
    fl = 2.0 * 9 * (sum(segc) * b + b * co) * N * R * R
    print("res %3d  %s->%d->%d : fwd two-launch %7.1f us fused %7.1f us | dgrad two-launch %7.1f us fused %7.1f us | fused fwd %.0f TF/s" % (
        R, segc, b, co, row[0][0], row[1][0], row[0][1], row[1][1], fl / (row[1][0] * 1e-6) / 1e12), flush=True)
