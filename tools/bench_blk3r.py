"""Forward + data-gradient launches of one light Block at the wide resolutions (row-streaming instance blk3r vs the two-launch path),
for rocprofv3 --kernel-trace --stats:    python tools/bench_blk3r.py [192|1922|96|962] [reps]     (FUSE=0: two launches per conv pair)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_gen_amd.engine import ConvSite, Engine

key = int(sys.argv[1]) if len(sys.argv) > 1 else 192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, R, segc, b, co = {192: (32, 192, [32], 8, 32), 1922: (32, 192, [32], 8, 64), 96: (32, 96, [64], 16, 64), 962: (32, 96, [64], 16, 32)}[key]
ci = sum(segc)
c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, co, 3, padding=1)
eng = Engine("cuda", "f16")
eng.blk3_on = 2
eng.wgrad_flush_frac = []
holder = torch.nn.ModuleList([c1, c2]).cuda()
s1, s2 = ConvSite("c1", holder[0], segc, [True], 0), ConvSite("c2", holder[1], [b], [True], 1)
s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
eng.bind(holder, [s1, s2])
eng.blk3_on = int(os.environ.get("FUSE", "2"))
for p in holder.parameters():
    p.requires_grad_(False)
x = torch.randn(N, ci, R, R).cuda()
gout = torch.randn(N, co, R, R).cuda()
for it in range(reps + 2):
    eng.begin(); eng.prepare_weights(force=(it == 0)); eng.recording = True
    xt = eng.from_nchw(x, rg=True)
    go = eng.from_nchw(gout)
    y = eng.block2(s1, s2, [xt], 1, res1=xt if co == ci else None)
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, N, R, R, go.cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.recording = False
    eng.backward()
torch.cuda.synchronize()
print("done", key, "fuse", eng.blk3_on)
