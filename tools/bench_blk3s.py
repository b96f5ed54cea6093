"""Forward launches of the small-image fused Block (blk3s) alone, for rocprofv3 --kernel-trace --stats:
    python tools/bench_blk3s.py [res 12|6] [reps]      (CGEN_BLK3S_DBG: ablation bits, csrc/block.hip)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_gen_amd.engine import ConvSite, Engine

res = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, R, segc, b, co = {12: (32, 12, [160], 40, 160), 6: (32, 6, [192], 48, 192), 121: (32, 12, [160, 4, 160], 40, 32)}[res]
ci = sum(segc)
c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, co, 3, padding=1)
eng = Engine("cuda", "f16")
eng.blk3_on = 2
eng.wgrad_flush_frac = []
holder = torch.nn.ModuleList([c1, c2]).cuda()
rgs = [c >= 8 for c in segc]
s1, s2 = ConvSite("c1", holder[0], segc, rgs, 0), ConvSite("c2", holder[1], [b], [True], 1)
s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
eng.bind(holder, [s1, s2])
eng.blk3_on = int(os.environ.get("FUSE", "2"))
xs = [torch.randn(N, c, R, R).cuda() for c in segc]
resid = torch.randn(N, co, R, R).cuda() if co == ci else None
eng.begin(); eng.prepare_weights(force=True)
xts = [eng.from_nchw(x) for x in xs]
rt = eng.from_nchw(resid) if resid is not None else None
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    eng.block2(s1, s2, xts, 1, res1=rt)
torch.cuda.synchronize()
e0.record()
for it in range(reps):
    eng.block2(s1, s2, xts, 1, res1=rt)
e1.record()
torch.cuda.synchronize()
print("res %d %s->%d->%d fuse %d dbg %s: %.1f us per forward Block (back-to-back launches)" % (R, segc, b, co, eng.blk3_on, os.environ.get("CGEN_BLK3S_DBG", "0"), 1e3 * e0.elapsed_time(e1) / reps))
