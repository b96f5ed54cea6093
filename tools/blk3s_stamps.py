"""Shader-clock stamps of workgroup 0 of the small-image fused Block (CGEN_BLK3_STAMPS): where a launch's time goes, per wave.
usage: python tools/blk3s_stamps.py [12|6|121]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_ST = torch.zeros(1024, dtype=torch.int64, device="cuda")
os.environ["CGEN_BLK3_STAMPS"] = str(_ST.data_ptr())
from causal_gen_amd.engine import ConvSite, Engine

res = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N, R, segc, b, co = {12: (32, 12, [160], 40, 160), 6: (32, 6, [192], 48, 192), 121: (32, 12, [160, 4, 160], 40, 32)}[res]
ci = sum(segc)
c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, co, 3, padding=1)
eng = Engine("cuda", "f16")
eng.blk3_on = 2
holder = torch.nn.ModuleList([c1, c2]).cuda()
rgs = [c >= 8 for c in segc]
s1, s2 = ConvSite("c1", holder[0], segc, rgs, 0), ConvSite("c2", holder[1], [b], [True], 1)
s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
eng.bind(holder, [s1, s2])
xs = [torch.randn(N, c, R, R).cuda() for c in segc]
resid = torch.randn(N, co, R, R).cuda() if co == ci else None
eng.begin(); eng.prepare_weights(force=True)
xts = [eng.from_nchw(x) for x in xs]
rt = eng.from_nchw(resid) if resid is not None else None
for it in range(4):
    eng.block2(s1, s2, xts, 1, res1=rt)
torch.cuda.synchronize()
names = ["entry -> set-up, touches, table", "zero fill + DMA requests", "lane set-up, mask loads", "fragment prologue + barrier (tile landed)", "phase A K loop", "exchange write + barrier",
         "sum + finalise", "barrier", "phase B (all pairs + epilogues)"]
t = _ST.cpu().tolist()
for w in range(4):
    tw = t[w * 32:(w + 1) * 32]
    print("wave %d: " % w + " | ".join("%s %d" % (names[k], tw[k + 1] - tw[k]) for k in range(8)) + " | total %d cycles (100 MHz shader clock counter: x%.0f ns)" % (tw[8] - tw[0], 10))
