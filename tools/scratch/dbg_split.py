"""Debug: which 'early final' gradient ranges still change after the DP split point?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from causal_gen_amd.train import TrainStep

m, hp = bench.build_model("ukbb192", "f16")
m = m.cuda()
ts = TrainStep(m, hp, ema=False, use_graph=False)
x, pa = bench.synth_batch("ukbb192", hp, 2, "cuda", 1)
ts.step(x, pa)
eng = ts.eng
snap = {}
def at_split():
    torch.cuda.synchronize()
    snap["g"] = eng.flat_g.clone()
    snap["final"] = set(eng.early_final)
eng.on_split = at_split
ts._coef_for(x, ts.beta)
ts._fwd_bwd(x, pa, ts.beta)
eng.on_split = None
torch.cuda.synchronize()
names = {id(p): n for n, p in m.named_parameters()}
bad = 0
for p in eng.params:
    if id(p) in snap["final"]:
        o, k = eng.p_off[id(p)], p.numel()
        a, b = snap["g"][o:o + k], eng.flat_g[o:o + k]
        if not torch.equal(a, b):
            bad += 1
            if bad < 12:
                print("CHANGED after split:", names[id(p)], float((a - b).abs().max()), float(b.abs().max()), "snap zero?", float(a.abs().max()))
print("early-final params", len(snap["final"]), "changed", bad)
