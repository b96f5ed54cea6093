"""Encoder.stem at batch 32: the direct 7x7 kernel against im2col + 1x1 conv (forward only, graph-free, min of 20).
usage: python tools/bench_stem.py [res] [cin] [co]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_gen_amd.engine import ConvSite, Engine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 192
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 1
co = int(sys.argv[3]) if len(sys.argv) > 3 else 32
for dt in ("f16", "f32"):
    conv = torch.nn.Conv2d(cin, co, 7, padding=3)
    eng = Engine("cuda", dt)
    holder = torch.nn.ModuleList([conv]).cuda()
    site = ConvSite("stem", holder[0], [cin * 49], [False], 0, as_1x1=True)
    eng.bind(holder, [site])
    x = torch.randn(32, cin, R, R).cuda()
    for direct in (False, True):
        eng.stem_direct = direct
        ts = []
        for it in range(23):
            eng.begin(); eng.prepare_weights(force=(it == 0)); eng.recording = False
            xt = eng.from_nchw(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = eng.stem(site, xt); e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        print("%s res %d cin %d co %d  %-16s %7.1f us" % (dt, R, cin, co, "direct 7x7" if direct else "im2col + 1x1", min(ts)), flush=True)
