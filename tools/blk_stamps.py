"""Per-phase shader-clock stamps of workgroup 0 of the fused Block kernel (CGEN_BLK3_STAMPS), forward and data gradient.
usage: python tools/blk_stamps.py [res ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_STAMPS = torch.zeros(1024, dtype=torch.int64, device="cuda")
os.environ["CGEN_BLK3_STAMPS"] = str(_STAMPS.data_ptr())
from causal_gen_amd.engine import ConvSite, Engine

SHAPES = {192: (32, 192, [32], 8, 32), 96: (32, 96, [64], 16, 64), 48: (32, 48, [96], 24, 96), 24: (32, 24, [128], 32, 128),
          481: (32, 48, [96, 4, 96], 24, 32), 961: (32, 96, [64], 16, 96)}
names = ["issue", "first chunk landed", "phase A done", "partials exchanged", "mid epilogue", "phase B mfma", "epilogue"]
for key in [int(a) for a in sys.argv[1:]] or [192, 96, 48, 24]:
    N, R, segc, b, co = SHAPES[key]
    ci = sum(segc)
    c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, co, 3, padding=1)
    eng = Engine("cuda", "f16")
    eng.blk3_on, eng.blk3_minres = 2, 8
    eng.blk3_res, eng.blk3_res3 = [], []
    eng.wgrad_flush_frac = []
    holder = torch.nn.ModuleList([c1, c2]).cuda()
    rgs = [c >= 8 for c in segc]
    s1, s2 = ConvSite("c1", holder[0], segc, rgs, 0), ConvSite("c2", holder[1], [b], [True], 1)
    s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
    eng.bind(holder, [s1, s2])
    for p in holder.parameters():
        p.requires_grad_(False)
    xs = [torch.randn(N, c, R, R).cuda() for c in segc]
    res = torch.randn(N, co, R, R).cuda() if co == ci else None
    gout = torch.randn(N, co, R, R).cuda()
    st = _STAMPS  # (the library reads CGEN_BLK3_STAMPS once per process: the buffer exists before its first call and is stamped by every launch)
    for mode in ("fwd", "bwd"):
        snap = None
        for it in range(3):
            eng.begin(); eng.prepare_weights(force=(it == 0)); eng.recording = True
            xts = [eng.from_nchw(x, rg=r) for x, r in zip(xs, rgs)]
            rt = eng.from_nchw(res) if res is not None else None
            go = eng.from_nchw(gout)
            torch.cuda.synchronize()
            if mode == "fwd" and it == 2:
                st.zero_()
            y = eng.block2(s1, s2, xts, 1, res1=rt)
            torch.cuda.synchronize()
            if mode == "fwd" and it == 2:
                snap = st.clone()
            gy = eng.seed_grad(y)
            eng.lib.axpby(eng.dt, N, R, R, go.cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
            eng.recording = False
            torch.cuda.synchronize()
            if mode == "bwd" and it == 2:
                st.zero_()
            eng.backward()
            torch.cuda.synchronize()
            if mode == "bwd" and it == 2:
                snap = st.clone()
        tall = snap.cpu().tolist()
        for wv in (0, 3):
          t = tall[wv * 256:(wv + 1) * 256]
          print("res %d %s->%d->%d %s wave %d: prologue %d cycles, start offset vs wave 0 %d" % (R, segc, b, co, mode, wv, t[1] - t[0], t[0] - tall[0]))
          k, prev_end = 2, t[1]
          while k + 6 < 256 and t[k] > 0:
              row = t[k:k + 8]
              if (k - 2) // 8 not in ((0,) if R <= 24 else (0, 1, 2)):
                  prev_end = row[6]; k += 8
                  continue
              d = [row[0] - prev_end] + [row[i + 1] - row[i] for i in range(6)]
              print("   tile %2d: " % ((k - 2) // 8) + " | ".join("%s %d" % (nm, v) for nm, v in zip(names, d)) + " | total %d | last chunk: requests %d, mfma %d" % (row[6] - prev_end, row[7] - row[1] if row[7] > row[1] else -1, row[2] - row[7]))
              print("            absolute (vs wave 0 kernel start): " + " ".join("%d" % (v - tall[0]) for v in row[:7]))
              prev_end = row[6]
              k += 8
              if (k - 2) // 8 >= 6:
                  break
