"""Per-shape timing of the fused light-Block kernel against the two-launch path (forward and data gradient), batch 32.
usage: python tools/bench_blk.py [reps]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from causal_gen_amd.engine import ConvSite, Engine

SHAPES = [(32, 192, [32], 8, 32), (32, 192, [32], 8, 64), (32, 96, [64], 16, 64), (32, 96, [64], 16, 96), (32, 48, [96], 24, 96),
          (32, 48, [96], 24, 128), (32, 24, [128], 32, 128), (32, 24, [128], 32, 160)]
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
        eng.blk_fuse, eng.blk_minres = fuse, 8
        eng.wgrad_flush_frac = []
        holder = torch.nn.ModuleList([c1, c2]).cuda()
        s1, s2 = ConvSite("c1", holder[0], segc, [True] * len(segc), 0), ConvSite("c2", holder[1], [b], [True], 1)
        eng.bind(holder, [s1, s2])
        for p in holder.parameters():
            p.requires_grad_(False)  # time the data path only
        x = torch.randn(N, ci, R, R).cuda()
        res = torch.randn(N, co, R, R).cuda() if co == ci else None
        gout = torch.randn(N, co, R, R).cuda()
        tf, tb = [], []
        for it in range(reps + 3):
            eng.begin(); eng.prepare_weights(force=(it == 0)); eng.recording = True
            xt = eng.from_nchw(x, rg=True); xt.rg = True
            rt = eng.from_nchw(res) if res is not None else None
            go = eng.from_nchw(gout)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            y = eng.block2(s1, s2, [xt], 1, res1=rt)
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
    if os.environ.get("BLK_STAMPS"):
        # one more fused forward + backward with per-phase shader-clock stamps of every workgroup's first tile
        st = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
        os.environ["CGEN_BLK_STAMPS"] = str(st.data_ptr())
        names = ["dma0 issue", "(loop top) load_wa", "relu pass", "barrier (U ready)", "phase A mfma", "finish/partials", "load_pair", "barrier", "reduce", "barrier (T ready)", "dma prefetch issue", "phase B"]
        for mode in ("fwd", "bwd"):
            st.zero_()
            eng.begin(); eng.prepare_weights(force=False); eng.recording = True
            xt = eng.from_nchw(x, rg=True); xt.rg = True
            rt = eng.from_nchw(res) if res is not None else None
            go = eng.from_nchw(gout)
            y = eng.block2(s1, s2, [xt], 1, res1=rt)
            torch.cuda.synchronize()
            if mode == "bwd":
                st.zero_()
                gy = eng.seed_grad(y)
                eng.lib.axpby(eng.dt, N, R, R, go.cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
                eng.recording = False
                eng.backward()
                torch.cuda.synchronize()
            t = st.view(-1, 16).cpu()
            t = t[t[:, 0] > 0]
            line = []
            prev = 0
            for k in range(1, 13):
                col = t[:, k]
                ok = col > 0
                if ok.sum() == 0:
                    continue
                d = (col[ok] - t[ok][:, prev]).float().mean().item()
                line.append("%s %.0f" % (names[k - 1], d))
                prev = k
            print("   stamps %s (%d WGs, cycles): " % (mode, t.shape[0]) + " | ".join(line) + " | total %.0f" % (t[:, 12] - t[:, 0]).float().mean().item(), flush=True)
        del os.environ["CGEN_BLK_STAMPS"]
    fl = 2.0 * 9 * (sum(segc) * b + b * co) * N * R * R
    print("res %3d  %s->%d->%d : fwd two-launch %7.1f us fused %7.1f us | dgrad two-launch %7.1f us fused %7.1f us | fused fwd %.0f TF/s" % (
        R, segc, b, co, row[0][0], row[1][0], row[0][1], row[1][1], fl / (row[1][0] * 1e-6) / 1e12), flush=True)
