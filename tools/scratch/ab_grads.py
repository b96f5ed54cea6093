#!/usr/bin/env python3
"""Gradients of one bf16 step under two environment settings must be bit-identical (launch-structure changes only).
usage: python tools/ab_grads.py CGEN_RIDER 0 1 [config]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import torch

    import bench
    cfg = sys.argv[2]
    m, hp = bench.build_model(cfg, "f16")
    m = m.cuda().train()
    B = 8 if hp.input_res > 64 else 64
    x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
    eng = m.engine()
    eng.rng_ptr()
    eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
    trace = []
    if os.environ.get("AB_TRACE"):
        from causal_gen_amd import engine as E
        main = torch.cuda.current_stream()

        def nts(a, out):
            if isinstance(a, E.NT):
                out.append(a)
            elif isinstance(a, (list, tuple)):
                for b in a:
                    nts(b, out)
            return out

        snaps = []

        def csum(buf):  # asynchronous snapshot now (no host sync: the overlap with the background kernel must survive), checksum later
            nbytes = buf.n * buf.sn * buf.es
            for ch in eng.arena.chunks:
                off = buf.ptr - ch.data_ptr()
                if 0 <= off and off + nbytes <= ch.numel():
                    snaps.append(ch[off:off + nbytes].clone())
                    return len(snaps) - 1
            return None

        def backward_traced():  # Engine.backward with a checksum of every input-gradient buffer after every op
            from causal_gen_amd._lib import F32
            if eng.wgrad_flush_frac:
                eng._wg_total = sum(2.0 * a[0].ci * a[0].taps * a[0].co * a[1][0].n * a[1][0].h * a[1][0].w
                                    for fn, a in eng.tape if fn == eng._bw_conv and a[0].conv.weight.requires_grad)
            win = [int(v) for v in os.environ.get("AB_WINDOW", "0,100000").split(",")]
            for k, (fn, args) in enumerate(reversed(eng.tape)):
                nf = eng._wg_nflush
                fn(*args)
                if eng._wg_nflush != nf:
                    trace.append((k, "FLUSH", "", (), None))
                if win[0] <= k < win[1]:
                    for t in nts(list(args), []):
                        e = eng.grads.get(id(t.base))
                        if e is not None and t.rg:
                            trace.append((k, fn.__name__, getattr(args[0], "name", ""), (t.n, t.h, t.w, t.c), csum(e[0])))
            for bid in list(eng._riders):
                gv, g, acc = eng._riders.pop(bid)
                eng.lib.axpby(eng.dt, g.n, g.h, g.w, g.cv(), gv.cv(), 1.0, 1.0, 1 << 30, 1 if acc else 0, eng.stream)
            eng._reduce_wgrads()
            for p, ptr in eng._pgrad_tmp.values():
                _, c, h, w = p.shape
                v = E.NT(ptr, 1, h, w, c, h * w * c, w * c, c, 4, rg=False)
                eng.lib.nhwc_to_nchw(F32, 1, c, h, w, v.cv(), eng.param_grad_ptr(p), eng.stream)
            eng.tape.clear()
        eng.backward_orig = eng.backward
        eng.backward = backward_traced
    for it in range(int(os.environ.get("AB_STEPS", "1"))):
        m.zero_grad()
        eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
        trace.clear()
        out = m(x, pa, beta=1.0)
        out["elbo"].backward()
    if trace:
        torch.cuda.synchronize()
        def fin(v):
            v = v.view(torch.int16).to(torch.int64)
            return int((v * (torch.arange(v.numel(), device=v.device) % 977 + 1)).sum())
        trace = [t[:4] + (None if t[4] is None else fin(snaps[t[4]]),) for t in trace]
        torch.save(trace, sys.argv[3] + ".trace")
    torch.cuda.synchronize()
    if os.environ.get("AB_TENSORS"):
        # checksum of every gradient buffer of the engine in the order the backward pass first wrote them
        sums = []
        for k, e in enumerate(eng.grads.values()):
            buf = e[0]
            nbytes = buf.n * buf.sn * buf.es
            for ch in eng.arena.chunks:
                off = buf.ptr - ch.data_ptr()
                if 0 <= off and off + nbytes <= ch.numel():
                    v = ch[off:off + nbytes].view(torch.int16).to(torch.int64)
                    sums.append((k, (buf.n, buf.h, buf.w, buf.c, (eng._dbg_names or {}).get(id(e[2]), "?")), int(v.sum()), int((v * (torch.arange(v.numel(), device=v.device) % 977 + 1)).sum())))
                    break
        torch.save(sums, sys.argv[3] + ".tensors")
        if os.environ.get("AB_DUMP"):
            want = [int(v) for v in os.environ["AB_DUMP"].split(",")]
            dump = {}
            for k, e in enumerate(eng.grads.values()):
                if k in want:
                    buf = e[0]
                    nbytes = buf.n * buf.sn * buf.es
                    for ch in eng.arena.chunks:
                        off = buf.ptr - ch.data_ptr()
                        if 0 <= off and off + nbytes <= ch.numel():
                            dump[k] = (ch[off:off + nbytes].view(torch.float16).float().cpu().view(buf.n, -1), (buf.n, buf.h, buf.w, buf.c, buf.sn, buf.sh, buf.sw))
            torch.save(dump, sys.argv[3] + ".dump")
    torch.save({n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}, sys.argv[3])
    print(float(out["elbo"]))
    sys.exit(0)
var, a, b = sys.argv[1:4]
cfg = sys.argv[4] if len(sys.argv) > 4 else "ukbb192"
import torch
outs = []
for v in (a, b):
    f = "/tmp/ab_%s.pt" % v
    r = subprocess.run([sys.executable, __file__, "--worker", cfg, f], env=dict(os.environ, **{var: v}), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs.append((torch.load(f), r.stdout.strip().splitlines()[-1]))
if os.environ.get("AB_TENSORS"):
    ta, tb = torch.load("/tmp/ab_%s.pt.tensors" % a), torch.load("/tmp/ab_%s.pt.tensors" % b)
    diff = [(x[0], x[1]) for x, y in zip(ta, tb) if x != y]
    print("gradient buffers %d, differing %d, first differing (backward order): %s" % (len(ta), len(diff), diff[:6]))
    if os.environ.get("AB_DUMP") and diff:
        da, db = torch.load("/tmp/ab_%s.pt.dump" % a), torch.load("/tmp/ab_%s.pt.dump" % b)
        for k in da:
            x, geo = da[k]; y = db[k][0]
            n, h, w, c, sn, sh, sw = geo
            d = (x != y)
            idx = d.nonzero()
            print("buffer %d geo %s: %d of %d elements differ; batch rows %s; max abs diff %.3g (max abs %.3g)" % (k, geo, int(d.sum()), d.numel(), sorted(set(idx[:, 0].tolist())), float((x - y).abs().max()), float(x.abs().max())))
            if len(idx):
                off = idx[:, 1]
                print("   rows(y) %s cols(x) %s chans %s" % (sorted(set((off // sh).tolist()))[:30], sorted(set(((off % sh) // sw).tolist()))[:30], sorted(set((off % sw).tolist()))[:40]))
    print("last identical before the first difference:", [(x[0], x[1]) for x in ta[:diff[0][0]]][-4:] if diff else None)
if os.environ.get("AB_TRACE"):
    ta, tb = torch.load("/tmp/ab_%s.pt.trace" % a), torch.load("/tmp/ab_%s.pt.trace" % b)
    first = next((i for i, (x, y) in enumerate(zip(ta, tb)) if x != y), None)
    print("flush at op", [t[0] for t in ta if t[1] == "FLUSH"])
    print("trace entries %d / %d" % (len(ta), len(tb)))
    if first is not None:
        for i in range(max(0, first - 6), min(len(ta), first + 10)):
            print("   %s %s" % ("DIFF" if ta[i] != tb[i] else "same", ta[i][:4]))
bad = [n for n in outs[0][0] if not torch.equal(outs[0][0][n], outs[1][0][n])]
if os.environ.get("AB_LIST"):
    names = list(outs[0][0])
    print("identical:", [n for n in names if n not in bad][:400])
    import collections
    mx = sorted(((float((outs[0][0][n].float() - outs[1][0][n].float()).abs().max()) / (float(outs[0][0][n].float().abs().max()) + 1e-20), n) for n in bad), reverse=True)[:12]
    print("largest relative-to-max differences:", mx)
print("%s=%s elbo %s | %s=%s elbo %s | params %d, differing %d %s" % (var, a, outs[0][1], var, b, outs[1][1], len(outs[0][0]), len(bad), bad[:3]))
sys.exit(1 if bad else 0)
