#!/usr/bin/env python3
"""Random HVAE architectures through the bf16 throughput path and the f32 parity path (same weights, same noise): ELBO and
per-parameter gradient agreement.  The graph shape changes with every case (block counts, widths, z_dim, light / default
blocks, cond_prior, q_correction, free bits, RGB, DMoL), which exercises the engine's gradient bookkeeping (adoption,
out-of-place accumulation, riders, background flush) beyond the presets.  usage: python tools/fuzz_model.py [n] [seed]
(FUZZ_ONLY=<case> runs one case of the sequence, FUZZ_XSEED=<k> changes its weights and data, FUZZ_FB=<v> overrides the
free-bits draw.  With a tiny batch at 1x1 resolution a single ReLU unit whose pre-activation is ~0 can be gated differently
in bf16 and f32; the failure print localises the error by output row -- all of it in ONE row, gone with another XSEED and
unchanged by CGEN_WGRAD_FLUSH_FRAC=2 CGEN_RIDER=0 is that discontinuity (seed 321 case 17), not bookkeeping.
Round 2 campaign (seeds 777, 2024, 31337, 40 cases each, parents handed over as the stride-0 view): 117 ok; the three FAILs --
777/17, 2024/7, 2024/21 -- are all batch 2, bit-identical under FUZZ_PA=repeat, CGEN_WG2_DBUF=0, CGEN_RIDER=0 and
CGEN_WGRAD_FLUSH_FRAC=2, and gone with FUZZ_XSEED=1 or 2: the same bf16 / f32 gating class.  Note that 2024/7 and 777/17
spread the error over many rows of the worst tensor (an INPUT unit of that conv flipped, so a whole column moves), so "all of
it in one row" is sufficient, not necessary; the invariance under every engine knob plus the XSEED test is the criterion)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from causal_gen_amd import dmol, vae
from causal_gen_amd.hps import setup_hparams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = 0
for case in range(n):
    R = rng.choice([32, 48, 64])
    res, r = [R], R
    while r > 4 and len(res) < 4:
        r //= 2
        res.append(r)
    res.append(1)
    widths = sorted(rng.choice([16, 24, 32]) * (k + 1) for k in range(len(res)))  # non-decreasing with depth, as every preset
    nb = [rng.choice([1, 2, 3]) for _ in res]
    enc = ",".join("%db%dd%d" % (rr, b, (res[i] // res[i + 1]) if i + 1 < len(res) else 1) for i, (rr, b) in enumerate(zip(res, nb)))
    enc = ",".join(enc.split(",")[:-1] + ["1b%d" % nb[-1]])
    dec = ",".join("%db%d" % (rr, rng.choice([1, 2, 3])) for rr in reversed(res))
    light = rng.random() < 0.5
    C = rng.choice([1, 3])
    ov = dict(input_res=R, enc_arch=enc, dec_arch=dec, widths=widths, z_dim=rng.choice([8, 16]), z_max_res=rng.choice([R, R // 2]),
              bias_max_res=R, input_channels=C, context_dim=rng.choice([4, 6, 12]), cond_prior=rng.random() < 0.5,
              q_correction=rng.random() < 0.3, kl_free_bits=rng.choice([0.0, 0.0, 0.05]))
    if os.environ.get("FUZZ_FB"):  # override the free-bits draw of every case (diagnosis)
        ov["kl_free_bits"] = float(os.environ["FUZZ_FB"])
    name = "ukbb192" if light else "morphomnist"
    use_dmol = C == 3 and rng.random() < 0.5
    B = rng.choice([2, 8, 16])
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        for _ in range(3):
            rng.random()  # the draws the skipped case would have made
        continue
    outs = {}
    try:
        for dt in ("f32", "f16", "bf16#2"):
            hp = setup_hparams(name, **ov)
            torch.manual_seed(case)
            m = vae.HVAE(hp)
            if use_dmol:
                m.likelihood = dmol.DmolNet(hp)
            g = torch.Generator().manual_seed(100 + case + 1000 * int(os.environ.get("FUZZ_XSEED", "0")))
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(torch.randn(p.shape, generator=g) * 0.02)
            m.compute_dtype = dt.split("#")[0]
            m = m.cuda().eval() if rng.random() < 0 else m.cuda().train()
            type(m.decoder).drop_cond  # noqa: B018
            if hp.cond_prior:
                m.decoder.__dict__["drop_cond"] = lambda: (1, 1)  # no conditioning dropout: both runs must see the same graph
            x = ((torch.randint(0, 256, (B, C, R, R), generator=g).float() - 127.5) / 127.5).cuda()
            pa = torch.randn(B, hp.context_dim, generator=g).cuda()[..., None, None]
            pa = pa.repeat(1, 1, R, R) if os.environ.get("FUZZ_PA") == "repeat" else pa.expand(-1, -1, R, R)  # stride-0 view: the virtual-parents path
            eng = m.engine()
            eng.rng_ptr()
            eng.rng.copy_(torch.tensor([5, 0], dtype=torch.int64, device=eng.rng.device))
            out = m(x, pa, beta=2.0)
            out["elbo"].backward()
            torch.cuda.synchronize()
            outs[dt] = ({k: float(out[k]) for k in ("elbo", "nll", "kl")}, {nm: p.grad.detach().float().cpu() for nm, p in m.named_parameters() if p.grad is not None})
            # inference path: abduct -> two replays -> counterfactual pixels (same Philox offsets in both dtypes)
            from causal_gen_amd import dscm
            m.eval()
            eng.rng.copy_(torch.tensor([9, 0], dtype=torch.int64, device=eng.rng.device))
            with torch.no_grad():
                cf = dscm.counterfactual(m, x, pa, pa.roll(1, 0))
            torch.cuda.synchronize()
            outs[dt] = outs[dt] + (cf.float().cpu(),)
            del m, eng
        a, b = outs["f32"], outs["f16"]
        rep = [nm for nm in b[1] if not torch.equal(b[1][nm], outs["bf16#2"][1][nm])]
        rep_cf = not torch.equal(b[2], outs["bf16#2"][2])
        rel = max(abs(a[0][k] - b[0][k]) / max(abs(a[0][k]), 1e-6) for k in ("elbo", "nll"))
        errs, cos = [], []
        med = sorted(float(v.norm()) for v in a[1].values())[len(a[1]) // 2]
        for nm, gf in a[1].items():
            den = float(gf.norm())
            if den < 1e-2 * med:  # a vanishing gradient (a deep block behind a 1x1 bottleneck) is rounding noise in bf16
                continue
            gb = b[1][nm]
            errs.append(float((gb - gf).norm()) / den)
            cos.append(float((gb * gf).sum()) / (den * float(gb.norm()) + 1e-30))
        worst = sorted(((float((b[1][nm] - gf).norm()) / max(float(gf.norm()), 1e-30), nm, tuple(gf.shape)) for nm, gf in a[1].items() if float(gf.norm()) > 0), reverse=True)[:4]
        errs.sort()
        cfd = float((a[2] - b[2]).abs().max()) if not use_dmol else 0.0  # (DMoL decodes by arg-max over mixtures: ties flip)
        cff = float(((a[2] - b[2]).abs() > 0.05).float().mean())
        ok = rel < 1e-2 and errs[len(errs) // 2] < 0.03 and min(cos) > 0.97 and cfd < 0.1 and not rep and not rep_cf
        fails += 0 if ok else 1
        print("%s case %d: R%d C%d %s z%d cond%d qc%d fb%.2f dmol%d B%d enc %s dec %s | elbo f32 %.5f bf16 %.5f; grad err median %.4f max %.4f min cos %.4f; cf max|d| %.4f frac>0.05 %.5f; non-reproducible grads %d cf %d" % (
            "ok  " if ok else "FAIL", case, R, C, "light" if light else "default", ov["z_dim"], ov["cond_prior"], ov["q_correction"], ov["kl_free_bits"], use_dmol, B,
            enc, dec, a[0]["elbo"], b[0]["elbo"], errs[len(errs) // 2], errs[-1], min(cos), cfd, cff, len(rep), rep_cf), flush=True)
        if not ok:
            print("     worst:", worst, flush=True)
            w0 = worst[0][1]
            if a[1][w0].dim() >= 2:  # where the error sits: one output row = one unit's gate (ReLU / clamp) decided differently in bf16
                rowe = (b[1][w0] - a[1][w0]).flatten(1).norm(dim=1)
                print("     rows of %s by error: %s of total %.3e" % (w0, [(int(i), "%.3e" % float(rowe[i])) for i in rowe.argsort(descending=True)[:3]], float(rowe.norm())), flush=True)
            print("     |g| f32 of those:", ["%.3e" % float(a[1][w[1]].norm()) for w in worst], "median |g| %.3e" % sorted(float(v.norm()) for v in a[1].values())[len(a[1]) // 2], flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        import traceback
        print("EXC  case %d (%s, %s): %s" % (case, enc, dec, "".join(traceback.format_exception_only(type(e), e)).strip()[:300]), flush=True)
print("%d failures" % fails)
