#!/usr/bin/env python3
"""CPU experiment: WHICH layers of the binary16 inference path set the counterfactual-pixel deviation?

The oracle's wiring (tools/bf16_trunk_sim.py's Sim over oracle/hvae_ref.py, forward only) with binary16 rounding of the conv operands
and of the tensors between the convs of a Block -- the residual trunk stays f32, which is what the (value, remainder) planes of the
inference path amount to -- on images BELOW a side threshold, and plain f32 on images at / above it.  Printed: deviation of the
counterfactual pixels (max over the pixels whose abducted scale is not degenerate, as tests/test_gpu_fullsize.py masks them) and of the
ELBO from the all-f32 run at identical weights, inputs and noise.

  thr = inf   the path as it is (every conv with 16-bit operands)
  thr = 192   the 192^2 Blocks + likelihood head in f32
  thr = 96 .. ditto from 96^2 up
  head        only the two likelihood 1x1 convs in f32

usage: tools/f16_cf_sim.py [preset] [batch] [fixture]
   fixture: weights perturbed and inputs drawn as the full-size parity fixtures are (oracle/fullsize_recipe.py: 0.35 / sqrt(fan-in) on
   the conv weights, white-noise pixels) instead of init-scale weights on smooth images
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torch.nn.functional as F

import bf16_trunk_sim as T
from oracle import hparams as ohp
from oracle import hvae_ref as R


def h16(x):
    return x.half().float()


class SimRes(T.Sim):
    """16-bit operands / Block-internal tensors on sides < thr (decoder only when enc32), f32 elsewhere; `head32`: likelihood head in f32."""

    def __init__(self, sd, hp, thr, head32=False, dec_only=False):
        self.thr, self.head32, self.dec_only = thr, head32, dec_only
        self.in_enc = False
        super().__init__(sd, hp, None, T.ident, None, None)
        self.ra = self._ra
        self.rh = self._ra

    def _lo(self, x):
        return x.shape[-1] < self.thr and not (self.dec_only and self.in_enc)

    def _ra(self, x):
        return h16(x) if self._lo(x) else x

    def conv(self, x, key, pad=0):
        w = self.sd[key + ".weight"]
        if key.startswith("likelihood") and self.head32:
            return F.conv2d(x, w, self.sd[key + ".bias"], padding=pad)
        if self._lo(x):
            x, w = h16(x), h16(w)
        return F.conv2d(x, w, self.sd[key + ".bias"], padding=pad)

    def encode(self, x):
        self.in_enc = True
        a = super().encode(x)
        self.in_enc = False
        return a


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    over = dict(cond_prior=False) if preset == "morphomnist" else {}
    hp = ohp.make_hparams(preset, **over)
    torch.manual_seed(0)
    sd = R.init_state_dict(hp)
    g = torch.Generator().manual_seed(3)
    fixture = len(sys.argv) > 3 and sys.argv[3] == "fixture"
    x = (torch.randint(0, 256, (B, hp.input_channels, hp.input_res, hp.input_res), generator=g).float() - 127.5) / 127.5
    if fixture:
        import math
        for k in sd:
            p = sd[k]
            s = 0.35 / math.sqrt(p[0].numel()) if p.dim() == 4 and "decoder.bias" not in k else 0.05
            sd[k] = p + torch.randn(p.shape, generator=g) * s
    else:
        for k in sd:
            sd[k] = sd[k] + torch.randn(sd[k].shape, generator=g) * 0.02
        x = F.avg_pool2d(F.pad(x, [2, 2, 2, 2], mode="replicate"), 5, 1)
    pa = torch.randn(B, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, hp.input_res, hp.input_res)
    cf_pa = torch.randn(B, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, hp.input_res, hp.input_res)
    noise = [torch.randn(s, generator=g) for s in T.noise_shapes(hp, B)]
    res = hp.input_res
    INF = 1 << 30
    modes = [("f32", 0, False, False), ("f16 everywhere", INF, False, False), ("f16, head f32", INF, True, False)]
    r = res
    while r >= 12:
        modes.append(("f32 on sides >= %d" % r, r, True, False))
        r //= 2
    modes.append(("f32 encoder + f16 decoder", INF, False, True))
    ref = None
    with torch.no_grad():
        for name, thr, head32, dec_only in modes:
            sim = SimRes(sd, hp, thr, head32, dec_only)
            acts = sim.encode(x)
            h, kls, zs = sim.decode(pa, acts=acts, noise=[e.clone() for e in noise])
            loc, ls = sim.head(h)
            nll = R.dgauss_nll_from_params(loc, ls, x).mean()
            rec_loc, rec_scale = loc.clamp(-1, 1), ls.exp()
            hc, _, _ = sim.decode(cf_pa, latents=zs)
            cl, cs = sim.head(hc)
            u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
            cf = (cl.clamp(-1, 1) + cs.exp() * u).clamp(-1, 1)
            if ref is None:
                ref = (float(nll), cf, rec_loc, rec_scale)
                ok = rec_scale > 1e-3
                print("%-28s nll %.6f  (masked pixels: %d of %d)" % (name, float(nll), int((~ok).sum()), ok.numel()), flush=True)
                continue
            d = (cf - ref[1]).abs()
            print("%-28s cf max abs %.2e (unmasked %.2e) mean %.2e | rec_loc max %.2e | log-scale max %.2e | nll rel %.2e" % (
                name, float(d[ok].max()), float(d.max()), float(d.mean()), float((rec_loc - ref[2]).abs().max()),
                float((rec_scale.log() - ref[3].log()).abs().max()), abs(float(nll) - ref[0]) / abs(ref[0])), flush=True)


if __name__ == "__main__":
    main()
