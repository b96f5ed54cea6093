#!/usr/bin/env python3
"""CPU experiment: WHERE does the bf16 path's deviation from f32 come from?  

Runs the oracle's wiring (oracle/hvae_ref.py, forward only) with bf16 rounding injected at chosen places and
prints the ELBO / counterfactual-pixel deviation from the plain f32 run at identical weights and noise:

  ops      conv operands (input and weight) rounded to bf16, everything stored in f32   (what bf16 MFMA forces)
  all      + every stored tensor bf16 (conv outputs, the residual trunk h, heads)        (the round-2 HIP path)
  trunk32  as `all`, but the residual trunk sums (Block residual, h + p_feat + z_proj, upsample + bias) stay f32
  trunk32h as trunk32 + the prior / posterior heads (p_loc, p_ls, q_loc, q_ls) and p_feat stay f32

usage: tools/bf16_trunk_sim.py [preset] [batch]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

from oracle import hparams as ohp
from oracle import hvae_ref as R


def bf(x):
    return x.bfloat16().float()


def ident(x):
    return x


class Sim:
    def __init__(self, sd, hp, rop, rt, ra, rh):
        self.sd, self.hp, self.rop, self.rt, self.ra, self.rh = sd, hp, rop, rt, ra, rh
        self.light = R._is_light(hp)

    def conv(self, x, key, pad=0):
        return F.conv2d(self.rop(x), self.rop(self.sd[key + ".weight"]), self.sd[key + ".bias"], padding=pad)

    def block(self, prefix, x, k, residual, down, head=False):
        pad = 0 if k == 1 else 1
        h = x
        slots = R.conv_slots(self.light)
        for j, s in enumerate(slots):
            kk = k if (self.light or j in (1, 2)) else 1
            h = self.conv(R._act(self.light, h), f"{prefix}.conv.{s}", pad if kk == 3 else 0)
            if j + 1 < len(slots):
                h = self.ra(h)
        if residual:
            if x.shape[1] != h.shape[1]:
                x = self.ra(self.conv(x, f"{prefix}.width_proj"))
            h = self.rt(x + h)
        else:
            h = (self.rh if head else self.ra)(h)
        if down:
            h = self.rt(F.avg_pool2d(h, kernel_size=down, stride=down))
        return h

    def encode(self, x):
        h = self.rt(F.conv2d(x, self.sd["encoder.stem.weight"], self.sd["encoder.stem.bias"], padding=3))
        acts = {}
        for i, (_, _, _, d) in enumerate(R.encoder_spec(self.hp)):
            h = self.block(f"encoder.blocks.{i}", h, 3, True, d)
            r = h.shape[2]
            if r % 2 and r > 1:
                h = F.pad(h, [0, 1, 0, 1])
            acts[h.size(-1)] = h
        return acts

    def decode(self, parents, acts=None, latents=None, noise=None):
        hp, sd = self.hp, self.sd
        blocks, bias_tab = R.decoder_spec(hp)
        bias = {r: sd[f"decoder.bias.{j}"] for j, (r, _) in enumerate(bias_tab)}
        zd = hp.z_dim
        h = z = bias[1].repeat(parents.shape[0], 1, 1, 1)
        kls, zs = [], []
        b = 0
        for i, (res, w, w_next) in enumerate(blocks):
            p = f"decoder.blocks.{i}"
            k = 3 if res > 2 else 1
            pa = parents[..., :res, :res]
            if h.size(-1) < res:
                b = bias[res] if res in bias else 0
                h = self.rt(b + F.interpolate(h, scale_factor=res / h.shape[-1]))
            p_in = self.ra(b + F.interpolate(z, scale_factor=res / z.shape[-1])) if z.size(-1) < res else z
            pin = torch.cat([p_in, pa], dim=1) if hp.cond_prior else p_in
            pout = self.block(p + ".prior", pin, k, False, None, head=True)
            p_loc, p_ls, p_feat = pout[:, :zd], pout[:, zd:2 * zd], pout[:, 2 * zd:]
            if res <= hp.z_max_res:
                if acts is not None:
                    qin = torch.cat([h, pa, acts[res]], dim=1)
                    q_loc, q_ls = self.block(p + ".posterior", qin, k, False, None, head=True).chunk(2, dim=1)
                    z = self.ra(q_loc + q_ls.exp() * noise.pop(0))
                    kls.append(R.gaussian_kl(q_loc, q_ls, p_loc, p_ls))
                    zs.append(z)
                else:
                    zi = latents[len(zs)] if latents is not None else None
                    z = zi
                    zs.append(z)
            else:
                z = p_loc
            zp = self.conv(torch.cat([z, pa], dim=1), p + ".z_proj")
            h = self.rt(h + p_feat + zp)
            h = self.block(p + ".conv", h, k, True, None)
            if i + 1 < len(blocks):
                z = self.ra(self.conv(torch.cat([z, p_feat], dim=1), p + ".z_feat_proj"))
        return h, kls, zs

    def head(self, h):
        loc = self.conv(h, "likelihood.x_loc")
        ls = self.conv(h, "likelihood.x_logscale").clamp(min=R.MIN_LOGSCALE)
        return loc, ls


def run(sim, x, pa, cf_pa, noise):
    acts = sim.encode(x)
    h, kls, zs = sim.decode(pa, acts=acts, noise=[e.clone() for e in noise])
    loc, ls = sim.head(h)
    nll = R.dgauss_nll_from_params(loc, ls, x).mean()
    kl = sum(k.sum(dim=(1, 2, 3)) for k in kls)
    kl = (kl / np.prod(x.shape[1:])).mean()
    rec_loc, rec_scale = loc.clamp(-1, 1), ls.exp()
    hc, _, _ = sim.decode(cf_pa, latents=zs)
    cl, cs = sim.head(hc)
    cf_loc, cf_scale = cl.clamp(-1, 1), cs.exp()
    u = (x - rec_loc) / rec_scale.clamp(min=1e-12)
    cf = (cf_loc + cf_scale * u).clamp(-1, 1)
    return float(nll + kl), float(nll), float(kl), cf, rec_loc


def noise_shapes(hp, B):
    blocks, _ = R.decoder_spec(hp)
    return [(B, hp.z_dim, r, r) for r, _, _ in blocks if r <= hp.z_max_res]


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "morphomnist"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    over = {}
    if preset == "morphomnist":
        over = dict(cond_prior=False)  # exogenous-prior form (what DSCM.forward serves)
    hp = ohp.make_hparams(preset, **over)
    torch.manual_seed(0)
    sd = R.init_state_dict(hp)
    g = torch.Generator().manual_seed(3)
    for k in sd:  # move off the zero-prior init so every path carries signal (as tools/f16_vs_f32.py does)
        sd[k] = sd[k] + torch.randn(sd[k].shape, generator=g) * 0.02
    x = (torch.randint(0, 256, (B, hp.input_channels, hp.input_res, hp.input_res), generator=g).float() - 127.5) / 127.5
    x = F.avg_pool2d(F.pad(x, [2, 2, 2, 2], mode="replicate"), 5, 1)  # smooth, image-like
    pa = torch.randn(B, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, hp.input_res, hp.input_res)
    cf_pa = torch.randn(B, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, hp.input_res, hp.input_res)
    noise = [torch.randn(s, generator=g) for s in noise_shapes(hp, B)]
    modes = {
        "f32": (ident, ident, ident, ident),
        "ops": (bf, ident, ident, ident),
        "all": (bf, bf, bf, bf),
        "trunk32": (bf, ident, bf, bf),
        "trunk32h": (bf, ident, bf, ident),
    }
    ref = None
    with torch.no_grad():
        for name, (rop, rt, ra, rh) in modes.items():
            out = run(Sim(sd, hp, rop, rt, ra, rh), x, pa, cf_pa, noise)
            if ref is None:
                ref = out
                print("%-9s elbo %.6f nll %.6f kl %.6f" % (name, out[0], out[1], out[2]))
                continue
            print("%-9s elbo rel %.2e  nll rel %.2e  kl rel %.2e | cf max abs %.2e mean abs %.2e | rec_loc max abs %.2e" % (
                name, abs(out[0] - ref[0]) / abs(ref[0]), abs(out[1] - ref[1]) / abs(ref[1]), abs(out[2] - ref[2]) / max(abs(ref[2]), 1e-30),
                float((out[3] - ref[3]).abs().max()), float((out[3] - ref[3]).abs().mean()), float((out[4] - ref[4]).abs().max())))


if __name__ == "__main__":
    main()
