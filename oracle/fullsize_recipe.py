"""The seeded recipe shared by oracle/make_fullsize.py (reference side, build container) and the full-size parity tests
(GPU box): how the weights are perturbed, which inputs / noise are used, which gradient entries are sampled.  Data
recipe only -- test infrastructure like everything under oracle/."""
import math

import torch

# (preset, batch, DmolNet head)
CASES = [("morphomnist", 4, False), ("cmnist", 4, True), ("ukbb192", 2, False), ("mimic224", 1, False)]


def key(name, dmol):
    return name + ("+dmol" if dmol else "")


def init_bias(mod):  # main.py:51-55
    if type(mod) == torch.nn.Conv2d:
        torch.nn.init.zeros_(mod.bias)


def perturb(model, seed=5):
    """At the reference's init the prior heads are x0 and every bias is 0, so the KL is ~5e-3 and half the graph is dead
    (SURVEY probe C.7).  Add seeded noise, in named_parameters() order, on the CPU generator (device independent)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 4 and "decoder.bias" not in n:
                d = torch.randn(p.shape, generator=g) * (0.35 / math.sqrt(p[0].numel()))
            else:
                d = torch.randn(p.shape, generator=g) * 0.05
            p.add_(d.to(p.device))


def inputs(hp, B, seed=123):
    g = torch.Generator().manual_seed(seed)
    R, C = hp.input_res, hp.input_channels
    x = (torch.randint(0, 256, (B, C, R, R), generator=g).float() - 127.5) / 127.5
    pa = torch.randn(B, hp.context_dim, generator=g)[..., None, None].repeat(1, 1, R, R)
    return x, pa


def eps_sequence(seed, shapes):
    """What ``torch.manual_seed(seed)`` followed by the reference's randn_like draws yields (CPU generator)."""
    torch.manual_seed(seed)
    return [torch.randn(s) for s in shapes]


def grad_names(names):
    """A dozen parameters spread over encoder / decoder / likelihood (first, middle and last of each family)."""
    fam = {}
    for n in names:
        if n.startswith("encoder.stem"):
            k = "stem"
        elif n.startswith("encoder.blocks"):
            k = "enc"
        elif ".prior." in n:
            k = "prior"
        elif ".posterior." in n:
            k = "post"
        elif ".z_proj." in n or ".z_feat_proj." in n:
            k = "zproj"
        elif n.startswith("decoder.bias"):
            k = "dbias"
        elif n.startswith("decoder.blocks"):
            k = "dconv"
        else:
            k = "like"
        fam.setdefault(k, []).append(n)
    out = []
    for k in ("stem", "enc", "prior", "post", "zproj", "dconv", "dbias", "like"):
        v = fam.get(k, [])
        if not v:
            continue
        picks = {v[0], v[len(v) // 2], v[-1]} if k in ("enc", "prior", "post", "dconv") else {v[0], v[-1]}
        out += [n for n in v if n in picks]
    return out


def sample(t, cap=4096):
    f = t.detach().reshape(-1)
    step = max(1, (f.numel() + cap - 1) // cap)
    return f[::step]


def sample_img(t, step=7):
    return t.detach()[:, :, ::step, ::step]
