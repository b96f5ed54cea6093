"""Discretised mixture of logistics likelihood head on MI355X -- the ``src/dmol.py`` surface.

``DmolNet`` has the same ``nll(h, x)`` / ``sample(h, return_loc, t)`` role as ``DGaussNet`` (dmol.py:218-245) and is
swapped into ``HVAE.likelihood`` exactly as in the reference (SURVEY probe C.6): ``m.likelihood = DmolNet(args)``.
The 1x1 conv (w0 -> 100 logits) runs through the MFMA conv kernel; the log-prob / mean / sample math is the fused
kernels cgen_dmol_nll_fwd/bwd and cgen_dmol_decode (10 mixtures, RGB only; ``mask`` in {"soft","hard","top<k>"}).
"""
from torch import nn


class DmolNet(nn.Module):
    kind = "dmol"

    def __init__(self, args):
        super().__init__()
        if args.input_channels != 3:
            raise ValueError("DmolNet models RGB images (the reference's reshape fails for 1 channel too)")
        self.width = args.widths[0]
        self.num_mixtures = 10
        self.conv = nn.Conv2d(self.width, self.num_mixtures * 10, kernel_size=1, stride=1, padding=0)
        self.mask = "soft"

    def forward(self, h):
        raise RuntimeError("DmolNet is a parameter holder; run it through HVAE (HIP engine)")


def use_dmol(hvae, args):
    """Swap the likelihood of an HVAE for a DMoL head (config 3) and drop any cached engine."""
    hvae.likelihood = DmolNet(args).to(next(hvae.parameters()).device)
    hvae.__dict__["_eng"] = None
    return hvae
