"""Discretised mixture of logistics likelihood on MI355X -- the ``src/dmol.py`` surface.

Module-level functions, import-compatible with the reference (channels-LAST tensors, 10 mixtures x RGB):

    discretized_mix_logistic_loss(x, l)                         dmol.py:24-118   -> [B] nats/dim, differentiable w.r.t. ``l``
    sample_from_discretized_mix_logistic(l, nr_mix, ...)        dmol.py:121-161
    mean_discretized_mix_logistic(l, nr_mix, mask, ...)         dmol.py:164-215

and ``DmolNet`` with ``forward / nll / sample`` (dmol.py:218-245), swapped into ``HVAE.likelihood`` exactly as in the
reference (SURVEY probe C.6): ``m.likelihood = DmolNet(args)``.  The 1x1 conv (w0 -> 100 logits) runs through the MFMA conv
kernel; the log-prob / mean / sample math is the fused kernels ``cgen_dmol_nll_fwd/bwd`` and ``cgen_dmol_decode``
(``mask`` in {"soft", "hard", "top<k>"}).  HIP only: there is no ATen / CPU fallback (CPU tensors raise ``CgenError``).
Inside an HVAE the model's engine drives the same kernels on its own tape; the standalone entry points below run on a
private engine (``DmolNet``) or straight on the caller's tensors (module-level functions).
"""
import torch
from torch import nn

from . import _lib
from ._lib import ACT_NONE

NR_MIX = 10


def _cl_view(t, c):
    """[B,H,W,c] f32 CUDA tensor -> cgen_view (channel stride 1; any other strides)."""
    assert t.dim() == 4 and t.shape[3] == c and t.stride(3) == 1
    return _lib.View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2), c, 0)


def _prep(name, l, nr_mix=NR_MIX):
    lib = _lib.load()
    _lib.require_gpu()
    if not (isinstance(l, torch.Tensor) and l.is_cuda):
        raise _lib.CgenError(f"{name}: expects tensors on the GPU (there is no CPU path)")
    if l.dim() != 4 or l.shape[-1] != 10 * nr_mix or nr_mix != NR_MIX:
        raise ValueError(f"{name}: logits must be [B,H,W,{10 * NR_MIX}] (10 mixtures x RGB, channels last), got {tuple(l.shape)}")
    return lib, l.detach().to(torch.float32).contiguous()


class _DmolLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, x, low_bit=False):
        lib, lc = _prep("discretized_mix_logistic_loss", l)
        ctx.dt = _lib.F32 | (_lib.DMOL_LOW_BIT if low_bit else 0)
        xc = x.detach().to(lc.device, torch.float32).contiguous()
        B, H, W, _ = lc.shape
        if tuple(xc.shape) != (B, H, W, 3):
            raise ValueError(f"discretized_mix_logistic_loss: x must be [B,H,W,3], got {tuple(xc.shape)}")
        st = torch.cuda.current_stream(lc.device).cuda_stream
        part = torch.empty((B, lib.like_chunks(H, W)), dtype=torch.float32, device=lc.device)
        lib.dmol_nll_fwd(ctx.dt, B, H, W, _cl_view(lc, 100), _cl_view(xc, 3), part.data_ptr(), st)
        ctx.save_for_backward(lc, xc)
        return part.sum(1) / float(H * W * 3)

    @staticmethod
    def backward(ctx, g):
        lc, xc = ctx.saved_tensors
        lib = _lib.load()
        B, H, W, _ = lc.shape
        coef = (g.to(torch.float32) / float(H * W * 3)).contiguous()
        gl = torch.empty_like(lc)
        lib.dmol_nll_bwd(ctx.dt, B, H, W, _cl_view(lc, 100), _cl_view(xc, 3), coef.data_ptr(), 1, _cl_view(gl, 100),
                         torch.cuda.current_stream(lc.device).cuda_stream)
        return gl, None, None


def discretized_mix_logistic_loss(x, l, low_bit=False):
    """-log p(x) / (H*W*3) per sample for the mixture of discretised logistics (dmol.py:24-118): ``x`` [B,H,W,3] in [-1,1],
    ``l`` [B,H,W,100].  Differentiable w.r.t. ``l`` (`cgen_dmol_nll_bwd`).  ``low_bit``: the reference's 5-bit branch (half-bin 1/31,
    mid-bin fallback log 15.5; dmol.py:52-60, 88-102)."""
    return _DmolLoss.apply(l, x, bool(low_bit))


_FREE_RNG = {}


def _decode(name, l, nr_mix, mode, logt, return_scale):
    lib, lc = _prep(name, l, nr_mix)
    B, H, W, _ = lc.shape
    dev = lc.device
    st = torch.cuda.current_stream(dev).cuda_stream
    xo = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    so = torch.empty_like(xo)
    rng = None
    if mode == 2:
        rng = _FREE_RNG.get(dev)
        if rng is None:
            rng = _FREE_RNG[dev] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=dev)
        lib.rng_advance(rng.data_ptr(), 1, st)
    lib.dmol_decode(_lib.F32, B, H, W, _cl_view(lc, 100), mode, rng.data_ptr() if rng is not None else None, 977, float(logt),
                    xo.data_ptr(), so.data_ptr(), st)
    x, s = xo.permute(0, 2, 3, 1), so.permute(0, 2, 3, 1)  # channels last, as the reference returns them
    return (x, s) if return_scale else x


def _mask_mode(mask):
    if mask == "soft":
        return 0
    if mask == "hard":
        return 1
    if isinstance(mask, str) and "top" in mask:  # dmol.py:178-180: "top3" -> int(mask[-1]), must be < nr_mix
        k = int(mask[-1])
        assert 1 <= k < NR_MIX, "invalid top_k"
        return 10 + k
    raise NotImplementedError(f"mask={mask!r}")


def sample_from_discretized_mix_logistic(l, nr_mix, return_scale=False, t=None):
    """dmol.py:121-161: Gumbel-max over the mixture logits, one logistic draw per channel (temperature ``t`` on the scale),
    sequential RGB clamp.  Noise: device Philox seeded from torch.initial_seed(); every call advances the counter."""
    logt = 0.0 if t is None else float(torch.as_tensor(t, dtype=torch.float32).log())
    return _decode("sample_from_discretized_mix_logistic", l, nr_mix, 2, logt, return_scale)


def mean_discretized_mix_logistic(l, nr_mix, mask="soft", return_scale=False):
    """dmol.py:164-215: mixture mean under a soft / hard / top-k mask, no sampling in observation space."""
    return _decode("mean_discretized_mix_logistic", l, nr_mix, _mask_mode(mask), 0.0, return_scale)


class DmolNet(nn.Module):
    kind = "dmol"

    def __init__(self, args):
        super().__init__()
        if args.input_channels != 3:
            raise ValueError("DmolNet models RGB images (the reference's reshape fails for 1 channel too)")
        self.width = args.widths[0]
        self.num_mixtures = NR_MIX
        self.conv = nn.Conv2d(self.width, self.num_mixtures * 10, kernel_size=1, stride=1, padding=0)
        self.mask = "soft"

    # ---- standalone use (inference only; inside an HVAE the model's engine drives these kernels on its tape)
    def _logits(self, h):
        from .engine import ConvSite
        from .vae import _standalone_engine

        eng = _standalone_engine(self, lambda: [ConvSite("conv", self.conv, [self.conv.in_channels], [True], 0)])
        ht = eng.from_nchw(h.to(eng.device, torch.float32))
        return eng, eng.conv(eng.site_by_id[id(self.conv)], [ht], ACT_NONE)

    def _check(self, *ts):
        from .vae import _no_standalone_grad

        _no_standalone_grad(self, *ts)

    def forward(self, h):
        """dmol.py:228-229: the 100 mixture parameters per pixel, channels LAST [B,H,W,100] (f32)."""
        self._check(h)
        with torch.no_grad():
            return self._forward_sa(h)

    def _forward_sa(self, h):
        eng, lt = self._logits(h)
        return eng.to_nchw(lt).permute(0, 2, 3, 1)

    def nll(self, h, x):
        """dmol.py:231-232 (`cgen_dmol_nll_fwd` on the conv's output in place): [B] nats/dim; ``x`` is NCHW."""
        self._check(h, x)
        with torch.no_grad():
            return self._nll_sa(h, x)

    def _nll_sa(self, h, x):
        eng, lt = self._logits(h)
        B, H, W = lt.n, lt.h, lt.w
        xt = eng.from_nchw(x.to(eng.device, torch.float32))
        part = torch.empty((B, eng.lib.like_chunks(H, W)), dtype=torch.float32, device=eng.device)
        eng.lib.dmol_nll_fwd(eng.dt, B, H, W, lt.cv(), xt.cv(), part.data_ptr(), eng.stream)
        return part.sum(1) / float(H * W * 3)

    def sample(self, h, return_loc=True, t=None):
        """dmol.py:234-245 (`cgen_dmol_decode`): (x in [-1, 1], scale), both NCHW."""
        self._check(h)
        with torch.no_grad():
            return self._sample_sa(h, return_loc, t)

    def _sample_sa(self, h, return_loc, t):
        eng, lt = self._logits(h)
        B, H, W = lt.n, lt.h, lt.w
        mode = _mask_mode(self.mask) if return_loc else 2
        logt = 0.0 if (t is None or return_loc) else float(torch.as_tensor(t, dtype=torch.float32).log())
        xo = torch.empty((B, 3, H, W), dtype=torch.float32, device=eng.device)
        so = torch.empty_like(xo)
        if mode == 2:
            eng.rng_advance(1)
        eng.lib.dmol_decode(eng.dt, B, H, W, lt.cv(), mode, eng.rng_ptr(), 977, logt, xo.data_ptr(), so.data_ptr(), eng.stream)
        return xo, so


def use_dmol(hvae, args):
    """Swap the likelihood of an HVAE for a DMoL head (config 3) and drop any cached engine."""
    hvae.likelihood = DmolNet(args).to(next(hvae.parameters()).device)
    hvae.__dict__["_eng"] = None
    return hvae
