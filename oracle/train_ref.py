"""Functional CPU restatement of the reference's per-step training harness -- ORACLE, test-only.

  preprocess_batch ..... trainer.py:16-21
  linear_warmup ........ utils.py:32-36
  RefTrainer.step ...... trainer.py:54-87 (fwd, bwd, clip_grad_norm_(350), NaN / grad_norm<500 skip,
                         AdamW with betas [0.9,0.9] (train_setup.py:42-53, hps.py:118-124),
                         LambdaLR warm-up, EMA)
  ema_decay / EMA ...... utils.py:169-225 (update_after_step=100, inv_gamma=1, power=1, beta=.999)

AdamW is written out (decoupled decay, bias-corrected moments, eps=1e-8) rather than calling
torch.optim so that the product's fused multi-tensor kernel has an explicit formula to match; the
golden fixtures (tests/golden/train_steps.pt) were produced with torch.optim.AdamW + the reference's
own EMA class and pin this restatement.
"""
import math
from collections import OrderedDict

import torch

from . import hvae_ref


def preprocess_batch(x_u8, pa, input_res=None):
    x = (x_u8.float() - 127.5) / 127.5
    pa = pa.float()
    if input_res is not None:
        pa = pa[..., None, None].repeat(1, 1, input_res, input_res)
    return x, pa


def linear_warmup(warmup_iters):
    return lambda it: 1.0 if it > warmup_iters else it / warmup_iters


def ema_decay(step_before_increment, beta=0.999, update_after_step=100):
    """Decay used by the call that sees ``self.step == step_before_increment`` (utils.py:169-193).
    Returns None for the straight-copy phase.  The first call past ``update_after_step`` finds
    ``initted`` False, copies the online weights and then averages against that copy (a no-op), so it
    is a copy too: the first effective decay is 2/3 on call 103, not 1/2 on call 102."""
    if step_before_increment <= update_after_step + 1:
        return None
    epoch = max((step_before_increment + 1) - update_after_step - 1, 0.0)
    if epoch <= 0:
        return 0.0
    return min(max(1 - (1 + epoch) ** -1.0, 0.0), beta)


def clip_coef(total_norm, max_norm):
    """torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1."""
    return min(max_norm / (total_norm + 1e-6), 1.0)


class RefTrainer:
    def __init__(self, sd, hp, frozen=()):
        self.hp = hp
        self.sd = OrderedDict((k, v.detach().clone().requires_grad_(k not in frozen)) for k, v in sd.items())
        self.m = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.ema = OrderedDict((k, v.detach().clone()) for k, v in self.sd.items())
        self.opt_steps = 0  # successful optimiser steps (drives lr schedule, Adam t, EMA step)
        self.skipped = 0
        self.warm = linear_warmup(hp.lr_warmup_steps)

    def lr(self):
        # LambdaLR is stepped after each optimiser step and evaluated at construction (=> lr(0) = 0)
        return self.hp.lr * self.warm(self.opt_steps)

    def apply_grads(self, grads, nll=None, kl=None):
        """clip -> skip test -> AdamW -> EMA.  ``grads``: dict name -> tensor.  Returns grad_norm."""
        hp = self.hp
        names = [k for k in self.sd if k in grads and grads[k] is not None]
        total = torch.norm(torch.stack([torch.norm(grads[k].detach(), 2.0) for k in names]), 2.0).item()
        c = clip_coef(total, hp.grad_clip)
        bad = (nll is not None and math.isnan(float(nll))) or (kl is not None and math.isnan(float(kl)))
        if not (total < hp.grad_skip) or bad or math.isnan(total):
            self.skipped += 1
            return total
        lr = self.lr()
        b1, b2 = hp.betas
        t = self.opt_steps + 1
        with torch.no_grad():
            for k in names:
                g = grads[k].detach() * c
                p = self.sd[k]
                p.mul_(1 - lr * hp.wd)
                self.m[k].lerp_(g, 1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(1 - b2 ** t)).add_(1e-8)
                p.addcdiv_(self.m[k], denom, value=-lr / (1 - b1 ** t))
            d = ema_decay(self.opt_steps, hp.ema_rate)
            for k in self.sd:
                if d is None:
                    self.ema[k].copy_(self.sd[k])
                else:
                    # utils.py:216-218: ma -= (ma - cur) * (1 - decay)
                    self.ema[k].sub_((self.ema[k] - self.sd[k]) * (1.0 - d))
        self.opt_steps += 1
        return total

    def step(self, x, pa, beta=None, noise=None, drop=(1, 1)):
        beta = self.hp.beta if beta is None else beta
        out = hvae_ref.hvae_forward(self.sd, self.hp, x, pa, beta=beta, noise=noise, drop=drop)
        params = [v for v in self.sd.values() if v.requires_grad]
        gs = torch.autograd.grad(out["elbo"] / self.hp.accu_steps, params, allow_unused=True)
        grads = {k: g for (k, v), g in zip([(k, v) for k, v in self.sd.items() if v.requires_grad], gs)}
        gn = self.apply_grads(grads, out["nll"].item(), out["kl"].item())
        return {k: float(v) for k, v in out.items()}, gn

    def iteration(self, i, x, pa, beta=None, noise=None, drop=(1, 1)):
        """One loader iteration of trainer.py:62-87 with gradient accumulation: backward of elbo / accu_steps into the
        running .grad; the optimiser runs when ``i % accu_steps == 0`` (i is the 0-based enumerate index, so iteration 0
        steps on a single micro-batch), tested on the LAST micro-batch's nll / kl; then zero_grad.  Returns
        (out, grad_norm or None)."""
        beta = self.hp.beta if beta is None else beta
        out = hvae_ref.hvae_forward(self.sd, self.hp, x, pa, beta=beta, noise=noise, drop=drop)
        named = [(k, v) for k, v in self.sd.items() if v.requires_grad]
        gs = torch.autograd.grad(out["elbo"] / self.hp.accu_steps, [v for _, v in named], allow_unused=True)
        acc = getattr(self, "_acc", None) or {}
        for (k, _), g in zip(named, gs):
            if g is not None:
                acc[k] = g.detach() if acc.get(k) is None else acc[k] + g.detach()
        self._acc = acc
        gn = None
        if i % self.hp.accu_steps == 0:
            gn = self.apply_grads(acc, out["nll"].item(), out["kl"].item())
            self._acc = {}
        return {k: float(v) for k, v in out.items()}, gn
