"""tests/golden/dmol_lowbit.pt: the REFERENCE's ``discretized_mix_logistic_loss(x, l, low_bit=True)`` (src/dmol.py:24-118: 5-bit pixels,
half-bin 1/31, mid-bin fallback log 15.5) and its gradient, on logits with extreme entries (fallback branch) and edge pixels.

    python3 -B oracle/make_dmol_lowbit_golden.py      (build container only)"""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
import torch  # noqa: E402

import dmol as ref_dmol  # noqa: E402  (reference)

gen = torch.Generator().manual_seed(77)
B, R = 3, 7
l = torch.randn(B, R, R, 100, generator=gen) * 1.5
l[0, 0, 0, 10:] *= 6.0  # extreme parameters -> the mid-bin fallback
x = (torch.randint(0, 32, (B, R, R, 3), generator=gen).float() - 15.5) / 15.5  # 5-bit levels
x[0, 0, 1], x[0, 0, 2] = -1.0, 1.0
l.requires_grad_(True)
loss = ref_dmol.discretized_mix_logistic_loss(x, l, low_bit=True)
(gl,) = torch.autograd.grad(loss.sum(), l)
loss8 = ref_dmol.discretized_mix_logistic_loss(x, l.detach())
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dmol_lowbit.pt")
torch.save(dict(l=l.detach(), x=x, loss=loss.detach(), grad_l=gl, loss_8bit=loss8), out)
print(out, os.path.getsize(out) // 1024, "KiB; loss", loss.tolist(), "(8-bit branch on the same inputs:", loss8.tolist(), ")")
