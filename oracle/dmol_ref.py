"""Functional CPU restatement of the reference's discretised mixture of logistics (src/dmol.py)
-- ORACLE, test-only.

  dmol_nll .......... discretized_mix_logistic_loss, dmol.py:24-118 (8-bit and low_bit branches)
  dmol_mean ......... mean_discretized_mix_logistic, dmol.py:164-215 (soft / hard / top-k)
  dmol_sample ....... sample_from_discretized_mix_logistic, dmol.py:121-161
  dmolnet_* ......... DmolNet, dmol.py:218-245

Layout is channels-last as in the reference: logits ``l`` [B,H,W,100], image ``x`` [B,H,W,3].
The 100 logits are [10 mixture logits | R:(10 means,10 log-scales,10 coeffs) | G:(...) | B:(...)].
"""
import numpy as np
import torch
import torch.nn.functional as F

NMIX = 10
MIN_LOG_SCALE = -7.0  # dmol.py:37


def _log_softmax(v):
    """log_prob_from_logits, dmol.py:7-11."""
    m = v.max(dim=-1, keepdim=True)[0]
    return v - m - torch.log(torch.exp(v - m).sum(dim=-1, keepdim=True))


def _unpack(l):
    B, H, W, _ = l.shape
    logits = l[..., :NMIX]
    rest = l[..., NMIX:].reshape(B, H, W, 3, 3 * NMIX)
    return logits, rest[..., :NMIX], rest[..., NMIX:2 * NMIX], rest[..., 2 * NMIX:]


def dmol_nll(x, l, low_bit=False):
    """-log p(x) / (H*W*3) per sample (nats/dim), dmol.py:24-118; ``low_bit``: the 5-bit branch (dmol.py:52-60, 88-102)."""
    hb, scale = (1.0 / 31.0, 15.5) if low_bit else (1.0 / 255.0, 127.5)
    logits, means, ls, co = _unpack(l)
    ls = torch.clamp(ls, min=MIN_LOG_SCALE)
    co = torch.tanh(co)
    xe = x.unsqueeze(-1).expand(*x.shape, NMIX)  # [B,H,W,3,M]
    m_r = means[..., 0, :]
    m_g = means[..., 1, :] + co[..., 0, :] * xe[..., 0, :]
    m_b = means[..., 2, :] + co[..., 1, :] * xe[..., 0, :] + co[..., 2, :] * xe[..., 1, :]
    mu = torch.stack([m_r, m_g, m_b], dim=3)
    d = xe - mu
    inv = torch.exp(-ls)
    up = inv * (d + hb)
    um = inv * (d - hb)
    delta = torch.sigmoid(up) - torch.sigmoid(um)
    log_cdf_plus = up - F.softplus(up)
    log_one_minus_cdf_min = -F.softplus(um)
    mid = inv * d
    log_pdf_mid = mid - ls - 2.0 * F.softplus(mid)
    lp = torch.where(
        xe < -0.999, log_cdf_plus,
        torch.where(xe > 0.999, log_one_minus_cdf_min,
                    torch.where(delta > 1e-5, torch.log(torch.clamp(delta, min=1e-12)),
                                log_pdf_mid - np.log(scale))))
    lp = lp.sum(dim=3) + _log_softmax(logits)
    return -1.0 * torch.logsumexp(lp, -1).sum(dim=[1, 2]) / np.prod(x.shape[1:])


def _autoregress(v, co):
    """Sequential RGB clamp shared by mean/sample, dmol.py:142-150 / 196-204."""
    x0 = v[..., 0].clamp(-1.0, 1.0)
    x1 = (v[..., 1] + co[..., 0] * x0).clamp(-1.0, 1.0)
    x2 = (v[..., 2] + co[..., 1] * x0 + co[..., 2] * x1).clamp(-1.0, 1.0)
    return torch.stack([x0, x1, x2], dim=3)


def dmol_mean(l, mask="soft"):
    """dmol.py:164-215 -> (x [B,H,W,3], scale [B,H,W,3])."""
    logits, means, ls, co = _unpack(l)
    if mask == "soft":
        sel = _log_softmax(logits).exp().unsqueeze(-2)
    elif mask == "hard":
        sel = F.one_hot(torch.argmax(logits, dim=3), num_classes=NMIX).float().unsqueeze(-2)
    elif "top" in mask:
        k = int(mask[-1])
        assert k < NMIX
        v, _ = torch.sort(logits, descending=True, dim=-1)
        lg = logits.clone()
        lg[lg < v[..., k - 1][..., None]] -= np.inf
        sel = _log_softmax(lg).exp().unsqueeze(-2)
    else:
        raise ValueError(mask)
    mu = (means * sel).sum(dim=4)
    s = torch.clamp((ls * sel).sum(dim=4), min=MIN_LOG_SCALE)
    c = (torch.tanh(co) * sel).sum(dim=4)
    return _autoregress(mu, c), s.exp()


def dmol_sample(l, t=None, u_mix=None, u_pix=None):
    """dmol.py:121-161; uniforms in [1e-5, 1-1e-5] are injectable for parity."""
    logits, means, ls, co = _unpack(l)
    if u_mix is None:
        u_mix = torch.empty(logits.shape).uniform_(1e-5, 1.0 - 1e-5)
    sel = F.one_hot(torch.argmax(logits - torch.log(-torch.log(u_mix)), dim=3), num_classes=NMIX).float().unsqueeze(-2)
    mu = (means * sel).sum(dim=4)
    s = torch.clamp((ls * sel).sum(dim=4), min=MIN_LOG_SCALE)
    c = (torch.tanh(co) * sel).sum(dim=4)
    if u_pix is None:
        u_pix = torch.empty(mu.shape).uniform_(1e-5, 1.0 - 1e-5)
    if t is not None:
        s = s + torch.tensor(t).log()
    v = mu + s.exp() * (torch.log(u_pix) - torch.log(1.0 - u_pix))
    return _autoregress(v, c), s.exp()


def dmolnet_logits(sd, h):
    """DmolNet.forward, dmol.py:228-229."""
    return F.conv2d(h, sd["likelihood.conv.weight"], sd["likelihood.conv.bias"]).permute(0, 2, 3, 1)


def dmolnet_nll(sd, h, x):
    return dmol_nll(x.permute(0, 2, 3, 1), dmolnet_logits(sd, h))


def dmolnet_sample(sd, h, return_loc=True, t=None, mask="soft"):
    """dmol.py:234-245 -> NCHW (x, scale)."""
    l = dmolnet_logits(sd, h)
    x, s = dmol_mean(l, mask) if return_loc else dmol_sample(l, t=t)
    return x.clamp(-1, 1).permute(0, 3, 1, 2), s.permute(0, 3, 1, 2)
