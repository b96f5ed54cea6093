"""Op-level parity of the C-ABI kernels (called through the engine) against plain torch-CPU formulas / the oracle."""
import math

import pytest
import numpy as np
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def make_engine(convs, segs, dtype="f32"):
    from causal_gen_amd.engine import ConvSite, Engine

    eng = Engine("cuda", dtype)
    holder = torch.nn.ModuleList(convs).cuda()
    sites = [ConvSite(f"c{i}", c, sc, [True] * len(sc), i) for i, (c, sc) in enumerate(zip(holder, segs))]
    eng.bind(holder, sites)
    eng.begin()
    eng.prepare_weights(force=True)
    return eng, sites


def nhwc_to_torch(eng, t):
    return eng.to_nchw(t).cpu()


CONV_CASES = [
    # (N, H, W, seg channels, Co, ks, act, with_res)
    (2, 9, 7, [5], 3, 3, 1, False),
    (3, 16, 16, [32], 8, 3, 1, True),
    (2, 8, 8, [16, 3], 36, 1, 2, True),
    (2, 12, 12, [64, 4, 64], 16, 3, 1, False),
    (4, 1, 1, [130], 70, 3, 2, False),
    (2, 20, 20, [1], 16, 7, 0, False),
    (2, 6, 6, [3], 32, 7, 0, False),
    (1, 33, 31, [40], 100, 1, 0, False),
    (2, 4, 4, [48, 20], 1, 1, 0, False),
    (130, 2, 2, [8], 8, 1, 2, True),
    (2, 24, 24, [24, 4, 40], 24, 3, 1, False),
    (2, 16, 32, [16], 96, 3, 2, True),
    (3, 40, 24, [8], 64, 3, 1, False),
    (2, 16, 16, [200], 16, 1, 0, False),
    # model-size shapes (px / ws / packed-wgrad kernels; > 6000 pixels so the small-image kernel does not take them).
    # [32] x 3x3 has 12 pixels per LDS piece: column 11 of a tile reads its x+1 / x+2 neighbours across a piece boundary
    (8, 32, 32, [32], 64, 3, 1, True),
    (1, 96, 96, [32], 8, 3, 1, False),
    (8, 32, 32, [48], 192, 3, 0, False),
    (8, 32, 32, [64], 16, 3, 2, True),
    (2, 64, 64, [56], 24, 3, 1, False),
    (4, 48, 48, [96, 4, 96], 24, 3, 1, False),
    (8, 32, 32, [16, 96], 96, 1, 0, True),
    (8, 32, 32, [24], 160, 3, 1, False),
]


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, dtype):
    N, H, W, segc, Co, ks, act, with_res = case
    g = torch.Generator().manual_seed(N * 1000 + H * 37 + Co)
    conv = torch.nn.Conv2d(sum(segc), Co, ks, padding=ks // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / math.sqrt(sum(segc) * ks * ks))
        conv.bias.copy_(torch.randn(Co, generator=g) * 0.3)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segc]
    res = torch.randn(N, Co, H, W, generator=g) if with_res else None
    gout = torch.randn(N, Co, H, W, generator=g)
    if dtype == "f16":  # quantise inputs so both sides see the same values
        xs = [x.half().float() for x in xs]
        res = res.half().float() if with_res else None
        gout = gout.half().float()
    # ---- reference (torch CPU, f32; weights quantised the same way for bf16)
    w_ref = conv.weight.detach().clone()
    if dtype == "f16":
        w_ref = w_ref.half().float()
    w_ref.requires_grad_(True)
    b_ref = conv.bias.detach().clone().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in xs]
    a = torch.cat(xr, dim=1)
    a = F.relu(a) if act == 1 else (F.gelu(a) if act == 2 else a)
    y_ref = F.conv2d(a, w_ref, b_ref, padding=ks // 2)
    if with_res:
        y_ref = y_ref + res
    y_ref.backward(gout)
    # ---- HIP
    eng, (site,) = make_engine([conv], [segc], dtype)
    eng.recording = True
    nts = [eng.from_nchw(x.cuda(), rg=True) for x in xs]
    for t in nts:
        t.rg = True
    rt = eng.from_nchw(res.cuda(), rg=False) if with_res else None
    y = eng.conv(site, nts, act, res1=rt)
    tol = dict(rtol=2e-4, atol=2e-5) if dtype == "f32" else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(nhwc_to_torch(eng, y), y_ref.detach(), **tol)
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.backward()
    torch.cuda.synchronize()
    for t, x in zip(nts, xr):
        gx = nhwc_to_torch(eng, eng.grad_read(t))
        torch.testing.assert_close(gx, x.grad, **tol)
    wtol = dict(rtol=1e-3, atol=1e-4 * max(1.0, math.sqrt(N * H * W) / 8)) if dtype == "f32" else dict(rtol=5e-2, atol=5e-2 * math.sqrt(N * H * W) / 4)
    torch.testing.assert_close(eng.param_grad_view(site.conv.weight).cpu(), w_ref.grad, **wtol)
    torch.testing.assert_close(eng.param_grad_view(site.conv.bias).cpu(), b_ref.grad, **wtol)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] in (1, 3)])
def test_conv_f32_split_operands(case):
    """CGEN_F32S (the inference flavour of the f32 engine): f32 tensors, every product from split binary16 operands (three f16
    MFMAs per K-step in the tiled kernel).  Against torch f64 on the same f32 inputs: far inside binary16's 2^-11 -- the bound is
    5e-6 of the output scale, where a 16-bit-operand conv of these shapes sits at ~3e-4 -- and it really is another kernel than
    the exact f32 path on the shapes the tiled kernel takes (bits differ), on the others the exact kernels serve the call."""
    N, H, W, segc, Co, ks, act, with_res = case
    g = torch.Generator().manual_seed(N * 1000 + H * 37 + Co + 5)
    conv = torch.nn.Conv2d(sum(segc), Co, ks, padding=ks // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / math.sqrt(sum(segc) * ks * ks))
        conv.bias.copy_(torch.randn(Co, generator=g) * 0.3)
    # values over six decades: the scaled remainder keeps small operands exact as well
    xs = [torch.randn(N, c, H, W, generator=g) * 10.0 ** (torch.rand(N, c, H, W, generator=g) * 5.5 - 4.0) for c in segc]
    res = torch.randn(N, Co, H, W, generator=g) if with_res else None
    a = torch.cat([x.double() for x in xs], dim=1)
    a = F.relu(a) if act == 1 else (F.gelu(a) if act == 2 else a)
    y_ref = F.conv2d(a, conv.weight.detach().double(), conv.bias.detach().double(), padding=ks // 2)
    if with_res:
        y_ref = y_ref + res.double()
    eng, (site,) = make_engine([conv], [segc], "f32")
    assert eng.f32_split == 1 and not eng.recording
    nts = [eng.from_nchw(x.cuda()) for x in xs]
    rt = eng.from_nchw(res.cuda()) if with_res else None
    y_split = nhwc_to_torch(eng, eng.conv(site, nts, act, res1=rt)).double()
    eng.f32_split = 0
    y_exact = nhwc_to_torch(eng, eng.conv(site, nts, act, res1=rt)).double()
    scale = float(y_ref.abs().max())
    err_s, err_e = float((y_split - y_ref).abs().max()) / scale, float((y_exact - y_ref).abs().max()) / scale
    assert err_e < 2e-6, err_e
    assert err_s < 5e-6, (err_s, err_e)
    tiled = ks in (1, 3) and H >= 8 and W >= 16 and all(c % 4 == 0 for c in segc)
    if tiled and min(H, W) >= 16:
        assert not torch.equal(y_split, y_exact), "the split-operand kernel did not run"


def test_double_use_accumulates_input_grad():
    """A tensor consumed by two convs and as a residual gets the sum of the three gradients."""
    g = torch.Generator().manual_seed(3)
    c1, c2 = torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.Conv2d(8, 8, 1)
    x = torch.randn(2, 8, 10, 10, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = c1(F.relu(xr)) + xr + c2(xr)
    gout = torch.randn(2, 8, 10, 10, generator=g)
    y_ref.backward(gout)
    eng, (s1, s2) = make_engine([c1, c2], [[8], [8]])
    eng.recording = True
    xt = eng.from_nchw(x.cuda(), rg=True)
    xt.rg = True
    t = eng.conv(s2, [xt], 0)
    y = eng.conv(s1, [xt], 1, res1=xt, res2=t)
    torch.testing.assert_close(nhwc_to_torch(eng, y), y_ref.detach(), rtol=1e-4, atol=1e-5)
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, 2, 10, 10, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.backward()
    torch.testing.assert_close(nhwc_to_torch(eng, eng.grad_read(xt)), xr.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("d", [2, 4, 6, 8])
def test_avgpool(d):
    g = torch.Generator().manual_seed(d)
    x = torch.randn(3, 12, 8 * d // 2 * 2 if d != 6 else 12, 8 * d // 2 * 2 if d != 6 else 12, generator=g)
    x = x[:, :, : (x.shape[2] // d) * d, : (x.shape[3] // d) * d].contiguous()
    xr = x.clone().requires_grad_(True)
    y_ref = F.avg_pool2d(xr, d, d)
    gout = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gout)
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    eng.recording = True
    xt = eng.from_nchw(x.cuda(), rg=True)
    xt.rg = True
    y = eng.pool(xt, d)
    torch.testing.assert_close(nhwc_to_torch(eng, y), y_ref.detach(), rtol=1e-5, atol=1e-6)
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, y.n, y.h, y.w, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.backward()
    torch.testing.assert_close(nhwc_to_torch(eng, eng.grad_read(xt)), xr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("size,d", [(12, 1.5), (14, 1.75), (10, 2.5), (9, 1.2), (16, 2.0), (7, 7.0)])
def test_adaptive_avgpool_float_down_rate(size, d):
    """Block with a float down-rate (vae.py:79-81): F.adaptive_avg_pool2d(out, int(W / d)), forward and backward (overlapping,
    uneven windows), accumulated into an existing gradient as well."""
    g = torch.Generator().manual_seed(int(size * 10 + d * 4))
    x = torch.randn(3, 12, size, size, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.adaptive_avg_pool2d(xr, int(xr.shape[-1] / d)) + 0.5 * xr.mean(dim=(2, 3), keepdim=True)
    gout = torch.randn(3, 12, int(size / d), int(size / d), generator=g)
    F.adaptive_avg_pool2d(xr, int(xr.shape[-1] / d)).backward(gout)
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    eng.recording = True
    xt = eng.from_nchw(x.cuda(), rg=True)
    xt.rg = True
    y = eng.pool(xt, d)
    assert (y.h, y.w) == (int(size / d), int(size / d))
    torch.testing.assert_close(nhwc_to_torch(eng, y), F.adaptive_avg_pool2d(x, int(size / d)), rtol=1e-5, atol=1e-6)
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, y.n, y.h, y.w, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.backward()
    torch.testing.assert_close(nhwc_to_torch(eng, eng.grad_read(xt)), xr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hi,ho", [(1, 4), (1, 6), (4, 8), (6, 12), (8, 14), (14, 28), (48, 96)])
def test_upsample_matches_interpolate(hi, ho):
    g = torch.Generator().manual_seed(hi * 100 + ho)
    x = torch.randn(2, 8, hi, hi, generator=g)
    bias = torch.nn.Parameter(torch.randn(1, 8, ho, ho, generator=g))
    xr = x.clone().requires_grad_(True)
    y_ref = bias + F.interpolate(xr, scale_factor=ho / hi)
    gout = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gout)
    bias_grad = bias.grad.clone()
    holder = torch.nn.Conv2d(1, 1, 1)
    holder.extra = bias
    eng, _ = make_engine([holder], [[1]])
    bias_dev = [p for p in eng.params if p.shape == bias.shape][0]
    eng.recording = True
    xt = eng.from_nchw(x.cuda(), rg=True)
    xt.rg = True
    y = eng.upsample(xt, ho, bias_dev)
    torch.testing.assert_close(nhwc_to_torch(eng, y), y_ref.detach(), rtol=1e-6, atol=1e-6)
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, y.n, y.h, y.w, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.backward()
    torch.testing.assert_close(nhwc_to_torch(eng, eng.grad_read(xt)), xr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(eng.param_grad_view(bias_dev).cpu(), bias_grad, rtol=1e-5, atol=1e-5)


def test_reparam_kl_golden_and_grads():
    from oracle import hvae_ref

    fx = load_golden("ops.pt")["gaussian_kl"]
    g = torch.Generator().manual_seed(1)
    ql, qs, pl, ps = (fx[k] for k in ("q_loc", "q_logscale", "p_loc", "p_logscale"))
    eps = torch.randn(ql.shape, generator=g)
    gz = torch.randn(ql.shape, generator=g)
    logt = math.log(0.8)
    leaves = [t.clone().requires_grad_(True) for t in (ql, qs, pl, ps)]
    z_ref = leaves[0] + (leaves[1] + logt).exp() * eps
    kl_ref = hvae_ref.gaussian_kl(leaves[0], leaves[1] + logt, leaves[2], leaves[3] + logt)
    coef = 0.37
    ((z_ref * gz).sum() + coef * kl_ref.sum()).backward()
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    eng.recording = True
    N, Cc, H, W = ql.shape
    ts = [eng.from_nchw(t.cuda(), rg=True) for t in (ql, qs, pl, ps)]
    for t in ts:
        t.rg = True
    nch = eng.lib.reparam_kl_chunks(H, W, Cc)
    klp = torch.zeros(N * nch, device="cuda")
    z = eng.reparam_kl(ts[0], ts[1], ts[2], ts[3], eng.from_nchw(eps.cuda()), 1, logt, klp.data_ptr(), nch)
    torch.testing.assert_close(nhwc_to_torch(eng, z), z_ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(klp.view(N, nch).sum(1).cpu(), kl_ref.detach().sum(dim=(1, 2, 3)), rtol=1e-5, atol=1e-3)
    # golden (logt = 0)
    klp0 = torch.zeros(N * nch, device="cuda")
    eng.recording = False
    eng.reparam_kl(ts[0], ts[1], ts[2], ts[3], eng.from_nchw(eps.cuda()), 1, 0.0, klp0.data_ptr(), nch)
    torch.testing.assert_close(klp0.view(N, nch).sum(1).cpu(), fx["kl"].sum(dim=(1, 2, 3)), rtol=1e-5, atol=1e-3)
    eng.recording = True
    gzv = eng.seed_grad(z)
    eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gz.cuda()).cv(), gzv.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    cf = torch.tensor([0.0, coef], device="cuda")
    eng.kl_coef_ptr = cf.data_ptr() + 4
    eng.backward()
    for t, leaf in zip(ts, leaves):
        got = nhwc_to_torch(eng, eng.grad_read(t))
        torch.testing.assert_close(got, leaf.grad, rtol=2e-4, atol=2e-3 * leaf.grad.abs().max().item())


@pytest.mark.parametrize("C", [1, 3])
def test_dgauss_nll_golden(C):
    d = load_golden("ops.pt")[f"dgauss_c{C}"]
    sdk = d["state_dict"]
    h, x = d["h"], d["x"]
    N, _, H, W = x.shape
    names = ["x_loc", "x_logscale"] + (["channel_coeffs"] if C == 3 else [])
    raw = torch.cat([F.conv2d(h, sdk[n + ".weight"], sdk[n + ".bias"]) for n in names], dim=1).requires_grad_(True)
    # torch reference on the raw head outputs
    from oracle import hvae_ref

    loc, ls = raw[:, :C], raw[:, C:2 * C].clamp(min=-9.0)
    if C == 3:
        k = torch.tanh(raw[:, 6:9])
        loc = torch.stack([loc[:, 0], loc[:, 1] + k[:, 0] * x[:, 0], loc[:, 2] + k[:, 1] * x[:, 0] + k[:, 2] * x[:, 1]], dim=1)
    nll_ref = hvae_ref.dgauss_nll_from_params(loc, ls, x)
    torch.testing.assert_close(nll_ref.detach(), d["nll"], rtol=1e-5, atol=1e-6)
    (g_ref,) = torch.autograd.grad(nll_ref.sum(), raw)
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    pt = eng.from_nchw(raw.detach().cuda())
    xt = eng.from_nchw(x.cuda())
    nch = eng.lib.like_chunks(H, W)
    part = torch.zeros(N * nch, device="cuda")
    eng.lib.dgauss_nll_fwd(eng.dt, N, H, W, C, pt.cv(), xt.cv(), part.data_ptr(), eng.stream)
    got = part.view(N, nch).sum(1).cpu() / (C * H * W)
    torch.testing.assert_close(got, d["nll"], rtol=2e-5, atol=1e-6)
    gp = eng.new(N, H, W, raw.shape[1])
    coef = torch.tensor([1.0 / (C * H * W)], device="cuda")
    eng.lib.dgauss_nll_bwd(eng.dt, N, H, W, C, pt.cv(), xt.cv(), coef.data_ptr(), 0, gp.cv(), eng.stream)
    torch.testing.assert_close(nhwc_to_torch(eng, gp), g_ref, rtol=2e-3, atol=1e-5 * g_ref.abs().max().item() + 1e-7)
    xo, so = torch.empty(N, C, H, W, device="cuda"), torch.empty(N, C, H, W, device="cuda")
    eng.lib.dgauss_sample(eng.dt, N, H, W, C, pt.cv(), 0.0, None, 0, xo.data_ptr(), so.data_ptr(), eng.stream)
    torch.testing.assert_close(xo.cpu(), d["sample_x"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(so.cpu(), d["sample_scale"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("C", [1, 3])
def test_logit_gaussian_nll_injected_and_philox_noise(C):
    """simple_vae's GaussNet likelihood (cgen_gauss_nll_fwd/bwd): against the oracle with injected dequantisation noise,
    then with device-side Philox uniforms -- the backward pass must see the uniforms the forward pass saw: with loc = 0
    and logscale = 0 its loc-gradient IS minus the logit target, from which the uniforms are recovered, injected, and
    the forward pass repeated."""
    from oracle import simple_ref

    N, H, W = 3, 20, 24
    g = torch.Generator().manual_seed(5 + C)
    raw = torch.randn(N, 2 * C, H, W, generator=g) * 0.5
    raw.requires_grad_(True)
    x = (torch.randint(0, 256, (N, C, H, W), generator=g).float() - 127.5) / 127.5
    u = torch.rand(N, C, H, W, generator=g)
    D = C * H * W
    # oracle: gauss_nll reads loc / logscale through like_params -> identity 1x1 heads over the raw parameters
    eye = torch.eye(2 * C).view(2 * C, 2 * C, 1, 1)
    sd = {"likelihood.x_loc.weight": eye[:C], "likelihood.x_loc.bias": torch.zeros(C),
          "likelihood.x_logscale.weight": eye[C:], "likelihood.x_logscale.bias": torch.zeros(C)}
    nll_ref = simple_ref.gauss_nll(sd, raw, x, u)
    (g_ref,) = torch.autograd.grad(nll_ref.sum(), raw)
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    pt, xt, ut = eng.from_nchw(raw.detach().cuda()), eng.from_nchw(x.cuda()), eng.from_nchw(u.cuda())
    nch = eng.lib.like_chunks(H, W)
    part = torch.zeros(N * nch, device="cuda")
    eng.lib.gauss_nll_fwd(eng.dt, N, H, W, C, pt.cv(), xt.cv(), ut.cv(), None, 0, part.data_ptr(), eng.stream)
    torch.testing.assert_close(part.view(N, nch).sum(1).cpu() / D, nll_ref.detach(), rtol=2e-5, atol=1e-6)
    gp = eng.new(N, H, W, 2 * C)
    coef = torch.tensor([1.0 / D], device="cuda")
    eng.lib.gauss_nll_bwd(eng.dt, N, H, W, C, pt.cv(), xt.cv(), ut.cv(), None, 0, coef.data_ptr(), 0, gp.cv(), eng.stream)
    torch.testing.assert_close(nhwc_to_torch(eng, gp), g_ref, rtol=2e-3, atol=1e-5 * g_ref.abs().max().item() + 1e-7)
    # Philox uniforms
    from causal_gen_amd._lib import NULL_VIEW

    rng = torch.tensor([1234, 3], dtype=torch.int64, device="cuda")
    zt = eng.from_nchw(torch.zeros(N, 2 * C, H, W).cuda())
    one = torch.tensor([1.0], device="cuda")
    eng.lib.gauss_nll_bwd(eng.dt, N, H, W, C, zt.cv(), xt.cv(), NULL_VIEW, rng.data_ptr(), 977, one.data_ptr(), 0, gp.cv(), eng.stream)
    tgt = -nhwc_to_torch(eng, gp)[:, :C].double()
    u_rec = torch.sigmoid(tgt) * 256.0 - (x.double() + 1.0) * 127.5
    assert float(u_rec.min()) > -1e-3 and float(u_rec.max()) < 1 + 1e-3 and abs(float(u_rec.mean()) - 0.5) < 0.02
    part2 = torch.zeros_like(part)
    eng.lib.gauss_nll_fwd(eng.dt, N, H, W, C, pt.cv(), xt.cv(), NULL_VIEW, rng.data_ptr(), 977, part2.data_ptr(), eng.stream)
    want = simple_ref.gauss_nll({k: v.double() for k, v in sd.items()}, raw.detach().double(), x.double(), u_rec.clamp(0, 1)).float()
    torch.testing.assert_close(part2.view(N, nch).sum(1).cpu() / D, want, rtol=1e-3, atol=1e-4)
    # a different Philox offset gives different uniforms
    rng2 = torch.tensor([1234, 4], dtype=torch.int64, device="cuda")
    part3 = torch.zeros_like(part)
    eng.lib.gauss_nll_fwd(eng.dt, N, H, W, C, pt.cv(), xt.cv(), NULL_VIEW, rng2.data_ptr(), 977, part3.data_ptr(), eng.stream)
    torch.cuda.synchronize()
    assert not torch.equal(part2, part3)
    # sample: sigmoid * 256 -> [-1, 1], scale carries the temperature in both modes
    xo, so = torch.empty(N, C, H, W, device="cuda"), torch.empty(N, C, H, W, device="cuda")
    eng.lib.gauss_sample(eng.dt, N, H, W, C, pt.cv(), float(np.log(0.8)), None, 0, xo.data_ptr(), so.data_ptr(), eng.stream)
    wx, ws = simple_ref.gauss_sample(sd, raw.detach(), True, 0.8)
    torch.testing.assert_close(xo.cpu(), wx, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(so.cpu(), ws, rtol=1e-5, atol=1e-7)


def test_dmol_golden():
    d = load_golden("ops.pt")["dmol"]
    l, x = d["l"], d["x"]  # channels-last already
    N, H, W, _ = l.shape
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    lt = eng.wrap_nhwc(l.cuda().contiguous())
    xt = eng.wrap_nhwc(x.cuda().contiguous())
    nch = eng.lib.like_chunks(H, W)
    part = torch.zeros(N * nch, device="cuda")
    eng.lib.dmol_nll_fwd(eng.dt, N, H, W, lt.cv(), xt.cv(), part.data_ptr(), eng.stream)
    got = part.view(N, nch).sum(1).cpu() / (3 * H * W)
    torch.testing.assert_close(got, d["loss"], rtol=1e-4, atol=1e-6)  # north-star: DMoL nats/dim within 1e-4 rel
    gl = eng.new(N, H, W, 100)
    coef = torch.tensor([1.0 / (3 * H * W)], device="cuda")
    eng.lib.dmol_nll_bwd(eng.dt, N, H, W, lt.cv(), xt.cv(), coef.data_ptr(), 0, gl.cv(), eng.stream)
    g = nhwc_to_torch(eng, gl).permute(0, 2, 3, 1)
    torch.testing.assert_close(g, d["grad_l"], rtol=2e-3, atol=2e-6)
    for mode, mask in ((0, "soft"), (1, "hard"), (13, "top3")):
        xo, so = torch.empty(N, 3, H, W, device="cuda"), torch.empty(N, 3, H, W, device="cuda")
        eng.lib.dmol_decode(eng.dt, N, H, W, lt.cv(), mode, None, 0, 0.0, xo.data_ptr(), so.data_ptr(), eng.stream)
        torch.testing.assert_close(xo.cpu().permute(0, 2, 3, 1), d[f"mean_{mask}"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(so.cpu().permute(0, 2, 3, 1), d[f"scale_{mask}"], rtol=1e-4, atol=1e-6)
    # sampling: finite, in range, deterministic per (seed, offset)
    rng = torch.tensor([7, 0], dtype=torch.int64, device="cuda")
    a, b = torch.empty(N, 3, H, W, device="cuda"), torch.empty(N, 3, H, W, device="cuda")
    eng.lib.dmol_decode(eng.dt, N, H, W, lt.cv(), 2, rng.data_ptr(), 5, 0.0, a.data_ptr(), b.data_ptr(), eng.stream)
    a2 = torch.empty_like(a)
    eng.lib.dmol_decode(eng.dt, N, H, W, lt.cv(), 2, rng.data_ptr(), 5, 0.0, a2.data_ptr(), b.data_ptr(), eng.stream)
    assert torch.isfinite(a).all() and a.abs().max() <= 1 and torch.equal(a, a2)


def test_dmol_low_bit_branch_against_reference_vectors():
    """discretized_mix_logistic_loss(x, l, low_bit=True) (dmol.py:52-60, 88-102) through the product's python surface and the C ABI
    flag CGEN_DMOL_LOW_BIT: nats/dim within 1e-4 of the reference-made vectors, gradient w.r.t. the logits."""
    from causal_gen_amd import dmol

    d = load_golden("dmol_lowbit.pt")
    l = d["l"].cuda().requires_grad_(True)
    loss = dmol.discretized_mix_logistic_loss(d["x"].cuda(), l, low_bit=True)
    torch.testing.assert_close(loss.detach().cpu(), d["loss"], rtol=1e-4, atol=1e-6)
    loss.sum().backward()
    torch.testing.assert_close(l.grad.cpu(), d["grad_l"], rtol=2e-3, atol=2e-6)
    loss8 = dmol.discretized_mix_logistic_loss(d["x"].cuda(), d["l"].cuda())
    torch.testing.assert_close(loss8.cpu(), d["loss_8bit"], rtol=1e-4, atol=1e-6)


def test_cf_pixels_and_particles():
    from causal_gen_amd.dscm import cf_pixels

    g = torch.Generator().manual_seed(2)
    sh = (3, 1, 9, 9)
    x, rl, cl = (torch.rand(sh, generator=g) * 2 - 1 for _ in range(3))
    rs, cs = torch.rand(sh, generator=g) * 0.2 + 1e-3, torch.rand(sh, generator=g) * 0.2
    rs[0, 0, 0, 0] = 0.0  # clamp(1e-12) branch
    ref = torch.clamp(cl + cs * ((x - rl) / rs.clamp(min=1e-12)), -1, 1)
    sx, sx2 = torch.zeros(sh, device="cuda"), torch.zeros(sh, device="cuda")
    out = cf_pixels(x.cuda(), rl.cuda(), rs.cuda(), cl.cuda(), cs.cuda(), sx, sx2)
    out = cf_pixels(x.cuda(), rl.cuda(), rs.cuda(), cl.cuda(), cs.cuda(), sx, sx2)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(sx.cpu(), 2 * ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(sx2.cpu(), 2 * ref ** 2, rtol=1e-6, atol=1e-6)


def test_module_level_gaussian_kl_and_sample_gaussian():
    """The import-compatible module functions (vae.py:14-30) run as HIP kernels: KL map vs the reference's own values
    (golden fixture), sampling statistics + determinism of the device Philox stream."""
    from causal_gen_amd import vae

    fx = load_golden("ops.pt")["gaussian_kl"]
    ql, qs, pl, ps = (fx[k].cuda() for k in ("q_loc", "q_logscale", "p_loc", "p_logscale"))
    kl = vae.gaussian_kl(ql, qs, pl, ps)
    assert kl.shape == ql.shape and kl.is_cuda
    torch.testing.assert_close(kl.cpu(), fx["kl"], rtol=1e-5, atol=1e-5)  # tolerance: expf vs torch.exp, f32
    with pytest.raises(Exception):
        vae.gaussian_kl(ql.cpu(), qs.cpu(), pl.cpu(), ps.cpu())  # no CPU path

    loc = torch.full((64, 4, 32, 32), 0.25, device="cuda")
    ls = torch.full_like(loc, math.log(0.5))
    a, b = vae.sample_gaussian(loc, ls), vae.sample_gaussian(loc, ls)
    assert a.shape == loc.shape and not torch.equal(a, b)  # the counter advances between calls
    for s in (a, b):
        assert abs(float(s.mean()) - 0.25) < 5e-3 and abs(float(s.std()) - 0.5) < 5e-3
    assert abs(float(((a - 0.25) / 0.5).pow(4).mean()) - 3.0) < 0.1  # Gaussian kurtosis


def test_reparam_kl_bf16_vec8_path_matches_scalar_formula():
    """The 8-channels-per-thread bf16 kernels (z_dim = 16 in every preset) against the f32 formula on the bf16-rounded
    inputs: z, per-sample KL and all four gradients; plus the Philox branch (same noise as the scalar kernel's mapping)."""
    from oracle import hvae_ref

    g = torch.Generator().manual_seed(4)
    N, Cc, H, W = 3, 16, 5, 7
    bf = lambda t: t.to(torch.float16).float()  # noqa: E731
    ql, pl = bf(torch.randn(N, Cc, H, W, generator=g)), bf(torch.randn(N, Cc, H, W, generator=g))
    qs, ps = bf(torch.randn(N, Cc, H, W, generator=g) * 0.3 - 0.5), bf(torch.randn(N, Cc, H, W, generator=g) * 0.3)
    eps, gz = bf(torch.randn(N, Cc, H, W, generator=g)), bf(torch.randn(N, Cc, H, W, generator=g))
    leaves = [t.clone().requires_grad_(True) for t in (ql, qs, pl, ps)]
    z_ref = leaves[0] + leaves[1].exp() * eps
    kl_ref = hvae_ref.gaussian_kl(*leaves)
    coef = 0.21
    ((z_ref * gz).sum() + coef * kl_ref.sum()).backward()
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]], dtype="f16")
    eng.recording = True
    ts = [eng.from_nchw(t.cuda(), rg=True) for t in (ql, qs, pl, ps)]
    for t in ts:
        t.rg = True
    nch = eng.lib.reparam_kl_chunks(H, W, Cc)
    klp = torch.zeros(N * nch, device="cuda")
    z = eng.reparam_kl(ts[0], ts[1], ts[2], ts[3], eng.from_nchw(eps.cuda()), 1, 0.0, klp.data_ptr(), nch)
    torch.testing.assert_close(nhwc_to_torch(eng, z), z_ref.detach(), rtol=1e-2, atol=1e-2)  # bf16 storage of z
    torch.testing.assert_close(klp.view(N, nch).sum(1).cpu(), kl_ref.detach().sum(dim=(1, 2, 3)), rtol=1e-4, atol=1e-3)
    gzv = eng.seed_grad(z)
    eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gz.cuda()).cv(), gzv.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    cf = torch.tensor([0.0, coef], device="cuda")
    eng.kl_coef_ptr = cf.data_ptr() + 4
    eng.backward()
    for t, leaf in zip(ts, leaves):
        got = nhwc_to_torch(eng, eng.grad_read(t))
        assert (got - leaf.grad).abs().max().item() < 2e-2 * leaf.grad.abs().max().item()  # bf16 z and bf16 gradient storage
    # Philox branch: z - q_loc = exp(q_ls) * N(0,1) with the generator's statistics, deterministic per (seed, offset)
    eng2, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]], dtype="f16")
    big = [eng2.from_nchw(torch.zeros(8, 16, 32, 32, device="cuda")) for _ in range(4)]
    nch2 = eng2.lib.reparam_kl_chunks(32, 32, 16)
    k2 = torch.zeros(8 * nch2, device="cuda")
    eng2.rng_ptr()
    eng2.rng.copy_(torch.tensor([99, 0], dtype=torch.int64, device="cuda"))
    za = nhwc_to_torch(eng2, eng2.reparam_kl(big[0], big[1], big[2], big[3], None, 7, 0.0, k2.data_ptr(), nch2))
    zb = nhwc_to_torch(eng2, eng2.reparam_kl(big[0], big[1], big[2], big[3], None, 7, 0.0, k2.data_ptr(), nch2))
    assert torch.equal(za, zb) and abs(za.mean().item()) < 1e-2 and abs(za.std().item() - 1.0) < 1e-2


def test_reparam_kl_bf16_vec8_on_channel_slices_at_model_size():
    """As the model calls it: q_loc | q_ls and p_loc | p_ls are channel slices of 32- and (32+C)-channel conv outputs
    (pixel stride != 16), 4 x 24 x 24 pixels; forward z / KL and all four gradients against the f32 formula."""
    from oracle import hvae_ref

    g = torch.Generator().manual_seed(8)
    N, Cc, H, W, Cf = 4, 16, 24, 24, 40
    bf = lambda t: t.to(torch.float16).float()  # noqa: E731
    q = bf(torch.randn(N, 2 * Cc, H, W, generator=g) * 0.5)
    p = bf(torch.randn(N, 2 * Cc + Cf, H, W, generator=g) * 0.5)
    eps, gz = bf(torch.randn(N, Cc, H, W, generator=g)), bf(torch.randn(N, Cc, H, W, generator=g))
    leaves = [t.clone().requires_grad_(True) for t in (q[:, :Cc], q[:, Cc:] - 0.5, p[:, :Cc], p[:, Cc:2 * Cc])]
    q = torch.cat([leaves[0].detach(), leaves[1].detach()], 1)  # (the shifted logscale is what the kernel sees too)
    q = bf(q)
    leaves[1] = q[:, Cc:].clone().requires_grad_(True)
    z_ref = leaves[0] + leaves[1].exp() * eps
    kl_ref = hvae_ref.gaussian_kl(*leaves)
    coef = 0.37
    ((z_ref * gz).sum() + coef * kl_ref.sum()).backward()
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]], dtype="f16")
    eng.recording = True
    tq, tp = eng.from_nchw(q.cuda(), rg=True), eng.from_nchw(p.cuda(), rg=True)
    tq.rg = tp.rg = True
    ts = [tq.chan(0, Cc), tq.chan(Cc, 2 * Cc), tp.chan(0, Cc), tp.chan(Cc, 2 * Cc)]
    nch = eng.lib.reparam_kl_chunks(H, W, Cc)
    klp = torch.zeros(N * nch, device="cuda")
    z = eng.reparam_kl(ts[0], ts[1], ts[2], ts[3], eng.from_nchw(eps.cuda()), 1, 0.0, klp.data_ptr(), nch)
    torch.testing.assert_close(nhwc_to_torch(eng, z), z_ref.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(klp.view(N, nch).sum(1).cpu(), kl_ref.detach().sum(dim=(1, 2, 3)), rtol=2e-4, atol=1e-2)
    gzv = eng.seed_grad(z)
    eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gz.cuda()).cv(), gzv.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    cf = torch.tensor([0.0, coef], device="cuda")
    eng.kl_coef_ptr = cf.data_ptr() + 4
    eng.backward()
    torch.cuda.synchronize()
    gq, gp = nhwc_to_torch(eng, eng.grad_read(tq)), nhwc_to_torch(eng, eng.grad_read(tp))
    for got, leaf in ((gq[:, :Cc], leaves[0]), (gq[:, Cc:], leaves[1]), (gp[:, :Cc], leaves[2]), (gp[:, Cc:2 * Cc], leaves[3])):
        assert (got - leaf.grad).abs().max().item() < 2e-2 * leaf.grad.abs().max().item()
    assert gp[:, 2 * Cc:].abs().max().item() == 0.0  # the feature channels of the prior output received no gradient here


def _model_like_cases(n, seed):
    """Random conv cases at model-like sizes (the generator of tools/fuzz_conv.py): bottleneck in / out, cat[h, pa, acts],
    z_proj-like; resolutions 1 .. 96, batch up to 256 at low resolution; they sweep the kernel-selection space."""
    import random

    rng = random.Random(seed)
    widths = [8, 16, 24, 32, 40, 48, 64, 96, 128, 160, 192, 256]
    out = []
    for _ in range(n):
        res = rng.choice([1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 96])
        N = rng.choice([1, 2, 8, 32, 64, 256]) if res <= 16 else (rng.choice([1, 2, 8, 32]) if res <= 48 else rng.choice([1, 2, 4]))
        kind = rng.random()
        if kind < 0.35:
            c = rng.choice(widths[3:]); segc, Co = [c], max(4, c // 4)
        elif kind < 0.65:
            c = rng.choice(widths[3:]); segc, Co = [max(4, c // 4)], c
        elif kind < 0.85:
            c = rng.choice(widths[3:10]); segc, Co = [c, rng.choice([4, 6, 12]), c], max(8, c // 4)
        else:
            c = rng.choice(widths[3:]); segc, Co = [16, rng.choice([4, c])], c
        ks = 1 if (res <= 2 or rng.random() < 0.3) else 3
        act = rng.choice([0, 1, 1, 2])
        with_res = rng.random() < 0.4
        H, W = (1, 1) if res == 1 else (res, res if rng.random() < 0.8 else max(1, res - rng.choice([1, 3])))
        out.append(((N, H, W, segc, Co, ks, act, with_res), "f16" if rng.random() < 0.8 else "f32"))
    return out


@pytest.mark.parametrize("case,dtype", _model_like_cases(28, 5))
def test_conv_fwd_bwd_model_like_shapes(case, dtype):
    test_conv_fwd_bwd(case, dtype)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("cin,co,h,w", [(1, 32, 40, 70), (3, 16, 33, 33), (1, 16, 8, 8), (3, 64, 20, 45), (1, 32, 5, 3)])
def test_stem_direct_7x7(dtype, cin, co, h, w):
    """Encoder.stem (vae.py:104-110) as a real 7x7 site: the direct forward kernel and the 7x7 instance of the tiled weight-gradient
    kernel (bf16; the generic kernel in f32), no patch tensor -- against torch conv2d and against the im2col + 1x1 route they
    replace; ragged tiles, images smaller than the halo."""
    from causal_gen_amd.engine import ConvSite, Engine

    g = torch.Generator().manual_seed(cin * 100 + co + h)
    conv = torch.nn.Conv2d(cin, co, 7, padding=3)
    x = torch.randn(3, cin, h, w, generator=g)
    gout = torch.randn(3, co, h, w, generator=g)
    if dtype == "f16":
        x, gout = x.half().float(), gout.half().float()
    outs = {}
    for direct in (True, False):
        eng = Engine("cuda", dtype)
        holder = torch.nn.ModuleList([conv]).cuda()
        site = ConvSite("stem", holder[0], [cin], [False], 0) if direct else ConvSite("stem", holder[0], [cin * 49], [False], 0, as_1x1=True)
        eng.bind(holder, [site])
        eng.begin()
        eng.prepare_weights(force=True)
        eng.recording = True
        xt = eng.from_nchw(x.cuda())
        n0 = eng.launches
        y = eng.stem(site, xt)
        nl = eng.launches - n0
        gy = eng.seed_grad(y)
        eng.lib.axpby(eng.dt, y.n, y.h, y.w, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
        eng.recording = False
        eng.backward()
        torch.cuda.synchronize()
        outs[direct] = (nhwc_to_torch(eng, y), eng.param_grad_view(holder[0].weight).cpu().clone(), eng.param_grad_view(holder[0].bias).cpu().clone(), nl)
    wq = conv.weight.detach().cpu().half().float() if dtype == "f16" else conv.weight.detach().cpu()
    wr = wq.clone().requires_grad_(True)
    br = conv.bias.detach().cpu().clone().requires_grad_(True)
    ref = F.conv2d(x, wr, br, padding=3)
    ref.backward(gout)
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(outs[True][0], ref.detach(), **tol)
    torch.testing.assert_close(outs[True][0], outs[False][0], **tol)
    gtol = 1e-4 if dtype == "f32" else 2e-2
    for k, r in ((1, wr.grad), (2, br.grad)):
        assert float((outs[True][k] - r).norm()) <= gtol * float(r.norm()) + 1e-5
        assert float((outs[False][k] - r).norm()) <= gtol * float(r.norm()) + 1e-5


def _philox_words(seed, offset, stream, idx):
    """numpy twin of Philox::gen (csrc/common.h): Philox4x32-10 words for counter `idx` (uint64 array) -> [len(idx), 4] uint32."""
    import numpy as np

    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    idx = np.asarray(idx, dtype=np.uint64)
    c0, c1 = idx & mask, idx >> np.uint64(32)
    c2 = np.full_like(idx, (np.uint64(offset) & mask) ^ np.uint64((stream * 0x9E3779B9) & 0xFFFFFFFF))
    c3 = np.full_like(idx, ((np.uint64(offset) >> np.uint64(32)) + np.uint64(stream)) & mask)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack([c0, c1, c2, c3], 1).astype(np.uint32)


def _u01(r):
    import numpy as np

    return ((r >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def test_dmol_sampling_is_the_reference_formula_on_the_kernels_own_uniforms():
    """discretized_mix_logistic sampling (dmol.py:121-161): Gumbel-max over the mixture logits, then a logistic draw per channel,
    autoregressive means, clamp.  The kernel's uniforms are Philox words at (pixel * 4 + {0,1,2}); a numpy Philox twin rebuilds
    them and the oracle's dmol_sample fed with them must give the same pixels and scales -- with and without a temperature."""
    import numpy as np
    from oracle import dmol_ref

    d = load_golden("ops.pt")["dmol"]
    l = d["l"]
    N, H, W, _ = l.shape
    eng, _ = make_engine([torch.nn.Conv2d(1, 1, 1)], [[1]])
    lt = eng.wrap_nhwc(l.cuda().contiguous())
    seed, offset, stream = 1234567, 3, 5
    rng = torch.tensor([seed, offset], dtype=torch.int64, device="cuda")
    gi = np.arange(N * H * W, dtype=np.uint64)
    r, r2, r3 = (_philox_words(seed, offset, stream, gi * np.uint64(4) + np.uint64(k)) for k in range(3))
    mixw = np.concatenate([r, r2, r3[:, :2]], 1)                                   # 10 mixture uniforms per pixel
    pixw = np.stack([r3[:, 2], r3[:, 3], r2[:, 3] ^ np.uint32(0x9E3779B9)], 1)     # one per channel
    f = lambda wds: torch.from_numpy(np.float32(1e-5) + np.float32(1.0 - 2e-5) * _u01(wds))
    u_mix = f(mixw).view(N, H, W, 10)
    u_pix = f(pixw).view(N, H, W, 3)
    for t in (None, 0.7):
        xo, so = torch.empty(N, 3, H, W, device="cuda"), torch.empty(N, 3, H, W, device="cuda")
        eng.lib.dmol_decode(eng.dt, N, H, W, lt.cv(), 2, rng.data_ptr(), stream, 0.0 if t is None else float(np.log(t)), xo.data_ptr(),
                            so.data_ptr(), eng.stream)
        rx, rs = dmol_ref.dmol_sample(l, t=t, u_mix=u_mix, u_pix=u_pix)
        rx = rx.clamp(-1, 1)
        # a Gumbel arg-max decided by < 1e-6 may fall the other way in f32 device math: allow a handful of pixels
        bad = ((xo.cpu().permute(0, 2, 3, 1) - rx).abs() > 1e-4).any(-1)
        assert int(bad.sum()) <= max(1, N * H * W // 500), int(bad.sum())
        torch.testing.assert_close(so.cpu().permute(0, 2, 3, 1)[~bad], rs[~bad], rtol=1e-4, atol=1e-6)


# ----------------------------------------------------------------------------- remainder planes of the f16 residual trunk
REM_CASES = [
    # (N, H, W, seg channels, Co, ks, act, second residual) -- the kernel each one lands in is noted
    (8, 32, 32, [8], 32, 3, 1, False),        # conv_px (the C/4 -> C half of a Block)
    (2, 48, 48, [24], 96, 3, 1, False),       # conv_px, three channel pairs
    (8, 32, 32, [16, 4], 64, 1, 0, True),     # conv_px, z_proj form: h + p_feat + conv(cat[z, pa])
    (4, 12, 12, [40], 160, 3, 1, False),      # conv_smallp (K split over waves, generic epilogue)
    (2, 24, 24, [128], 128, 3, 2, False),     # K too long for conv_px, conv_ws does not take remainder planes -> conv_tile
    (64, 2, 2, [16], 32, 1, 0, True),         # < 5x5 image
    (3, 9, 7, [5], 12, 3, 1, False),          # ragged everything -> generic kernel, scalar epilogue
]


@pytest.mark.parametrize("case", REM_CASES)
def test_conv_remainder_planes(case):
    """cgen_conv_args.out_rem / res1_rem (f16 engine): the residual trunk as value = hi + remainder.  out + out_rem must
    reproduce conv(act(x)) + bias + (res1 + res1_rem) [+ res2] to f32 accuracy (the operands x, w are f16 either way), and
    `out` itself must be the correctly rounded f16 of that sum."""
    from causal_gen_amd.engine import NT

    N, H, W, segc, Co, ks, act, with_r2 = case
    g = torch.Generator().manual_seed(N * 131 + H * 7 + Co)
    conv = torch.nn.Conv2d(sum(segc), Co, ks, padding=ks // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / math.sqrt(sum(segc) * ks * ks))
        conv.bias.copy_(torch.randn(Co, generator=g) * 0.3)
    xs = [torch.randn(N, c, H, W, generator=g).half().float() for c in segc]
    res = torch.randn(N, Co, H, W, generator=g) * 3.0          # a trunk value that needs more than 11 bits
    r_hi = res.half().float()
    r_rem = (res - r_hi).half().float()
    r2 = torch.randn(N, Co, H, W, generator=g).half().float() if with_r2 else None
    a = torch.cat(xs, dim=1)
    a = F.relu(a) if act == 1 else (F.gelu(a) if act == 2 else a)
    if act == 2:
        a = a.half().float()  # (the kernel applies the activation in LDS and rounds the operand back to f16)
    want = F.conv2d(a.double(), conv.weight.detach().half().double(), conv.bias.detach().double(), padding=ks // 2) + r_hi.double() + r_rem.double()
    if with_r2:
        want = want + r2.double()

    eng, (site,) = make_engine([conv], [segc], "f16")
    assert eng.trunk_rem
    nts = [eng.from_nchw(x.cuda()) for x in xs]
    rt = eng.new(N, H, W, Co, rg=False, rem=True)

    def plane(t):
        return NT(t.ptr + t.rem, t.n, t.h, t.w, t.c, t.sn, t.sh, t.sw, t.es, rg=False)

    def put(t, src):
        src = src.cuda().contiguous()
        eng.lib.nchw_to_nhwc(0, eng.dt, N, Co, H, W, src.data_ptr(), t.cv(), 0.0, 1.0, eng.stream)
        torch.cuda.synchronize()

    put(rt, r_hi)
    put(plane(rt), r_rem)
    r2t = eng.from_nchw(r2.cuda()) if with_r2 else None
    y = eng.conv(site, nts, act, res1=rt, res2=r2t, trunk=True)
    assert y.rem > 0
    hi, rem = nhwc_to_torch(eng, y).double(), nhwc_to_torch(eng, plane(y)).double()
    scale = float(want.abs().max())
    tol = (2e-3 if act == 2 else 3e-6) * scale  # GELU: the f16 rounding of the activated operand dominates, as without planes
    assert float((hi + rem - want).abs().max()) <= tol, (float((hi + rem - want).abs().max()), scale)
    # the plain tensor is the correctly rounded sum (what a conv downstream reads is unchanged by the feature), the plane
    # holds at most half an ulp of it
    ulp = torch.pow(2.0, torch.floor(torch.log2(hi.abs().clamp(min=2.0 ** -14))) - 10)
    if act != 2:
        assert bool(((hi - want).abs() <= 0.5 * ulp + 1e-6 * scale).all())
    assert bool((rem.abs() <= 0.5 * ulp * (1 + 2.0 ** -9)).all())
    # and without the planes the same call loses what they carry
    y0 = eng.conv(site, nts, act, res1=NT(rt.ptr, N, H, W, Co, rt.sn, rt.sh, rt.sw, rt.es, rg=False), res2=r2t)
    err0 = float((nhwc_to_torch(eng, y0).double() - want).abs().max())
    assert err0 > 20 * float((hi + rem - want).abs().max()) or act == 2
