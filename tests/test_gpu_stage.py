"""The per-image stage interpreter (csrc/stage.hip, causal-gen_amd/stage.py; opt-in: CGEN_STAGE=1): one launch walks a list of
low-resolution ops with one workgroup per image.  Same bodies / same K order as the stand-alone kernels, so the results must
equal the launch-per-op path -- bit for bit on these fixtures -- while the launch count collapses."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _build(fx, stage):
    from causal_gen_amd import dmol, vae
    from causal_gen_amd.hps import Hparams

    args = Hparams(**fx["hp"])
    m = vae.HVAE(args)
    if fx["likelihood"] == "dmol":
        m.likelihood = dmol.DmolNet(args)
    m.load_state_dict(fx["state_dict"])
    m.compute_dtype = "f16"
    m = m.cuda()
    eng = m.engine()
    eng.stage_enabled = stage
    return m, eng


@pytest.mark.parametrize("name", ["tiny_light_c1.pt", "tiny_default_c3.pt", "tiny_condprior_morpho_c1.pt"])
def test_staged_training_pass_equals_launch_per_op(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    fx = torch.load(path, weights_only=False)
    res = {}
    for stage in (False, True):
        m, eng = _build(fx, stage)
        m.train()
        if m.cond_prior:
            m.decoder.__dict__["drop_cond"] = lambda: (1, 1)
        m.noise = [e.clone() for e in fx["fwd"]["eps"]]
        n0 = eng.launches
        out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=fx["fwd"]["beta"])
        out["elbo"].backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone()
        res[stage] = ([float(out[k]) for k in ("elbo", "nll", "kl")], g, eng.launches - n0, eng.stage_launches, eng.stage_ops_total)
    (v0, g0, l0, _, _), (v1, g1, l1, sl, so) = res[False], res[True]
    assert sl > 0 and so > 3 * sl, "the stage interpreter did not run"
    assert l1 < 0.6 * l0, (l0, l1)
    # same f16 inputs, same epilogue; the f32 summation order inside a conv differs between the interpreter and the stand-alone
    # kernels (K split over waves there), and with 11 significand bits a last-bit f32 difference now and then rounds the other way
    for a, b in zip(v0, v1):
        assert abs(a - b) <= 2e-5 * abs(a) + 1e-9, (v0, v1)
    assert float((g0 - g1).norm()) <= 4e-3 * float(g0.norm()), (float((g0 - g1).norm()), float(g0.norm()))


def _close_to_an_ulp(u, v, rel=2e-3):
    """Equal up to f16 roundings that fell the other way (different f32 summation order): a few ulps of the largest value."""
    return float((u.float() - v.float()).abs().max()) <= rel * float(u.float().abs().max()) + 1e-12


def test_staged_inference_equals_launch_per_op():
    path = os.path.join(GOLD, "tiny_light_c1.pt")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    fx = torch.load(path, weights_only=False)
    outs = []
    for stage in (False, True):
        m, eng = _build(fx, stage)
        m.eval()
        eng.rng_ptr()
        eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        with torch.no_grad():
            zs = m.abduct(x, pa)
            a = m.forward_latents(zs, pa.roll(1, 0))
            s = m.sample(pa)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in zs] + [a[0].clone(), a[1].clone(), s[0].clone()])
    for u, v in zip(*outs):
        assert _close_to_an_ulp(u, v)


def test_stage_abi_conv_list_against_cgen_conv2d():
    """cgen_stage_accepts / cgen_stage_plan / cgen_stage_run through the C ABI: a two-conv list (3x3 with ReLU prologue, then
    1x1 with a residual and a ragged output width) against the same two convs as stand-alone cgen_conv2d launches."""
    from causal_gen_amd import _lib
    from causal_gen_amd.engine import ConvSite, Engine

    torch.manual_seed(0)
    c1 = torch.nn.Conv2d(24, 16, 3, padding=1).cuda()
    c2 = torch.nn.Conv2d(16, 12, 1).cuda()
    holder = torch.nn.ModuleList([c1, c2])
    eng = Engine(torch.device("cuda"), "f16")
    eng.bind(holder, [ConvSite("a", c1, [24], [True], 0), ConvSite("b", c2, [16], [True], 1)])
    eng.begin()
    eng.prepare_weights(force=True)
    lib = _lib.load()
    x = eng.from_nchw(torch.randn(5, 24, 9, 9, device="cuda"))
    r = eng.from_nchw(torch.randn(5, 12, 9, 9, device="cuda"))
    outs = []
    for staged in (False, True):
        t = eng.new(5, 9, 9, 16)
        y = eng.new(5, 9, 9, 12)
        eng.fill(eng._padded(y), 7.0)  # (poison: the kernel must write the zero padding of the ragged width itself)
        y.cpad = 16
        a1, a2 = _lib.ConvArgs(), _lib.ConvArgs()
        for a, site, src, dst, act, res in ((a1, eng.sites[0], x, t, _lib.ACT_RELU, None), (a2, eng.sites[1], t, y, _lib.ACT_NONE, r)):
            a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act, a.dact = _lib.F16, 5, 9, 9, site.ks, 1, act, 0
            a.seg[0] = src.cv()
            a.weight, a.bias, a.out = site.img_fwd, site.conv.bias.data_ptr(), dst.cv()
            a.aux = a.res2 = _lib.NULL_VIEW
            a.res1 = res.cv() if res is not None else _lib.NULL_VIEW
        torch.cuda.synchronize()
        if not staged:
            lib.conv2d(C.byref(a1), eng.stream)
            lib.conv2d(C.byref(a2), eng.stream)
        else:
            assert lib.stage_accepts(_lib.ST_CONV, C.addressof(a1)) > 0 and lib.stage_accepts(_lib.ST_CONV, C.addressof(a2)) > 0
            kinds = (C.c_int32 * 2)(_lib.ST_CONV, _lib.ST_CONV)
            ptrs = (C.c_void_p * 2)(C.addressof(a1), C.addressof(a2))
            nbytes, lds = C.c_int64(0), C.c_int32(0)
            lib.stage_plan(kinds, ptrs, 2, None, 0, C.byref(nbytes), C.byref(lds))
            host = (C.c_char * nbytes.value)()
            lib.stage_plan(kinds, ptrs, 2, host, nbytes.value, C.byref(nbytes), C.byref(lds))
            dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
            lib.stage_run(dev.data_ptr(), 2, 5, lds.value, eng.stream)
        torch.cuda.synchronize()
        outs.append((eng.to_nchw(t).clone(), eng.to_nchw(eng._padded(y)).clone()))
    assert _close_to_an_ulp(outs[0][0], outs[1][0])
    assert _close_to_an_ulp(outs[0][1][:, :12], outs[1][1][:, :12])
    assert float(outs[1][1][:, 12:].abs().max()) == 0.0  # channels [Co, cpad) zero-filled
    ref = torch.nn.functional.conv2d(torch.relu(eng.to_nchw(x)), c1.weight, c1.bias, padding=1)
    assert float((outs[1][0] - ref).abs().max()) <= 0.05 * float(ref.abs().max())
    # not served: f32, a 5x5 kernel
    a1.dtype = _lib.F32
    assert lib.stage_accepts(_lib.ST_CONV, C.addressof(a1)) == 0
