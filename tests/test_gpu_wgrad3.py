"""Streaming weight-gradient kernel of round 5 (csrc/wgrad3.hip) through the C ABI: cgen_conv2d_wgrad_plan -> cgen_conv2d_wgrad (or the
packed cgen_conv2d_wgrad_batch_plan / _run) -> cgen_wgrad_reduce, against aten::convolution_backward (weight, bias) in f64 on the SAME
binary16 operands (reference: what loss.backward() computes for the conv weights of src/vae.py:53-55, 63-65).  The kernel accumulates in
f32 over up to 10^6 pixels, so the bound is f32-summation sized -- two orders tighter than the engine-level test in test_gpu_ops.py.
Covers both operand roles (X or grad_out as the shifted, tap-packed operand => partial layouts 0 / 1), 1x1 and 3x3, concatenated and
stride-0 (parents) input segments, ragged sizes, P windows, every wave grid, ReLU / GELU / none, and determinism."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, H, W, seg channels, Co, ks, act)   act: 0 none, 1 relu, 2 gelu
CASES = [
    # ukbb192 Block shapes, reduced batch
    (2, 192, 192, [32], 8, 3, 1), (2, 192, 192, [8], 32, 3, 1), (1, 192, 192, [8], 64, 3, 1),
    (2, 96, 96, [64], 16, 3, 1), (2, 96, 96, [16], 64, 3, 1), (2, 96, 96, [64, 4, 64], 16, 3, 1), (2, 96, 96, [16], 32, 3, 1),
    (4, 48, 48, [96], 24, 3, 1), (4, 48, 48, [24], 96, 3, 1), (4, 48, 48, [96, 4, 96], 24, 3, 1), (4, 48, 48, [24], 128, 3, 1),
    (8, 24, 24, [128], 32, 3, 1), (8, 24, 24, [32], 128, 3, 1), (8, 24, 24, [128, 4, 128], 32, 3, 1), (8, 24, 24, [32], 160, 3, 1),
    (8, 12, 12, [160], 40, 3, 1), (8, 12, 12, [40], 160, 3, 1),
    # 1x1 projections / heads
    (4, 48, 48, [16, 4], 96, 1, 0), (4, 48, 48, [16, 96], 96, 1, 0), (2, 96, 96, [16, 4], 64, 1, 0), (8, 24, 24, [144], 128, 1, 0),
    (2, 192, 192, [32], 8, 1, 0),
    # default (GELU) Blocks of the 32x32 / 224^2 presets
    (16, 32, 32, [16], 4, 1, 2), (16, 32, 32, [8], 8, 3, 2), (16, 16, 16, [32], 8, 1, 2), (16, 8, 8, [16], 16, 3, 2), (4, 56, 56, [24], 24, 3, 2),
    (4, 28, 28, [32], 128, 1, 2), (8, 14, 14, [40], 40, 3, 2), (16, 4, 4, [32], 32, 3, 2),
    # ragged / odd
    (2, 9, 7, [5], 3, 3, 1), (3, 17, 33, [24], 40, 3, 1), (1, 33, 31, [40], 100, 1, 0), (2, 4, 4, [48, 20], 1, 1, 0), (5, 6, 6, [48], 16, 3, 1),
    (2, 13, 50, [8, 8], 8, 3, 2), (3, 2, 2, [16], 16, 3, 1),
    # S windows (more than 16 tap-packed fragments), the 7x7 stem, 1x1 images
    (5, 6, 6, [192], 48, 3, 1), (4, 6, 6, [48], 512, 3, 1), (4, 6, 6, [192, 4, 192], 48, 3, 1), (4, 6, 6, [48], 224, 3, 1),
    (2, 20, 20, [1], 16, 7, 0), (2, 6, 6, [3], 32, 7, 0), (2, 64, 64, [1], 32, 7, 0),
    (4, 1, 1, [130], 70, 3, 2), (3, 1, 1, [512], 128, 3, 1), (8, 1, 1, [128], 544, 1, 0), (8, 1, 1, [1028], 128, 1, 1),
]


def _view(t, c):
    from causal_gen_amd import _lib

    n, h, w, cs = t.shape
    return _lib.View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2), c, cs if cs != c else 0)


def _nhwc(x, pad8=True, broadcast=False):
    """NCHW f32 -> NHWC binary16 on the GPU, channels padded to a multiple of 8 with zeros (cgen_view.cpad)."""
    n, c, h, w = x.shape
    cp = (c + 7) // 8 * 8 if pad8 else c
    if broadcast:  # spatially constant segment (the parents): a stride-0 view of [N, 1, 1, cp]
        t = torch.zeros(n, 1, 1, cp, dtype=torch.float16, device="cuda")
        t[..., :c] = x[:, :, 0, 0].cuda().half()[:, None, None, :]
        return t.expand(n, h, w, cp)
    t = torch.zeros(n, h, w, cp, dtype=torch.float16, device="cuda")
    t[..., :c] = x.permute(0, 2, 3, 1).cuda().half()
    return t


def _problem(case, seed, broadcast_seg=None):
    N, H, W, segc, Co, ks, act = case
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segc]
    if broadcast_seg is not None:
        xs[broadcast_seg] = xs[broadcast_seg][:, :, :1, :1].expand(N, segc[broadcast_seg], H, W).contiguous()
    gout = torch.randn(N, Co, H, W, generator=g) * 0.5
    xs = [x.half().float() for x in xs]
    gout = gout.half().float()
    return xs, gout


def _reference(case, xs, gout):
    N, H, W, segc, Co, ks, act = case
    a = torch.cat(xs, 1)
    if act == 1:
        a = F.relu(a).half().float()
    elif act == 2:
        a = F.gelu(a).half().float()  # (the kernel rounds the activated operand to binary16 before the MFMA)
    a = a.double()
    w = torch.zeros(Co, sum(segc), ks, ks, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(a, w, b, padding=ks // 2)
    y.backward(gout.double())
    scale = float((a.abs().double().mean() * gout.abs().double().mean()) * math.sqrt(N * H * W))  # size of a typical sum
    return w.grad, b.grad, scale


def _args(case, xt, gt, with_bias=True):
    from causal_gen_amd import _lib

    N, H, W, segc, Co, ks, act = case
    a = _lib.WgradArgs()
    a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act = 1, N, H, W, ks, len(segc), act
    for k, (t, c) in enumerate(zip(xt, segc)):
        a.seg[k] = _view(t, c)
    a.gout = _view(gt, Co)
    return a


def _run_single(lib, case, xs, gout, broadcast_seg=None, expect_kind=None):
    from causal_gen_amd import _lib

    N, H, W, segc, Co, ks, act = case
    xt = [_nhwc(x, broadcast=(broadcast_seg == k)) for k, x in enumerate(xs)]
    gt = _nhwc(gout)
    a = _args(case, xt, gt)
    kind = C.c_int32(-1)
    nsplit = lib.conv2d_wgrad_plan(C.byref(a), C.byref(kind))
    if expect_kind is not None:
        assert kind.value in expect_kind, (case, kind.value)
    ci = sum(segc)
    nw = Co * ks * ks * ci
    part = torch.full((nsplit * (nw + Co),), float("nan"), dtype=torch.float32, device="cuda")
    a.nsplit = nsplit
    a.partial_w = part.data_ptr()
    a.partial_b = part.data_ptr() + 4 * nsplit * nw
    st = torch.cuda.current_stream().cuda_stream
    lib.conv2d_wgrad(C.byref(a), st)
    gw = torch.full((Co, ci, ks, ks), float("nan"), dtype=torch.float32, device="cuda")
    gb = torch.full((Co,), float("nan"), dtype=torch.float32, device="cuda")
    _reduce(lib, [(part, nsplit, Co, ci, ks, kind.value, gw, gb)], st)
    torch.cuda.synchronize()
    return gw.cpu(), gb.cpu(), kind.value, (xt, gt, part)


def _reduce(lib, items, st):
    from causal_gen_amd import _lib

    descs, csite, cidx = [], [], []
    for part, nsplit, Co, ci, ks, kind, gw, gb in items:
        nw = Co * ks * ks * ci
        d = _lib.WredDesc()
        d.partial_w, d.partial_b = part.data_ptr(), part.data_ptr() + 4 * nsplit * nw
        d.grad_w, d.grad_b = gw.data_ptr(), gb.data_ptr()
        d.co, d.ci_total, d.ks, d.nsplit, d.accumulate, d.unscale = Co, ci, ks, nsplit, 0, 1.0
        d.layout = 1 if kind == 3 else 0
        d.numel = nw + Co
        descs.append(d)
    for i, d in enumerate(descs):
        nch = (d.numel + 1023) // 1024
        csite += [i] * nch
        cidx += list(range(nch))
    arr = (_lib.WredDesc * len(descs))(*descs)
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    cs = torch.tensor(csite, dtype=torch.int32, device="cuda")
    cx = torch.tensor(cidx, dtype=torch.int32, device="cuda")
    lib.wgrad_reduce(dev.data_ptr(), cs.data_ptr(), cx.data_ptr(), len(csite), st)
    torch.cuda.synchronize()


def _check(case, gw, gb, ref_w, ref_b, scale):
    assert torch.isfinite(gw).all() and torch.isfinite(gb).all(), case
    ew = float((gw.double() - ref_w).abs().max())
    eb = float((gb.double() - ref_b).abs().max())
    N, H, W, segc, Co, ks, act = case
    # f32 accumulation of K = N*H*W products against the f64 reference: ~ eps_f32 * sqrt(K) * |typical sum| (scale); the GELU cases add
    # the one-ulp binary16 rounding flips between the kernel's erf approximation and torch's (a few elements per thousand)
    K = N * H * W
    tol = (8.0 if act == 2 else 1.0) * 4e-7 * scale * math.sqrt(K) + 2e-6 * float(ref_w.abs().max()) + 1e-6
    tol_b = 4e-7 * 0.4 * K + 2e-6 * float(ref_b.abs().max()) + 1e-6
    assert ew <= tol, (case, ew, tol, float(ref_w.abs().max()))
    assert eb <= tol_b, (case, eb, tol_b, float(ref_b.abs().max()))


@pytest.mark.parametrize("case", CASES, ids=[("%dx%dx%d_%s_%d_k%d_a%d" % (c[0], c[1], c[2], "+".join(map(str, c[3])), c[4], c[5], c[6])) for c in CASES])
def test_streaming_wgrad_matches_f64_convolution_backward(case):
    from causal_gen_amd import _lib

    lib = _lib.require_gpu()
    xs, gout = _problem(case, 1234 + case[0] * 7 + case[4])
    ref_w, ref_b, scale = _reference(case, xs, gout)
    gw, gb, kind, _ = _run_single(lib, case, xs, gout)
    N, H, W, segc, Co, ks, act = case
    assert kind in (2, 3), ("the streaming kernel must serve every DMA-clean binary16 shape", case, kind)
    _check(case, gw, gb, ref_w, ref_b, scale)


def test_parents_as_a_stride0_segment_and_layout_kinds():
    """The posterior Block's cat[h, pa, acts] with the parents as a spatially constant stride-0 view (engine.from_parents), and both
    partial layouts by name: 96 -> 24 shifts grad_out (kind 3), 24 -> 96 shifts X (kind 2)."""
    from causal_gen_amd import _lib

    lib = _lib.require_gpu()
    case = (4, 48, 48, [96, 4, 96], 24, 3, 1)
    xs, gout = _problem(case, 77, broadcast_seg=1)
    ref_w, ref_b, scale = _reference(case, xs, gout)
    gw, gb, kind, _ = _run_single(lib, case, xs, gout, broadcast_seg=1, expect_kind=(3,))
    _check(case, gw, gb, ref_w, ref_b, scale)
    case2 = (4, 48, 48, [24], 96, 3, 1)
    xs, gout = _problem(case2, 78)
    ref_w, ref_b, scale = _reference(case2, xs, gout)
    gw, gb, kind, _ = _run_single(lib, case2, xs, gout, expect_kind=(2,))
    _check(case2, gw, gb, ref_w, ref_b, scale)


def test_packed_launch_equals_single_launches_bit_for_bit_and_is_deterministic():
    """cgen_conv2d_wgrad_batch_plan / _run (the form the engine uses: every problem of a flush in one launch, optionally with a
    capped grid) writes the same partial slabs as the single launches, bit for bit, twice in a row."""
    from causal_gen_amd import _lib

    lib = _lib.require_gpu()
    st = torch.cuda.current_stream().cuda_stream
    cases = [CASES[7], CASES[8], CASES[11], CASES[3], CASES[17], CASES[23], CASES[30], CASES[0]]
    singles, keep, args, outs = [], [], [], []
    for k, case in enumerate(cases):
        xs, gout = _problem(case, 500 + k)
        gw, gb, kind, (xt, gt, part) = _run_single(lib, case, xs, gout)
        singles.append((gw, gb))
        N, H, W, segc, Co, ks, act = case
        a = _args(case, xt, gt)
        nsplit = lib.conv2d_wgrad_plan(C.byref(a), None)
        nw = Co * ks * ks * sum(segc)
        p2 = torch.full((nsplit * (nw + Co),), float("nan"), dtype=torch.float32, device="cuda")
        a.nsplit, a.partial_w, a.partial_b = nsplit, p2.data_ptr(), p2.data_ptr() + 4 * nsplit * nw
        keep.append((xt, gt, part, p2))
        args.append(a)
        outs.append((p2, nsplit, Co, sum(segc), ks, kind))
    n = len(args)
    arr = (_lib.WgradArgs * n)(*args)
    nbytes, nl = C.c_int64(0), C.c_int32(0)
    elig = (C.c_int32 * n)()
    lib.conv2d_wgrad_batch_plan(arr, n, None, 0, C.byref(nbytes), None, 0, C.byref(nl), elig)
    host = (C.c_char * max(nbytes.value, 1))()
    launches = (_lib.WgradBatchLaunch * max(nl.value, 1))()
    lib.conv2d_wgrad_batch_plan(arr, n, host, nbytes.value, C.byref(nbytes), launches, nl.value, C.byref(nl), elig)
    assert all(elig[i] == 1 for i in range(n))
    blob = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    for cap in (0, 48):
        for rep in range(2):
            for o in outs:
                o[0].fill_(float("nan"))
            lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl.value, cap, st)
            torch.cuda.synchronize()
            for (p2, nsplit, Co, ci, ks, kind), (xt, gt, part, _) in zip(outs, keep):
                assert torch.equal(p2, part), ("packed launch != single launch", cap, rep, Co, ci)


def _fuzz_cases(n, seed):
    import random

    rnd = random.Random(seed)
    out = []
    while len(out) < n:
        ks = rnd.choice([1, 3, 3, 3])
        nseg = rnd.choice([1, 1, 2, 3, 4])
        segc = [rnd.choice([1, 3, 4, 8, 12, 16, 20, 24, 32, 40, 56, 72]) for _ in range(nseg)]
        co = rnd.choice([1, 2, 5, 8, 16, 24, 32, 33, 48, 64, 100, 136])
        h, w = rnd.randint(1, 40), rnd.randint(1, 40)
        n_img = rnd.randint(1, 6)
        if n_img * h * w * (sum(segc) + co) > 600000:
            continue
        out.append((n_img, h, w, segc, co, ks, rnd.choice([0, 1, 1, 2])))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(48, 20260930), ids=lambda c: "%dx%dx%d_%s_%d_k%d_a%d" % (c[0], c[1], c[2], "+".join(map(str, c[3])), c[4], c[5], c[6]))
def test_streaming_wgrad_fuzz(case):
    """Seeded random shapes (1-4 ragged input segments, odd sizes down to 1x1, widths that need P windows and K-split grids): the same
    check as above; every DMA-clean binary16 problem must be served by the streaming kernel."""
    from causal_gen_amd import _lib

    lib = _lib.require_gpu()
    xs, gout = _problem(case, 99 + case[1] * 41 + case[2])
    ref_w, ref_b, scale = _reference(case, xs, gout)
    gw, gb, kind, _ = _run_single(lib, case, xs, gout)
    assert kind in (2, 3), (case, kind)
    _check(case, gw, gb, ref_w, ref_b, scale)
