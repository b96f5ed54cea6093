"""Fused default-Block kernel of round 6 (csrc/block4.hip, cgen_block4: GELU -> 1x1 -> GELU -> 3x3 -> GELU -> 3x3 -> GELU -> 1x1 per
launch, forward and data gradient; vae.py:57-71,73-84) against (a) the four-launch HIP path it replaces -- same binary16 storage points
(every intermediate tensor is rounded to f16 in both), so the results agree up to an f16 ulp here and there -- and (b) a torch f32
reference of the reference's Block on the same f16-quantised operands."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, H, W, input segments, differentiable?, bottleneck, out channels, residual?)  -- default-Block shapes of morphomnist / cmnist / mimic224
CASES = [
    (4, 32, 32, [16], [1], 4, 16, True),             # MNIST 32^2 trunk: ragged 4-channel bottleneck (zero padded to 8)
    (4, 32, 32, [16, 12, 16], [1, 0, 1], 4, 32, False),  # MNIST 32^2 posterior: cat[h, pa (12 channels), acts]
    (4, 16, 16, [32, 12], [1, 0], 8, 64, False),     # MNIST 16^2 conditional prior: 2 z_dim + width out
    (8, 8, 8, [64], [1], 16, 64, True),              # one tile per image
    (8, 4, 4, [128], [1], 32, 128, True),            # 4^2: a tile that is mostly outside the image
    (2, 40, 52, [32], [1], 8, 32, True),             # 224^2 widths, ragged tiles (52 = 3.25 tiles wide)
    (2, 112, 112, [64], [1], 16, 64, True),          # 112^2 trunk
    (2, 56, 56, [96], [1], 24, 96, True),            # bottleneck 24: two 16-channel groups, the second half full
    (2, 56, 56, [96, 6, 96], [1, 0, 1], 24, 32, False),   # mimic posterior at 56^2: 6 parent channels
    (2, 28, 28, [128], [1], 32, 160, False),         # 28^2 prior: five 32-row output blocks
    (2, 14, 14, [160], [1], 40, 160, True),          # bottleneck 40: two 32-row blocks, three 16-channel groups
    (2, 14, 14, [160, 6, 160], [1, 0, 1], 40, 32, False),
    (2, 8, 8, [192], [1], 48, 192, True),            # bottleneck 48
    (2, 8, 8, [192], [1], 48, 224, False),           # the widest forward output: 2 z_dim + 192
    (2, 7, 9, [72, 8], [1, 1], 20, 56, False),       # odd sides, two gradient outputs, ragged everything
    (32, 28, 28, [128], [1], 32, 128, True),         # a full batch
]


def _run(case, fuse, seed=0):
    from causal_gen_amd.engine import ConvSite, Engine
    from causal_gen_amd import _lib

    N, H, W, segc, segrg, b, co, with_res = case
    g = torch.Generator().manual_seed(1000 * seed + H * 7 + co)
    ci = sum(segc)
    cs = [torch.nn.Conv2d(ci, b, 1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, co, 1)]
    with torch.no_grad():
        for c in cs:
            fan = c.in_channels * c.kernel_size[0] ** 2
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) * 1.6 / math.sqrt(fan))
            c.bias.copy_(torch.randn(c.out_channels, generator=g) * 0.2)
    xs = [torch.randn(N, c, H, W, generator=g).half().float() for c in segc]
    res = torch.randn(N, co, H, W, generator=g).half().float() if with_res else None
    gout = torch.randn(N, co, H, W, generator=g).half().float()
    eng = Engine("cuda", "f16")
    holder = torch.nn.ModuleList(cs).cuda()
    sites = [ConvSite("c0", holder[0], segc, [bool(r) for r in segrg], 0)]
    sites += [ConvSite(f"c{k}", holder[k], [b], [True], k) for k in (1, 2, 3)]
    for r in range(4):
        sites[r].blk4 = (r, sites)
    eng.blk4_on = 1  # (images are planned at bind time)
    eng.bind(holder, sites)
    eng.blk4_on = fuse
    eng.begin()
    eng.prepare_weights(force=True)
    eng.recording = True
    nts = [eng.from_nchw(x.cuda(), rg=bool(r)) for x, r in zip(xs, segrg)]
    rt = eng.from_nchw(res.cuda(), rg=False) if with_res else None
    n0 = eng.launches
    y = eng.block4(sites, nts, res1=rt) if fuse else None
    if y is None:
        assert not fuse, "cgen_block4 declined a shape it should serve"
        h = eng.conv(sites[0], nts, _lib.ACT_GELU)
        h = eng.conv(sites[1], [h], _lib.ACT_GELU)
        h = eng.conv(sites[2], [h], _lib.ACT_GELU)
        y = eng.conv(sites[3], [h], _lib.ACT_GELU, res1=rt)
    fwd_launches = eng.launches - n0
    y_t = eng.to_nchw(y).cpu()
    gy = eng.seed_grad(y)
    eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.recording = False
    n1 = eng.launches
    eng.backward()
    bwd_launches = eng.launches - n1
    torch.cuda.synchronize()
    gxs = [eng.to_nchw(eng.grad_read(t)).cpu() if t.rg else None for t in nts]
    pg = [eng.param_grad_view(p).cpu().clone() for c in holder for p in (c.weight, c.bias)]
    return dict(y=y_t, gx=gxs, pg=pg, fwd_launches=fwd_launches, bwd_launches=bwd_launches, xs=xs, res=res, gout=gout, convs=cs)


@pytest.mark.parametrize("case", CASES, ids=[f"{c[3]}x{c[5]}x{c[6]}@{c[1]}x{c[2]}n{c[0]}" for c in CASES])
def test_fused_block4_matches_four_launch_path_and_torch(case):
    N, H, W, segc, segrg, b, co, with_res = case
    four = _run(case, 0)
    one = _run(case, 1)
    assert four["fwd_launches"] == 4 and one["fwd_launches"] == 1, (four["fwd_launches"], one["fwd_launches"])
    assert one["bwd_launches"] < four["bwd_launches"], (one["bwd_launches"], four["bwd_launches"])
    # ---- (a) against the four-launch path: f16 flips of the intermediates at most (the fused kernel evaluates GELU from a Taylor table,
    # the conv kernels by an erf polynomial: both ~2e-7 from the exact value, so a bottleneck value within that of a rounding boundary
    # lands on the other side, and every output sums a few hundred of them: up to ~10 % of the outputs move by one ulp)
    scale = float(four["y"].abs().max())
    dy = (one["y"] - four["y"]).abs()
    assert float(dy.max()) <= 0.02 * scale and float((dy > 0).float().mean()) < 0.2, (float(dy.max()), scale, float((dy > 0).float().mean()))
    assert float((one["y"] - four["y"]).norm()) <= 1e-3 * float(four["y"].norm())
    for a, c in zip(one["gx"], four["gx"]):
        if c is None:
            assert a is None
            continue
        s = float(c.abs().max())
        assert float((a - c).abs().max()) <= 0.03 * s, (float((a - c).abs().max()), s)
        assert float((a - c).norm()) <= 5e-3 * float(c.norm())
    for a, c in zip(one["pg"], four["pg"]):
        assert float((a - c).norm()) <= 1e-2 * float(c.norm()) + 1e-6, (float((a - c).norm()), float(c.norm()))
    # ---- (b) against torch f32 (vae.py:57-71,73-84) on the same f16-quantised operands
    cs = one["convs"]
    ws = [c.weight.detach().cpu().half().float().requires_grad_(True) for c in cs]
    bs = [c.bias.detach().cpu().clone().requires_grad_(True) for c in cs]
    xr = [x.clone().requires_grad_(True) for x in one["xs"]]
    h = F.conv2d(F.gelu(torch.cat(xr, 1)), ws[0], bs[0])
    h = F.conv2d(F.gelu(h), ws[1], bs[1], padding=1)
    h = F.conv2d(F.gelu(h), ws[2], bs[2], padding=1)
    y = F.conv2d(F.gelu(h), ws[3], bs[3])
    if with_res:
        y = y + one["res"]
    y.backward(one["gout"])
    sy = float(y.detach().abs().max())
    assert float((one["y"] - y.detach()).abs().max()) <= 1.5e-2 * sy, (float((one["y"] - y.detach()).abs().max()), sy)
    assert float((one["y"] - y.detach()).norm()) <= 4e-3 * float(y.detach().norm())
    for a, x in zip(one["gx"], xr):
        if a is None:
            continue
        r = x.grad
        assert float((a - r).norm()) <= 1.2e-2 * float(r.norm()), (float((a - r).norm()), float(r.norm()))
    refs = [t.grad for wb in zip(ws, bs) for t in wb]
    for a, r in zip(one["pg"], refs):
        assert float((a.reshape(-1) - r.reshape(-1)).norm()) <= 1.5e-2 * float(r.norm()) + 1e-5, (float((a.reshape(-1) - r.reshape(-1)).norm()), float(r.norm()))


def test_block4_declines_what_it_does_not_serve():
    """b > 64, a 1x1-only Block, the f32 engine: the caller runs the four convs."""
    import ctypes as C

    from causal_gen_amd import _lib

    lib = _lib.load()
    a = _lib.Block4Args()
    a.dtype, a.n, a.h, a.w, a.nseg, a.nout, a.fwd, a.b = _lib.F16, 2, 8, 8, 1, 1, 1, 128
    assert lib.block4_supported(C.byref(a)) == 0
    a.b, a.dtype = 16, _lib.F32
    assert lib.block4_supported(C.byref(a)) == 0


def _run_two(N, H, W, b, segA, rgA, coA, segB, rgB, coB, pair, share=False, seed=0):
    """Two independent default Blocks recorded back to back (the posterior and the prior Block of a decoder layer, vae.py:240-301), one
    backward pass: with `pair` their data gradients share a launch (cgen_block4_pair)."""
    from causal_gen_amd.engine import ConvSite, Engine

    g = torch.Generator().manual_seed(91 + seed + H)
    convs, blocks, ins, gouts = [], [], [], []
    for k, (segc, co) in enumerate(((segA, coA), (segB, coB))):
        ci = sum(segc)
        cs = [torch.nn.Conv2d(ci, b, 1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, co, 1)]
        with torch.no_grad():
            for c in cs:
                c.weight.copy_(torch.randn(c.weight.shape, generator=g) * 1.6 / math.sqrt(c.in_channels * c.kernel_size[0] ** 2))
                c.bias.copy_(torch.randn(c.out_channels, generator=g) * 0.2)
        convs += cs
        ins.append([torch.randn(N, c, H, W, generator=g).half().float() for c in segc])
        gouts.append(torch.randn(N, co, H, W, generator=g).half().float())
    eng = Engine("cuda", "f16")
    eng.blk4_on, eng.blk4_pair = 1, pair
    holder = torch.nn.ModuleList(convs).cuda()
    sites = []
    for k, (segc, rg) in enumerate(((segA, rgA), (segB, rgB))):
        four = [ConvSite("b%dc0" % k, holder[4 * k], segc, [bool(r) for r in rg], 4 * k)]
        four += [ConvSite("b%dc%d" % (k, j), holder[4 * k + j], [b], [True], 4 * k + j) for j in (1, 2, 3)]
        for r in range(4):
            four[r].blk4 = (r, four)
        blocks.append(four)
        sites += four
    eng.bind(holder, sites)
    eng.begin()
    eng.prepare_weights(force=True)
    eng.recording = True
    nts, ys = [], []
    for k in (1, 0):  # tape: [B][A] -- backward() meets A first, then B
        rg = (rgA, rgB)[k]
        if share and k == 0:
            t = nts[0][1]   # A reads B's INPUT tensor: both data gradients accumulate into one buffer -> no pair
        else:
            t = [eng.from_nchw(x.cuda(), rg=bool(r)) for x, r in zip(ins[k], rg)]
            nts.append((k, t))
        y = eng.block4(blocks[k], t)
        assert y is not None
        ys.append((k, y))
    for k, y in ys:
        gy = eng.seed_grad(y)
        eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gouts[k].cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.recording = False
    eng.backward()
    torch.cuda.synchronize()
    out = []
    for k, t in nts:
        out += [eng.to_nchw(eng.grad_read(v)).cpu() for v in t if v.rg]
    out += [eng.param_grad_view(p).cpu().clone() for c in holder for p in (c.weight, c.bias)]
    return out, eng.blk4_pairs


@pytest.mark.parametrize("shape", [
    (32, 28, 28, 32, [128, 6, 128], [1, 0, 1], 32, [128], [1], 160),      # a 28^2 decoder layer of mimic224: posterior cat[h, pa, acts] | prior
    (256, 16, 16, 8, [32, 12, 32], [1, 0, 1], 32, [32, 12], [1, 0], 64),  # a 16^2 layer of morphomnist (conditional prior) at batch 256
    (4, 8, 8, 48, [192, 6, 192], [1, 0, 1], 32, [192], [1], 224),         # 8^2: bottleneck 48
    (3, 20, 28, 16, [64], [1], 96, [64, 4], [1, 0], 32),                  # ragged image
], ids=["28x28", "16x16-b256", "8x8", "ragged"])
def test_block4_pair_launch_is_bit_identical_to_two_launches(shape):
    a, pa = _run_two(*shape, pair=False)
    b, pb = _run_two(*shape, pair=True)
    assert pa == 0 and pb == 1, (pa, pb)
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())


def test_block4_blocks_that_share_an_input_are_not_paired():
    shape = (4, 16, 16, 8, [32], [1], 32, [32], [1], 64)
    a, pa = _run_two(*shape, pair=False, share=True)
    b, pb = _run_two(*shape, pair=True, share=True)
    assert pa == 0 and pb == 0
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("shape", [(4, 28, 28, 128, 32), (8, 16, 16, 32, 8), (4, 14, 14, 160, 40)], ids=["28x28", "16x16", "14x14"])
def test_fused_block4_remainder_planes(shape):
    """Residual-trunk default Blocks of an inference pass (value = hi + rem): cgen_block4 reads res1 as hi + rem and writes
    out = rn16(v), out_rem = rn16(v - out), like the four-launch path's last conv.  Two chained trunk Blocks, so the second one CONSUMES
    a remainder plane; hi + rem against an f64 reference (torch erf GELU) on the f16-quantised operands, and against the plain tensor."""
    from causal_gen_amd.engine import NT, ConvSite, Engine

    N, H, W, ci, b = shape
    g = torch.Generator().manual_seed(H + ci)
    cs = [torch.nn.Conv2d(ci, b, 1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, b, 3, padding=1), torch.nn.Conv2d(b, ci, 1)]
    with torch.no_grad():
        for c in cs:
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) * 1.6 / math.sqrt(c.in_channels * c.kernel_size[0] ** 2))
            c.bias.copy_(torch.randn(c.out_channels, generator=g) * 0.2)  # (the default init draws from the unseeded global generator)
    x = (torch.randn(N, ci, H, W, generator=g) * 3).half().float()
    eng = Engine("cuda", "f16")
    holder = torch.nn.ModuleList(cs).cuda()
    sites = [ConvSite("c0", holder[0], [ci], [True], 0)] + [ConvSite(f"c{k}", holder[k], [b], [True], k) for k in (1, 2, 3)]
    for r in range(4):
        sites[r].blk4 = (r, sites)
    eng.blk4_on = 1
    eng.bind(holder, sites)
    eng.begin()
    eng.prepare_weights(force=True)
    eng.recording = False
    assert eng.trunk_rem
    h0 = eng.from_nchw(x.cuda())
    n0 = eng.launches
    h1 = eng.block4(sites, [h0], res1=h0, trunk=True)   # produces a remainder plane
    h2 = eng.block4(sites, [h1], res1=h1, trunk=True)   # consumes one and produces one
    assert eng.launches - n0 == 2 and h1.rem and h2.rem
    planes = []
    for t in (h1, h2):
        r = NT(t.ptr + t.rem, t.n, t.h, t.w, t.c, t.sn, t.sh, t.sw, t.es, rg=False, keep=t.keep)
        planes.append((eng.to_nchw(t).double().cpu(), eng.to_nchw(r).double().cpu()))
    torch.cuda.synchronize()
    ws = [c.weight.detach().cpu().half().double() for c in cs]
    bs = [c.bias.detach().cpu().double() for c in cs]

    def act(t):  # an MFMA operand: GELU of the stored f16 value, rounded to f16
        return F.gelu(t).half().double()

    def block(hi, v):  # conv inputs read hi alone; the residual adds the full value; the bottleneck tensors are stored in f16
        t = F.conv2d(act(hi), ws[0], bs[0]).half().double()
        t = F.conv2d(act(t), ws[1], bs[1], padding=1).half().double()
        t = F.conv2d(act(t), ws[2], bs[2], padding=1).half().double()
        return v + F.conv2d(act(t), ws[3], bs[3])

    (hi1, rem1), (hi2, rem2) = planes
    v1 = block(x.double(), x.double())
    assert float((hi1.half() != (hi1 + rem1).half()).float().mean()) < 1e-3
    # hi + rem carries ~22 bits; what is left is a bottleneck value within f32 rounding of an f16 boundary taking the other side than the
    # f64 reference's (one f16 ulp of a bottleneck value times a weight, a handful of pixels): the MEAN error is held tight and against
    # the plain tensor's, the maximum loosely
    e1 = ((hi1 + rem1) - v1).abs()
    assert float(e1.mean()) <= 2e-5 * float(v1.abs().max()) and float(e1.max()) <= 1e-3 * float(v1.abs().max()), (float(e1.mean()), float(e1.max()))
    v2 = block(hi1, hi1 + rem1)
    e2, plain = ((hi2 + rem2) - v2).abs(), (hi2 - v2).abs()
    assert float(e2.mean()) <= 2e-5 * float(v2.abs().max()) and float(e2.max()) <= 1e-3 * float(v2.abs().max()), (float(e2.mean()), float(e2.max()))
    assert float(e2.mean()) * 5 < float(plain.mean()), (float(e2.mean()), float(plain.mean()))


def test_fused_block4_fuzzed_shapes():
    """Random default-Block shapes (sides 1..40, bottlenecks 4..48, one to three segments, ragged channel counts) through the kernel
    against the four-launch path: forward and every gradient."""
    import random

    rnd = random.Random(20260930)
    n_ok = 0
    for it in range(24):
        H, W = rnd.randint(1, 40), rnd.randint(1, 40)
        b = rnd.choice([4, 8, 12, 16, 20, 24, 32, 40, 48])
        nseg = rnd.randint(1, 3)
        segc = [rnd.choice([8, 16, 24, 32, 40, 64, 72, 96]) for _ in range(nseg)]
        segrg = [1] + [rnd.randint(0, 1) for _ in range(nseg - 1)]
        if nseg >= 2 and rnd.random() < 0.5:
            segc[1], segrg[1] = rnd.choice([4, 6, 12, 20]), 0  # a parents-like segment: ragged, no gradient
        with_res = rnd.random() < 0.5
        co = segc[0] if with_res else rnd.choice([8, 16, 32, 48, 64, 104, 160])
        case = (rnd.randint(1, 5), H, W, segc, segrg, b, co, with_res)
        four, one = _run(case, 0, seed=it), _run(case, 1, seed=it)
        assert one["fwd_launches"] == 1, case
        sy = float(four["y"].abs().max())
        assert float((one["y"] - four["y"]).abs().max()) <= 0.02 * sy + 1e-6, (case, float((one["y"] - four["y"]).abs().max()), sy)
        for a, c in zip(one["gx"], four["gx"]):
            if c is None:
                assert a is None
                continue
            assert float((a - c).abs().max()) <= 0.04 * float(c.abs().max()) + 1e-6, case
            assert float((a - c).norm()) <= 1e-2 * float(c.norm()) + 1e-6, case
        for a, c in zip(one["pg"], four["pg"]):
            assert float((a - c).norm()) <= 2e-2 * float(c.norm()) + 1e-5, (case, float((a - c).norm()), float(c.norm()))
        n_ok += 1
    assert n_ok == 24
