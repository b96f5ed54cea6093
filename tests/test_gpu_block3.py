"""Fused light-Block kernel of round 4 (csrc/block.hip, cgen_block3: two 3x3 convs per launch, bottleneck in LDS, forward and
data gradient) against (a) the two-launch HIP path it replaces -- same binary16 storage points, so the results agree to an f16 ulp
here and there (the f32 summation order inside a conv differs; a bottleneck value within rounding of an f16 boundary may round
the other way) -- and (b) a torch f32 reference of vae.py:60-71,73-84 on the same f16-quantised operands."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, H, W, input segments, differentiable?, bottleneck, out channels, residual?)  -- the light-Block shapes of ukbb192 and friends
CASES = [
    (2, 40, 40, [32], [1], 8, 32, True),          # 192^2 trunk widths, ragged tiles (40 = 2.5 tiles wide)
    (2, 24, 48, [32], [1], 8, 64, False),         # prior / down-block widths at 192^2
    (2, 96, 96, [64], [1], 16, 64, True),         # 96^2 trunk
    (3, 24, 24, [64], [1], 16, 96, False),        # 96^2 prior (3 channel pairs)
    (4, 48, 48, [96], [1], 24, 96, True),         # 48^2 trunk
    (2, 48, 48, [96], [1], 24, 128, False),
    (8, 24, 24, [128], [1], 32, 128, True),       # 24^2 trunk
    (4, 24, 24, [128], [1], 32, 160, False),      # 24^2 prior: five pairs -> a wave takes two
    (2, 32, 32, [32, 32], [1, 1], 16, 64, False),        # two differentiable segments (two gradient outputs)
    (2, 32, 32, [24, 8, 32], [1, 0, 1], 16, 32, False),  # three segments, 8-channel middle one without a gradient
    (2, 48, 48, [96, 4, 96], [1, 0, 1], 24, 32, False),  # the posterior Block at 48^2: cat[h, pa, acts], 4 parent channels (zero padded to 8)
    (2, 20, 28, [64, 4], [1, 0], 16, 96, False),         # the prior Block: cat[h, pa]; ragged image
    # batch 32 at 48^2: 576 eight-row tiles would be two rounds, so the kernel takes its TWELVE-row tiles (forward and the trunk's
    # data gradient: three chunks through a two-slot ring; the posterior's data gradient: one chunk, two gradient outputs)
    (32, 48, 48, [96], [1], 24, 96, True),
    (32, 48, 48, [96, 4, 96], [1, 0, 1], 24, 32, False),
    (32, 44, 48, [64], [1], 16, 96, False),              # twelve-row tiles on a ragged image (44 = 3.67 tiles high), 16-wide bottleneck
    # several twelve-row tiles per workgroup (more tiles than resident workgroups): the next tile's requests go out at the end of a
    # tile; three chunks through the two-slot ring (one chunk ahead) / two chunks
    (64, 48, 48, [96], [1], 24, 96, True),
    (32, 96, 96, [64], [1], 16, 96, False),
]


# the small-image instance (blk3s: images up to 14 pixels wide, bottlenecks up to 64 channels; one workgroup per row strip of an image)
CASES += [
    (32, 12, 12, [160], [1], 40, 160, True),               # 12^2 trunk: bottleneck 40 = two 32-row blocks, 4 strips of 3 rows
    (32, 12, 12, [160, 4, 160], [1, 0, 1], 40, 32, False),  # 12^2 posterior
    (8, 12, 12, [160, 4], [1, 0], 40, 192, False),         # 12^2 prior
    (32, 6, 6, [192], [1], 48, 192, True),                 # 6^2 trunk: one workgroup per image
    (32, 6, 6, [192, 4, 192], [1, 0, 1], 48, 32, False),   # 6^2 posterior
    (4, 6, 6, [192], [1], 48, 512, False),                 # the widest output (encoder down-block at 6^2)
    (3, 14, 14, [64], [1], 16, 64, True),                  # 14 wide: strips of 2 rows; one 32-row block
    (5, 8, 8, [40], [1], 8, 48, True),                     # ragged chunk (40 = 32 + 8 channels), 8-wide bottleneck
    (2, 12, 10, [160], [1], 40, 160, True),                # not square: strips of 4 rows
    (2, 7, 5, [72, 8], [1, 1], 24, 56, False),             # odd sides, two gradient outputs, ragged everything
]


# the row-streaming instance (blk3r: >= 48 pixels wide, one segment of 32 / 64 channels, bottleneck 8 / 16, 32 / 64 out)
CASES += [
    (2, 192, 192, [32], [1], 8, 32, True),     # 192^2 trunk: residual read from the input ring
    (2, 192, 192, [32], [1], 8, 64, False),    # 192^2 prior / down Block: two channel pairs
    (4, 96, 96, [64], [1], 16, 64, True),      # 96^2 trunk: two chunks, 16-wide bottleneck
    (3, 52, 70, [32], [1], 8, 32, True),       # ragged: 70 = 2.19 strips wide, 52 rows = 13 bands
    (2, 100, 60, [64], [1], 16, 32, False),    # ragged, 64 -> 16 -> 32
    (2, 18, 48, [32], [1], 16, 64, False),     # a strip of five bands with a partial last one; 32 -> 16 -> 64
]


def _run(case, fuse, seed=0):
    from causal_gen_amd.engine import ConvSite, Engine

    N, H, W, segc, segrg, b, co, with_res = case
    g = torch.Generator().manual_seed(1000 * seed + H * 7 + co)
    ci = sum(segc)
    c1 = torch.nn.Conv2d(ci, b, 3, padding=1)
    c2 = torch.nn.Conv2d(b, co, 3, padding=1)
    with torch.no_grad():
        c1.weight.copy_(torch.randn(c1.weight.shape, generator=g) / math.sqrt(ci * 9 / 2))
        c2.weight.copy_(torch.randn(c2.weight.shape, generator=g) / math.sqrt(b * 9 / 2))
        c1.bias.copy_(torch.randn(b, generator=g) * 0.2)
        c2.bias.copy_(torch.randn(co, generator=g) * 0.2)
    xs = [torch.randn(N, c, H, W, generator=g).half().float() for c in segc]
    res = torch.randn(N, co, H, W, generator=g).half().float() if with_res else None
    gout = torch.randn(N, co, H, W, generator=g).half().float()
    eng = Engine("cuda", "f16")
    eng.blk3_on, eng.blk3_minres = fuse, 8
    eng.blk3_res, eng.blk3_res3 = [], []
    holder = torch.nn.ModuleList([c1, c2]).cuda()
    s1 = ConvSite("c1", holder[0], segc, [bool(r) for r in segrg], 0)
    s2 = ConvSite("c2", holder[1], [b], [True], 1)
    s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
    eng.blk3_on = 2  # (images are planned at bind time)
    eng.bind(holder, [s1, s2])
    eng.blk3_on = fuse
    eng.begin()
    eng.prepare_weights(force=True)
    eng.recording = True
    nts = [eng.from_nchw(x.cuda(), rg=bool(r)) for x, r in zip(xs, segrg)]
    rt = eng.from_nchw(res.cuda(), rg=False) if with_res else None
    n0 = eng.launches
    y = eng.block2(s1, s2, nts, 1, res1=rt)
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
    pg = [eng.param_grad_view(p).cpu().clone() for p in (s1.conv.weight, s1.conv.bias, s2.conv.weight, s2.conv.bias)]
    return dict(y=y_t, gx=gxs, pg=pg, fwd_launches=fwd_launches, bwd_launches=bwd_launches, xs=xs, res=res, gout=gout, convs=(c1, c2))


@pytest.mark.parametrize("case", CASES, ids=[f"{c[3]}x{c[5]}x{c[6]}@{c[1]}x{c[2]}" for c in CASES])
def test_fused_block3_matches_two_launch_path_and_torch(case):
    N, H, W, segc, segrg, b, co, with_res = case
    two = _run(case, 0)
    one = _run(case, 2)
    assert two["fwd_launches"] == 2 and one["fwd_launches"] == 1, (two["fwd_launches"], one["fwd_launches"])
    assert one["bwd_launches"] < two["bwd_launches"], (one["bwd_launches"], two["bwd_launches"])
    # ---- (a) against the two-launch path: a handful of f16 flips at most
    scale = float(two["y"].abs().max())
    dy = (one["y"] - two["y"]).abs()
    assert float(dy.max()) <= 0.02 * scale and float((dy > 0).float().mean()) < 0.05, (float(dy.max()), scale, float((dy > 0).float().mean()))
    for a, c in zip(one["gx"], two["gx"]):
        if c is None:
            assert a is None
            continue
        s = float(c.abs().max())
        assert float((a - c).abs().max()) <= 0.03 * s, (float((a - c).abs().max()), s)
        assert float((a - c).norm()) <= 5e-3 * float(c.norm())
    for a, c in zip(one["pg"], two["pg"]):
        assert float((a - c).norm()) <= 1e-2 * float(c.norm()) + 1e-6, (float((a - c).norm()), float(c.norm()))
    # ---- (b) against torch f32 on the same f16-quantised operands
    c1, c2 = one["convs"]
    w1, w2 = c1.weight.detach().cpu().half().float().requires_grad_(True), c2.weight.detach().cpu().half().float().requires_grad_(True)
    xr = [x.clone().requires_grad_(True) for x in one["xs"]]
    t = F.conv2d(F.relu(torch.cat(xr, 1)), w1, c1.bias.detach().cpu(), padding=1)
    y = F.conv2d(F.relu(t), w2, c2.bias.detach().cpu(), padding=1)
    if with_res:
        y = y + one["res"]
    y.backward(one["gout"])
    tol = dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(one["y"], y.detach(), **tol)
    for a, x in zip(one["gx"], xr):
        if a is not None:
            assert float((a - x.grad).norm()) <= 2e-2 * float(x.grad.norm())
    assert float((one["pg"][0] - w1.grad).norm()) <= 3e-2 * float(w1.grad.norm())
    assert float((one["pg"][2] - w2.grad).norm()) <= 3e-2 * float(w2.grad.norm())


@pytest.mark.parametrize("shape", [(4, 48, 48, 96, 24), (8, 24, 24, 128, 32), (32, 48, 48, 96, 24), (8, 12, 12, 160, 40), (8, 6, 6, 192, 48)],
                         ids=["48x48", "24x24", "48x48-twelve-row", "12x12-small", "6x6-small"])
def test_fused_block3_remainder_planes(shape):
    """Residual-trunk Blocks of an inference pass (value = hi + rem, remainder planes): the fused kernel reads res1 as
    hi + rem and writes out = rn16(v), out_rem = rn16(v - out), like the two-launch path's second conv (cgen_conv_args.out_rem /
    res1_rem).  Two chained trunk Blocks, so the second one CONSUMES a remainder plane; hi + rem of the fused path against the
    two-launch path and against an f64 reference on the f16-quantised operands."""
    from causal_gen_amd.engine import NT, ConvSite, Engine

    N, H, W, ci, b = shape
    g = torch.Generator().manual_seed(H + ci)
    c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, ci, 3, padding=1)
    with torch.no_grad():
        c1.weight.copy_(torch.randn(c1.weight.shape, generator=g) / math.sqrt(ci * 9 / 2))
        c2.weight.copy_(torch.randn(c2.weight.shape, generator=g) / math.sqrt(b * 9 / 2))
    x = (torch.randn(N, ci, H, W, generator=g) * 3).half().float()
    outs = {}
    for fuse in (0, 2):
        eng = Engine("cuda", "f16")
        eng.blk3_minres, eng.blk3_res, eng.blk3_res3 = 8, [], []
        holder = torch.nn.ModuleList([c1, c2]).cuda()
        s1, s2 = ConvSite("c1", holder[0], [ci], [True], 0), ConvSite("c2", holder[1], [b], [True], 1)
        s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
        eng.blk3_on = 2
        eng.bind(holder, [s1, s2])
        eng.blk3_on = fuse
        eng.begin()
        eng.prepare_weights(force=True)
        eng.recording = False
        assert eng.trunk_rem
        h0 = eng.from_nchw(x.cuda())
        n0 = eng.launches
        h1 = eng.block2(s1, s2, [h0], 1, res1=h0, trunk=True)   # produces a remainder plane
        h2 = eng.block2(s1, s2, [h1], 1, res1=h1, trunk=True)   # consumes one and produces one
        assert eng.launches - n0 == (2 if fuse else 4) and h1.rem and h2.rem
        planes = []
        for t in (h1, h2):
            r = NT(t.ptr + t.rem, t.n, t.h, t.w, t.c, t.sn, t.sh, t.sw, t.es, rg=False, keep=t.keep)
            planes.append((eng.to_nchw(t).double().cpu(), eng.to_nchw(r).double().cpu()))
        torch.cuda.synchronize()
        outs[fuse] = planes
    w1, w2 = c1.weight.detach().cpu().half().double(), c2.weight.detach().cpu().half().double()
    b1, b2 = c1.bias.detach().cpu().double(), c2.bias.detach().cpu().double()

    def block(hi, v):  # conv inputs read hi alone; the residual adds the full value; the bottleneck is stored in f16
        t = F.conv2d(F.relu(hi), w1, b1, padding=1).half().double()
        return v + F.conv2d(F.relu(t), w2, b2, padding=1)

    for fuse in (0, 2):
        (hi1, rem1), (hi2, rem2) = outs[fuse]
        v1 = block(x.double(), x.double())
        # hi is the f16 rounding of hi + rem (but for ties: a remainder that rounds UP to exactly half an ulp)
        assert float((hi1.half() != (hi1 + rem1).half()).float().mean()) < 1e-3
        # hi + rem carries ~22 bits; what is left is a bottleneck value within f32 rounding of an f16 boundary taking the other
        # side than the f64 reference's (one f16 ulp of t times a weight: ~1e-3, a handful of pixels) -- so the MEAN error is held
        # tight and against the plain tensor's, the maximum loosely
        e1 = ((hi1 + rem1) - v1).abs()
        assert float(e1.mean()) <= 2e-6 * float(v1.abs().max()) and float(e1.max()) <= 4e-3, (fuse, float(e1.mean()), float(e1.max()))
        v2 = block(hi1, hi1 + rem1)
        e2, plain = ((hi2 + rem2) - v2).abs(), (hi2 - v2).abs()
        assert float(e2.mean()) <= 2e-6 * float(v2.abs().max()) and float(e2.max()) <= 4e-3, (fuse, float(e2.mean()), float(e2.max()))
        assert float(e2.mean()) * 20 < float(plain.mean()), (fuse, float(e2.mean()), float(plain.mean()))
    d = float(((outs[2][1][0] + outs[2][1][1]) - (outs[0][1][0] + outs[0][1][1])).abs().max())
    assert d <= 0.02 * float(outs[0][1][0].abs().max()), d  # (a bottleneck value within rounding of an f16 boundary may round the other way)


def _run_two(N, H, W, b, segA, rgA, coA, segB, rgB, coB, pair, seed=0, fuse=2, chain=False, share=False):
    """Two independent light Blocks recorded back to back (the posterior and the prior Block of a decoder layer, vae.py:240-301), one
    backward pass: with `pair` their data gradients share a launch (cgen_block3_pair)."""
    from causal_gen_amd.engine import ConvSite, Engine

    g = torch.Generator().manual_seed(77 + seed + H)
    convs, sites, ins, gouts = [], [], [], []
    for k, (segc, co) in enumerate(((segA, coA), (segB, coB))):
        ci = sum(segc)
        c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, co, 3, padding=1)
        with torch.no_grad():
            c1.weight.copy_(torch.randn(c1.weight.shape, generator=g) / math.sqrt(ci * 9 / 2))
            c2.weight.copy_(torch.randn(c2.weight.shape, generator=g) / math.sqrt(b * 9 / 2))
            c1.bias.copy_(torch.randn(b, generator=g) * 0.2)
            c2.bias.copy_(torch.randn(co, generator=g) * 0.2)
        convs += [c1, c2]
        ins.append([torch.randn(N, c, H, W, generator=g).half().float() for c in segc])
        gouts.append(torch.randn(N, co, H, W, generator=g).half().float())
    eng = Engine("cuda", "f16")
    eng.blk3_on, eng.blk3_minres, eng.blk3_res, eng.blk3_res3 = 2, 8, [], []
    eng.blk3_pair = eng.conv_pair = pair
    holder = torch.nn.ModuleList(convs).cuda()
    for k, (segc, rg) in enumerate(((segA, rgA), (segB, rgB))):
        s1 = ConvSite("c%da" % k, holder[2 * k], segc, [bool(r) for r in rg], 2 * k)
        s2 = ConvSite("c%db" % k, holder[2 * k + 1], [b], [True], 2 * k + 1)
        s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
        sites += [s1, s2]
    eng.bind(holder, sites)
    eng.blk3_on = fuse
    eng.begin()
    eng.prepare_weights(force=True)
    eng.recording = True
    nts, ys = [], []
    for k in (1, 0):  # tape: [B][A] -- backward() meets A first, then B
        rg = (rgA, rgB)[k]
        if chain and k == 0:
            t = [ys[0][1]]  # A reads B's output: the two Blocks are NOT independent in the backward pass
        elif share and k == 0:
            t = nts[0][1]   # A reads B's INPUT tensor: both data gradients accumulate into one buffer
        else:
            t = [eng.from_nchw(x.cuda(), rg=bool(r)) for x, r in zip(ins[k], rg)]
            nts.append((k, t))
        ys.append((k, eng.block2(sites[2 * k], sites[2 * k + 1], t, 1)))
    for k, y in (ys[1:] if chain else ys):
        gy = eng.seed_grad(y)
        eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gouts[k].cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
    eng.recording = False
    eng.backward()
    torch.cuda.synchronize()
    out = []
    for k, t in nts:
        out += [eng.to_nchw(eng.grad_read(v)).cpu() for v in t if v.rg]
    out += [eng.param_grad_view(p).cpu().clone() for c in holder for p in (c.weight, c.bias)]
    return out, (eng.blk3_pairs if fuse else eng.conv_pairs)


@pytest.mark.parametrize("shape", [
    (2, 48, 48, 24, [96, 4, 96], [1, 0, 1], 32, [96, 4], [1, 0], 128),    # a 48^2 decoder layer at batch 2: posterior cat[h, pa, acts] | prior cat[z, pa]
    (32, 48, 48, 24, [96, 4, 96], [1, 0, 1], 32, [96, 4], [1, 0], 128),   # ... at the bench batch: twelve-row tiles, 768 workgroups for 512 slots
    (8, 24, 24, 32, [128, 4, 128], [1, 0, 1], 32, [128], [1], 160),       # 24^2
    (3, 20, 28, 16, [64], [1], 96, [64, 4], [1, 0], 32),                   # ragged image, different output classes (no pair: two launches)
    (32, 12, 12, 40, [160, 4, 160], [1, 0, 1], 32, [160, 4], [1, 0], 192),  # 12^2 decoder layer: two small-image problems of 128 workgroups
    (32, 6, 6, 48, [192, 4, 192], [1, 0, 1], 32, [192], [1], 224),          # 6^2
], ids=["48x48-b2", "48x48-b32", "24x24", "mismatch", "12x12-small", "6x6-small"])
def test_block3_pair_launch_is_bit_identical_to_two_launches(shape):
    a, pa = _run_two(*shape, pair=False)
    b, pb = _run_two(*shape, pair=True)
    assert pa == 0
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())
    print("pair launches:", pb)
    if shape[0] != 3:
        assert pb == 1


@pytest.mark.parametrize("shape", [
    (32, 12, 12, 40, [160, 4, 160], [1, 0, 1], 32, [160, 4], [1, 0], 192),   # a 12^2 decoder layer: bottleneck 40 (not served fused)
    (32, 6, 6, 48, [192, 4, 192], [1, 0, 1], 32, [192], [1], 224),           # 6^2
    (4, 1, 1, 32, [128, 4, 128], [1, 0, 1], 32, [128, 4], [1, 0], 160),      # 1x1 images (centre tap only)
], ids=["12x12", "6x6", "1x1"])
def test_small_image_conv_pair_launches_are_bit_identical_to_single_launches(shape):
    """Unfused posterior / prior Blocks on the small-image conv path: conv2 of one Block with conv2 of the other, and the two gradient
    outputs of the posterior's conv1, share launches (cgen_conv2d_pair); same bits as five single launches."""
    a, pa = _run_two(*shape, pair=False, fuse=0)
    b, pb = _run_two(*shape, pair=True, fuse=0)
    assert pa == 0 and pb == 2, (pa, pb)
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())


def test_dependent_unfused_blocks_are_not_reordered_for_a_pair_launch():
    """Block A consumes Block B's output (consecutive trunk Blocks without residuals): the backward tape shows the same four-conv
    pattern as a posterior / prior pair, but B.conv2's gradient comes from A.conv1 -- backward() must leave the order alone."""
    shape = (8, 12, 12, 40, [160], [1], 160, [160], [1], 160)
    a, pa = _run_two(*shape, pair=False, fuse=0, chain=True)
    b, pb = _run_two(*shape, pair=True, fuse=0, chain=True)
    assert pa == 0 and pb == 0, (pa, pb)
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())


def test_dependent_fused_blocks_do_not_share_a_launch():
    """... and the fused form: A's data gradient WRITES the gradient B's reads, so the held launch must go out on its own."""
    shape = (8, 24, 24, 32, [128], [1], 128, [128], [1], 128)
    a, pa = _run_two(*shape, pair=False, fuse=2, chain=True)
    b, pb = _run_two(*shape, pair=True, fuse=2, chain=True)
    assert pa == 0 and pb == 0, (pa, pb)
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())


@pytest.mark.parametrize("fuse,shape", [(2, (8, 24, 24, 32, [128], [1], 128, [128], [1], 160)), (0, (8, 12, 12, 40, [160], [1], 160, [160], [1], 192))],
                         ids=["fused", "unfused"])
def test_blocks_sharing_an_input_keep_their_order(fuse, shape):
    """Two Blocks read the same tensor (a decoder layer with q_correction: the prior reads h like the posterior): their data gradients
    accumulate into ONE buffer, the second one's bookkeeping may copy or extend it -- no held launch, same bits."""
    a, pa = _run_two(*shape, pair=False, fuse=fuse, share=True)
    b, pb = _run_two(*shape, pair=True, fuse=fuse, share=True)
    assert pa == 0 and pb == 0, (pa, pb)
    for x, y in zip(a, b):
        assert torch.equal(x, y), float((x - y).abs().max())


@pytest.mark.parametrize("shape", [(2, 192, 192, 32, 8), (3, 96, 96, 64, 16), (2, 52, 70, 32, 8)], ids=["192x192", "96x96", "ragged"])
def test_row_streaming_block_reads_the_residual_from_its_input_ring(shape):
    """A trunk Block (out = x + f(x): the residual IS the input, vae.py:73-78) on the row-streaming instance: the residual comes from
    the LDS ring of input rows (RES = 1), not from HBM.  Forward and data gradient against the two-launch path."""
    from causal_gen_amd.engine import ConvSite, Engine

    N, H, W, ci, b = shape
    g = torch.Generator().manual_seed(H + ci)
    c1, c2 = torch.nn.Conv2d(ci, b, 3, padding=1), torch.nn.Conv2d(b, ci, 3, padding=1)
    with torch.no_grad():
        c1.weight.copy_(torch.randn(c1.weight.shape, generator=g) / math.sqrt(ci * 9 / 2))
        c2.weight.copy_(torch.randn(c2.weight.shape, generator=g) / math.sqrt(b * 9 / 2))
    x = torch.randn(N, ci, H, W, generator=g).half().float()
    gout = torch.randn(N, ci, H, W, generator=g).half().float()
    outs = {}
    for fuse in (0, 2):
        eng = Engine("cuda", "f16")
        eng.blk3_minres, eng.blk3_res, eng.blk3_res3 = 8, [], []
        holder = torch.nn.ModuleList([c1, c2]).cuda()
        s1, s2 = ConvSite("c1", holder[0], [ci], [True], 0), ConvSite("c2", holder[1], [b], [True], 1)
        s1.blk3, s2.blk3 = ("a", s2), ("b", s1)
        eng.blk3_on = 2
        eng.bind(holder, [s1, s2])
        eng.blk3_on = fuse
        eng.begin()
        eng.prepare_weights(force=True)
        eng.recording = True
        xt = eng.from_nchw(x.cuda(), rg=True)
        n0 = eng.launches
        y = eng.block2(s1, s2, [xt], 1, res1=xt, trunk=True)
        assert eng.launches - n0 == (1 if fuse else 2)
        yt = eng.to_nchw(y).cpu()
        gy = eng.seed_grad(y)
        eng.lib.axpby(eng.dt, N, H, W, eng.from_nchw(gout.cuda()).cv(), gy.cv(), 1.0, 1.0, 1 << 30, 0, eng.stream)
        eng.recording = False
        eng.backward()
        torch.cuda.synchronize()
        outs[fuse] = (yt, eng.to_nchw(eng.grad_read(xt)).cpu())
    (y0, g0), (y2, g2) = outs[0], outs[2]
    sy, sg = float(y0.abs().max()), float(g0.abs().max())
    assert float((y2 - y0).abs().max()) <= 0.02 * sy and float(((y2 - y0).abs() > 0).float().mean()) < 0.05
    assert float((g2 - g0).abs().max()) <= 0.03 * sg and float((g2 - g0).norm()) <= 5e-3 * float(g0.norm())
    w1, w2 = c1.weight.detach().cpu().half().float(), c2.weight.detach().cpu().half().float()
    ref = x + F.conv2d(F.relu(F.conv2d(F.relu(x), w1, c1.bias.detach().cpu(), padding=1).half().float()), w2, c2.bias.detach().cpu(), padding=1)
    torch.testing.assert_close(y2, ref, rtol=3e-2, atol=3e-2)
