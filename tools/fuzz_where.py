import sys, math
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from test_gpu_ops import make_engine, nhwc_to_torch
N, H, W, segc, Co, ks = (int(v) for v in sys.argv[1:7]) if len(sys.argv) > 6 else (1, 96, 96, 32, 8, 3)
segc = [segc]
g = torch.Generator().manual_seed(5)
conv = torch.nn.Conv2d(sum(segc), Co, ks, padding=ks // 2)
with torch.no_grad():
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / math.sqrt(sum(segc) * ks * ks))
    conv.bias.zero_()
x = torch.randn(N, segc[0], H, W, generator=g).half().float()
y_ref = F.conv2d(x, conv.weight.detach().half().float(), None, padding=ks // 2)
eng, (site,) = make_engine([conv], [segc], "f16")
y = nhwc_to_torch(eng, eng.conv(site, [eng.from_nchw(x.cuda())], 0))
bad = ((y - y_ref).abs() > 0.05).any(dim=1)  # [N,H,W]
idx = bad.nonzero()
print("wrong pixels %d of %d" % (bad.sum(), bad.numel()))
for n in sorted(set(idx[:, 0].tolist()))[:4]:
    m = bad[n]
    ys = sorted(set(m.nonzero()[:, 0].tolist())); xs = sorted(set(m.nonzero()[:, 1].tolist()))
    print(" n=%d rows %s cols %s" % (n, ys, xs))
    # per 8x16 tile count
    t = m.float().view(H // 8 if H % 8 == 0 else 1, -1)
    tiles = {}
    for yy, xx in m.nonzero().tolist():
        tiles[(yy // 8, xx // 16)] = tiles.get((yy // 8, xx // 16), 0) + 1
    print("   tiles (ty,tx):count", sorted(tiles.items())[:20])
