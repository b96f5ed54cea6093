import sys
sys.path.insert(0, "/root/repo")
import torch, bench
from causal_gen_amd import dscm
for cfg, B in (("ukbb192", 4), ("morphomnist", 32), ("mimic224", 2)):
    m, hp = bench.build_model(cfg, "f16")
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
    x, pa = bench.synth_batch(cfg, hp, B, "cuda", 1)
    with torch.no_grad():
        same = dscm.counterfactual(m, x, pa, pa)
        diff = dscm.counterfactual(m, x, pa, pa.roll(1, 0))
    print(cfg, "null intervention max|cf - x| = %.3e ; real intervention mean|cf - x| = %.3e" % (float((same - x).abs().max()), float((diff - x).abs().mean())))
