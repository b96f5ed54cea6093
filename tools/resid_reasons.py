import sys, collections
sys.path.insert(0, "/root/repo")
import torch, bench
from causal_gen_amd import engine as E
m, hp = bench.build_model("ukbb192", "f16")
m = m.cuda().train()
x, pa = bench.synth_batch("ukbb192", hp, 32, "cuda", 1)
out = m(x, pa, beta=1.0); out["elbo"].backward()
eng = m.engine()
cnt = collections.Counter()
orig = E.Engine._grad_residual
def spy(self, r, g, out, segs):
    gbuf = self.grads[id(out.base)][0]
    whole_r = r.base is r and id(r) not in self.grads
    has = id(r.base) in self.grads
    whole_g = out.base is out and g.c == out.c
    seg_clash = any(sg.base is r for sg in segs)
    adopted = id(gbuf) in self._adopted
    same = (r.n, r.h, r.w, r.c, r.sn, r.sh, r.sw) == (g.n, g.h, g.w, g.c, g.sn, g.sh, g.sw)
    cnt[(("base" if r.base is r else "view"), ("hasgrad" if has else "nograd"), ("wholeg" if whole_g else "partg"), ("segclash" if seg_clash else "-"), ("adopted" if adopted else "-"), ("same" if same else "diffgeom"), r.h)] += 1
    return orig(self, r, g, out, segs)
E.Engine._grad_residual = spy
m.zero_grad(); out = m(x, pa, beta=1.0); out["elbo"].backward(); torch.cuda.synchronize()
for k, v in cnt.most_common(): print(v, k)
