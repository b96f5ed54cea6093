import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from causal_gen_amd.train import TrainStep
cfg = sys.argv[1]
m, hp = bench.build_model(cfg, "f16", False)
m = m.cuda()
x, pa = bench.synth_batch(cfg, hp, 64, torch.device("cuda"), seed=5)
ts = TrainStep(m, hp, ema=False, use_graph=False)
torch.manual_seed(1)
for _ in range(3):
    ts.step(x, pa)   # make prior heads non-zero
torch.manual_seed(2)
m.__dict__["_eng"].rng = None
out = ts._fwd_bwd(x, pa, 1.0)
torch.cuda.synchronize()
g = ts.eng.flat_g.clone().cpu()
print("elbo", [float(v) for v in out.cpu()], "gnorm", float(g.norm()))
torch.save(g, os.environ["OUT"])
