"""Debug (2 gloo ranks on one GPU): run a few DP steps, dump rank 0's parameters.  usage: torchrun ... tools/dbg_dp.py <tag> <steps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from causal_gen_amd.train import TrainStep

tag, steps = sys.argv[1], int(sys.argv[2])
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
m, hp = bench.build_model("ukbb192", "f16")
m = m.cuda()
ts = TrainStep(m, hp, ema=False, use_graph=os.environ.get("DBG_GRAPH", "1") == "1", process_group=dist.group.WORLD)
x, pa = bench.synth_batch("ukbb192", hp, 2, "cuda", 100 + rank)
for _ in range(steps):
    ts.step(x, pa)
torch.cuda.synchronize()
if rank == 0:
    torch.save({"sd": {k: v.cpu() for k, v in m.state_dict().items()}, "g": ts.eng.flat_g.cpu(), "early": ts.early_ranges}, f"gpurun_out/dp_{tag}.pt")
dist.barrier()
dist.destroy_process_group()
