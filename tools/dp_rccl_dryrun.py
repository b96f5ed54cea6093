"""One-GPU dry run of the data-parallel train step THROUGH RCCL: a 1-rank `nccl` process group and CGEN_DP_FORCE=1 switch on
everything a multi-GPU run uses -- communication stream, bucketed asynchronous all-reduces of flat-gradient ranges, the
backward hipGraph cut in two around the early exchange, the optimiser as a third graph -- with real RCCL calls, and the result
must equal the single-GPU step bit for bit (the mean over one rank is the identity).  What it cannot show is bandwidth.
usage: python tools/dp_rccl_dryrun.py [config] [batch] [steps]      (prints PASS / FAIL; exit code 1 on failure)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

from causal_gen_amd import vae
from causal_gen_amd.hps import setup_hparams
from causal_gen_amd.train import TrainStep

name = sys.argv[1] if len(sys.argv) > 1 else "morphomnist"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hp = setup_hparams(name)
hp.lr_warmup_steps = 2


def run(dp_on, overlap):
    os.environ["CGEN_DP_FORCE"] = "1" if dp_on else "0"
    os.environ["CGEN_DP_OVERLAP"] = "1" if overlap else "0"
    torch.manual_seed(0)
    m = vae.HVAE(hp).cuda()
    m.compute_dtype = "f16"
    ts = TrainStep(m, hp, ema=True, process_group=dist.group.WORLD if dp_on else None)
    g = torch.Generator().manual_seed(1)
    outs = []
    for it in range(steps):
        x = torch.randint(0, 256, (B, hp.input_channels, hp.input_res, hp.input_res), generator=g, dtype=torch.uint8).cuda()
        pa = torch.randn(B, hp.context_dim, generator=g).cuda()[..., None, None].expand(-1, -1, hp.input_res, hp.input_res)
        torch.manual_seed(100 + it)  # (the drop_cond draw of a conditional prior)
        outs.append(ts.step(x, pa).clone())
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in m.parameters()]).clone(), torch.stack(outs), ts


dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ref_p, ref_o, _ = run(False, False)
ok = True
for overlap in (False, True):
    p, o, ts = run(True, overlap)
    same = torch.equal(p, ref_p) and torch.equal(o, ref_o)
    graphs = [k for k in ts.graphs]
    split = any(v[5] is not None for v in ts.graphs.values())
    print("dp forced, overlap %d: parameters %s the single-GPU step after %d steps; graphs %d, backward graph split %s; |p| %.6f"
          % (overlap, "EQUAL" if same else "DIFFER from", steps, len(graphs), split, float(p.abs().sum())), flush=True)
    ok = ok and same and (split or not overlap)
dist.destroy_process_group()
print("PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
