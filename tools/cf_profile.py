"""Counterfactual loop alone (ukbb192, batch 32, f16, hipGraph replay) -- for `rocprofv3 --kernel-trace --stats -- python tools/cf_profile.py`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

m, hp = bench.build_model("ukbb192", "f16")
m = m.cuda().eval()
x, pa = bench.synth_batch("ukbb192", hp, 32, "cuda", 1)
out = bench.cf_leg(m, x, pa, "ukbb192", n_cf=int(os.environ.get("N_CF", "10")))
print(out)
