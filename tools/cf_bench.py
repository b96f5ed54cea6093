#!/usr/bin/env python3
"""Counterfactual loop alone (abduct -> act -> predict, one hipGraph replay per batch) for one config and compute dtype:
    python tools/cf_bench.py [--config ukbb192] [--dtype f32] [--batch 32] [--n 6] [--prep 3]
Prints one JSON line (counterfactuals/s, executed TFLOP/s).  Meant to sit under `rocprofv3 --kernel-trace --stats`."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ukbb192")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--n", type=int, default=6)
    ap.add_argument("--prep", type=int, default=3, help="optimiser steps (f16) before the loop, so the prior heads are not zero")
    ap.add_argument("--dmol", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from causal_gen_amd.train import TrainStep

    m, hp = bench.build_model(a.config, "f16", a.dmol)
    m = m.to(dev)
    x, pa = bench.synth_batch(a.config, hp, a.batch, dev, seed=100)
    if a.prep:
        ts = TrainStep(m, hp, ema=False, use_graph=False)
        for _ in range(a.prep):
            ts.step(x, pa)
        del ts
    mm, _ = bench.build_model(a.config, a.dtype, a.dmol)
    mm = mm.to(dev)
    mm.load_state_dict(m.state_dict())
    mm.eval()
    r = bench.cf_leg(mm, x, pa, a.config, n_cf=a.n)
    r.update(config=a.config, dtype=a.dtype, batch=a.batch, cf_mode=os.environ.get("CGEN_F32_SPLIT", ""))
    print(json.dumps(r))


if __name__ == "__main__":
    main()
