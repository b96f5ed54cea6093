#!/usr/bin/env python3
"""Random conv fwd / dgrad / wgrad parity cases at model-like sizes through tests/test_gpu_ops.test_conv_fwd_bwd
(the committed test covers 14 small shapes; this sweeps the kernel-selection space: px / ws / smallp / tile / mega wgrad).
usage: python tools/fuzz_conv.py [n_cases] [seed]"""
import os
import random
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_gpu_ops import test_conv_fwd_bwd  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # generate the same sequence but execute only cases >= skip
widths = [8, 16, 24, 32, 40, 48, 64, 96, 128, 160, 192, 256]
fails = 0
for i in range(n_cases):
    res = rng.choice([1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 96])
    N = rng.choice([1, 2, 8, 32, 64, 256]) if res <= 16 else (rng.choice([1, 2, 8, 32]) if res <= 48 else rng.choice([1, 2, 4]))
    kind = rng.random()
    if kind < 0.35:       # bottleneck in: C -> C/4
        c = rng.choice(widths[3:]); segc, Co = [c], max(4, c // 4)
    elif kind < 0.65:     # bottleneck out: C/4 -> C
        c = rng.choice(widths[3:]); segc, Co = [max(4, c // 4)], c
    elif kind < 0.85:     # cat[h, pa, acts]
        c = rng.choice(widths[3:10]); segc, Co = [c, rng.choice([4, 6, 12]), c], max(8, c // 4)
    else:                 # z_proj / z_feat_proj like
        c = rng.choice(widths[3:]); segc, Co = [16, rng.choice([4, c])], c
    ks = 1 if (res <= 2 or rng.random() < 0.3) else 3
    act = rng.choice([0, 1, 1, 2])
    with_res = rng.random() < 0.4
    if res == 1:
        H = W = 1
    else:
        H, W = res, res if rng.random() < 0.8 else max(1, res - rng.choice([1, 3]))
    case = (N, H, W, segc, Co, ks, act, with_res)
    for dtype in (["f16"] if rng.random() < 0.8 else ["f16", "f32"]):
        if i < skip:
            continue
        print("run  %s %s" % (dtype, case), flush=True)
        try:
            test_conv_fwd_bwd(case, dtype)
            print("ok   %s %s" % (dtype, case), flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print("FAIL %s %s: %s" % (dtype, case, str(e).splitlines()[0][:300]), flush=True)
print("%d failures" % fails)
sys.exit(1 if fails else 0)
