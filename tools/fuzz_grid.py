import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from test_gpu_ops import test_conv_fwd_bwd
for N, R in ((4, 24), (1, 96), (32, 24), (2, 48)):
    for ci, co in ((32, 8), (32, 32), (32, 128), (32, 160), (32, 192), (32, 256), (48, 192), (8, 32), (16, 64), (24, 96), (40, 160)):
        c = (N, R, R, [ci], co, 3, 0, False)
        try:
            test_conv_fwd_bwd(c, "f16"); print("ok  ", c, flush=True)
        except Exception as e:
            ls = str(e).splitlines()
            print("FAIL", c, ls[2] if len(ls) > 2 else ls[0], flush=True)
