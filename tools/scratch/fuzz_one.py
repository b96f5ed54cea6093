import sys, ast
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from test_gpu_ops import test_conv_fwd_bwd
case = ast.literal_eval(sys.argv[1]); dtypes = sys.argv[2].split(","); n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for i in range(n):
    for dt in dtypes:
        test_conv_fwd_bwd(case, dt)
    print("ok", i, flush=True)
