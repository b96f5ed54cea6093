import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_block3 as T
cases = []
for (R, C, b) in ((192, 32, 8), (96, 64, 16), (48, 96, 24), (24, 128, 32)):
    cases += [(2, R, R, [C], [1], b, C, True), (2, R, R, [C, 4], [1, 0], b, C + 32, False), (2, R, R, [C], [1], b, {32: 64, 64: 96, 96: 128, 128: 160}[C], False)]
    if R <= 96:
        cases.append((2, R, R, [C, 4, C], [1, 0, 1], b, 32, False))
for c in cases:
    two = T._run(c, 0); one = T._run(c, 2)
    bad = []
    for k in ("y",):
        if not torch.isfinite(one[k]).all(): bad.append("y nonfinite")
    dy = (one["y"] - two["y"]).abs().max().item() / two["y"].abs().max().item()
    gerr = []
    for a, r in zip(one["gx"], two["gx"]):
        if a is None: continue
        if not torch.isfinite(a).all(): bad.append("gx nonfinite")
        gerr.append(((a - r).norm() / r.norm()).item())
    perr = [((a - r).norm() / (r.norm() + 1e-12)).item() for a, r in zip(one["pg"], two["pg"])]
    print(c[1], c[3], c[5], c[6], "launches", one["fwd_launches"], one["bwd_launches"], "dy %.2e" % dy, "gx", ["%.1e" % e for e in gerr], "pg", ["%.1e" % e for e in perr], bad, flush=True)
