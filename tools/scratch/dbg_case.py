import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch
import test_gpu_block3 as T
case = (32, 12, 12, [160, 4, 160], [1, 0, 1], 40, 32, False)
two = T._run(case, 0); one = T._run(case, 2)
print("launches", one["fwd_launches"], one["bwd_launches"], two["bwd_launches"])
d = (one["y"] - two["y"]).abs()
sc = float(two["y"].abs().max())
print("y max diff", float(d.max()), "scale", sc, "frac>0", float((d > 0).float().mean()))
bad = d > 0.02 * sc
print("bad per image:", [int(bad[i].sum()) for i in range(32)])
print("bad per row:", [int(bad[:, :, r].sum()) for r in range(12)])
print("bad per ch:", [int(bad[:, c].sum()) for c in range(32)])
for a, c in zip(one["gx"], two["gx"]):
    if c is None: continue
    dd = (a - c).abs(); s = float(c.abs().max())
    b2 = dd > 0.03 * s
    print("gx diff", float(dd.max()), s, "bad rows", [int(b2[:, :, r].sum()) for r in range(12)], "bad ch blocks", [int(b2[:, 32*i:32*i+32].sum()) for i in range(5)])
for a, c in zip(one["pg"], two["pg"]):
    print("pg rel", float((a - c).norm() / (c.norm() + 1e-9)))
