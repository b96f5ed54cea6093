import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_block3 as T
c = (2, 96, 96, [64, 4, 64], [1, 0, 1], 16, 32, False)
two = T._run(c, 0); one = T._run(c, 2)
e = (one["pg"][1] - two["pg"][1]) / two["pg"][1].abs().max()
print("bias-grad rel err per bottleneck channel:", ["%.1e" % v for v in e.tolist()])
d = (one["gx"][0] - two["gx"][0]).abs()
print("gx0 max err", d.max().item(), "ref max", two["gx"][0].abs().max().item())
bad = (d > 0.02 * two["gx"][0].abs().max()).nonzero()
print("bad count", bad.shape[0])
import collections
cy = collections.Counter((int(b[2]) % 8 for b in bad)); cx = collections.Counter((int(b[3]) % 16 for b in bad)); cn = collections.Counter((int(b[0]) for b in bad))
print("by y%8", sorted(cy.items())); print("by x%16", sorted(cx.items())); print("by n", sorted(cn.items()))
ty = collections.Counter((int(b[2]) // 8 for b in bad)); tx = collections.Counter((int(b[3]) // 16 for b in bad))
print("tile rows", sorted(ty.items())); print("tile cols", sorted(tx.items()))
