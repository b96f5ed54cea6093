"""Random default-Block shapes through cgen_block4 against the four-launch path (the committed test's comparison, more cases):
    python tools/fuzz_blk4.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_block4 import _run

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
worst = 0.0
for it in range(n):
    H, W = rnd.randint(1, 70), rnd.randint(1, 70)
    b = rnd.choice([4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 60, 64])
    nseg = rnd.randint(1, 3)
    segc = [rnd.choice([8, 16, 24, 32, 40, 56, 64, 72, 96, 128, 200]) for _ in range(nseg)]
    segrg = [1] + [rnd.randint(0, 1) for _ in range(nseg - 1)]
    if nseg >= 2 and rnd.random() < 0.5:
        segc[1], segrg[1] = rnd.choice([4, 6, 12, 20]), 0
    with_res = rnd.random() < 0.5
    co = segc[0] if with_res else rnd.choice([8, 16, 32, 48, 64, 104, 160, 224, 256])
    case = (rnd.randint(1, 6), H, W, segc, segrg, b, co, with_res)
    four, one = _run(case, 0, seed=it), _run(case, 1, seed=it)
    assert one["fwd_launches"] == 1, ("declined", case)
    sy = float(four["y"].abs().max())
    dy = float((one["y"] - four["y"]).abs().max())
    assert dy <= 0.02 * sy + 1e-6, (case, dy, sy)
    for a, c in zip(one["gx"], four["gx"]):
        if c is None:
            continue
        assert float((a - c).norm()) <= 1e-2 * float(c.norm()) + 1e-6, (case, float((a - c).norm()), float(c.norm()))
    for a, c in zip(one["pg"], four["pg"]):
        assert float((a - c).norm()) <= 2e-2 * float(c.norm()) + 1e-5, (case, float((a - c).norm()), float(c.norm()))
    worst = max(worst, dy / (sy + 1e-9))
print("%d cases ok, worst forward deviation %.2e of the scale" % (n, worst))
