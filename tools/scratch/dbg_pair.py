import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_model import build, load_golden
fx = load_golden("tiny_light_c1.pt")
outs = []
for pair in ("0", "1", "1", "0"):
    os.environ["CGEN_CF_PAIR"] = pair
    m, _ = build(fx, "f16")
    x, pa = fx["x"].cuda(), fx["pa"].cuda()
    eng = m.engine()
    print("stage_res", sorted(eng.stage_res))
    eng.rng_ptr()
    eng.rng.copy_(torch.tensor([21, 0], dtype=torch.int64, device=eng.rng.device))
    with torch.no_grad():
        zs = m.abduct(x, pa)
        zs = zs[: max(1, len(zs) // 2)]
        if pair == "0":
            a, b = m.forward_latents(zs, pa), m.forward_latents(zs, pa.roll(1, 0))
        else:
            a, b = m.forward_latents_pair(zs, pa, pa.roll(1, 0))
    torch.cuda.synchronize()
    outs.append((a[0].clone(), a[1].clone(), b[0].clone(), b[1].clone()))
    print(pair, [float(t.abs().sum()) for t in outs[-1]], "launches", eng.launches, "stage", eng.stage_launches, eng.stage_ops_total)
for o in outs[1:]:
    print([float((u - v).abs().max()) for u, v in zip(outs[0], o)])
