"""Do two INDEPENDENT half-batch train steps on two streams overlap?  (aggregate images/s of two B/2 models replayed
concurrently vs one model at B)   usage: python tools/twin_probe.py [B] [k]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from causal_gen_amd.train import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda")
def mk(b, seed):
    m, hp = bench.build_model("ukbb192", "f16")
    m = m.to(dev)
    ts = TrainStep(m, hp, ema=True, use_graph=True)
    x, pa = bench.synth_batch("ukbb192", hp, b, dev, seed)
    return ts, x, pa
def run(sets, steps=20):
    streams = [torch.cuda.Stream() for _ in sets]
    for _ in range(8):
        for (ts, x, pa), st in zip(sets, streams):
            with torch.cuda.stream(st):
                ts.step(x, pa)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (ts, x, pa), st in zip(sets, streams):
            with torch.cuda.stream(st):
                ts.step(x, pa)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(s[1].shape[0] for s in sets) * steps / dt, 1e3 * dt / steps
one = [mk(B, 100)]
print("one model  B=%d: %.0f img/s, %.2f ms per round" % ((B,) + run(one)))
del one
torch.cuda.empty_cache()
many = [mk(B // K, 100 + i) for i in range(K)]
print("%d models B=%d each, %d streams: %.0f img/s, %.2f ms per round" % ((K, B // K, K) + run(many)))
