"""Per-block timeline of the packed weight-gradient launches inside the real train step (csrc/wgrad3.hip, CGEN_WG3_BLOCKLOG): how long
the blocks of each problem class take next to each other, how many are resident over time, what the end of each launch looks like.
usage: python tools/wg3_blocklog.py [config [batch]]      (default ukbb192, batch 32; f16)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CAP = 1 << 18
buf = torch.zeros(1 + 4 * CAP, dtype=torch.int64, device="cuda")
os.environ["CGEN_WG3_BLOCKLOG"] = hex(buf.data_ptr())
os.environ["CGEN_WG3_BLOCKLOG_CAP"] = str(CAP)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if cfg in ("morphomnist", "cmnist") else 32)
    m, hp = bench.build_model(cfg, "f16", False)
    m = m.cuda()
    from causal_gen_amd.train import TrainStep
    ts = TrainStep(m, hp, ema=True, use_graph=True)
    x, pa = bench.synth_batch(cfg, hp, B, "cuda", seed=100)
    for _ in range(26):
        ts.step(x, pa)
    torch.cuda.synchronize()
    n0 = int(buf[0])
    ts.step(x, pa)
    torch.cuda.synchronize()
    n1 = int(buf[0])
    rec = buf[1 + 4 * n0: 1 + 4 * n1].view(-1, 4).cpu()
    print("blocks in the step: %d" % rec.shape[0])
    t = rec[:, 2:4].double() / 100.0  # us
    t -= t.min()
    # launches: split at the largest gap in start times
    order = t[:, 0].argsort()
    st = t[order, 0]
    gaps = st[1:] - st[:-1]
    cut = int(gaps.argmax()) + 1 if len(st) > 1 and gaps.max() > 200 else len(st)
    for name, idx in (("launch 1", order[:cut]), ("launch 2", order[cut:])):
        if len(idx) == 0:
            continue
        a, b = t[idx, 0], t[idx, 1]
        t0, t1 = float(a.min()), float(b.max())
        dur = b - a
        print("%s: %d blocks, %.0f .. %.0f us (%.0f us); block duration mean %.0f us, max %.0f us; sum of durations / 512 = %.0f us" % (
            name, len(idx), t0, t1, t1 - t0, float(dur.mean()), float(dur.max()), float(dur.sum()) / 512))
        # residency over time
        line = []
        for k in range(10):
            tt = t0 + (t1 - t0) * (k + 0.5) / 10
            line.append("%d" % int(((a <= tt) & (b > tt)).sum()))
        print("   resident blocks at 5 %, 15 % ... 95 % of the launch: " + " ".join(line))
        r0, r1 = rec[idx, 0], rec[idx, 1]
        by = {}
        for q0, q1, d_, s_ in zip(r0.tolist(), r1.tolist(), dur.tolist(), a.tolist()):
            key = ((q0 >> 16) & 0xffff, (q0 >> 8) & 0xff, (q1 >> 48) & 0xffff, (q1 >> 32) & 0xffff, "%dx%d" % ((q0 & 0xff) >> 3, q0 & 7))
            by.setdefault(key, []).append((d_, s_, q0 >> 32))
        print("   class (side, ks, ci, co, fragment block): problems, blocks, share of the block time, mean / max duration, first .. last start")
        tot = float(dur.sum())
        for key, v in sorted(by.items(), key=lambda kv: -sum(d for d, _, _ in kv[1]))[:22]:
            ds = [d for d, _, _ in v]
            print("   %3d^2 k%d %3d->%3d %s: %2d problems %4d blocks %5.1f %% | %5.0f / %5.0f us | %5.0f .. %5.0f" % (
                key[0], key[1], key[2], key[3], key[4], len(set(p_ for _, _, p_ in v)), len(v), 100 * sum(ds) / tot, sum(ds) / len(ds), max(ds),
                min(s_ for _, s_, _ in v) - t0, max(s_ for _, s_, _ in v) - t0))


if __name__ == "__main__":
    main()
