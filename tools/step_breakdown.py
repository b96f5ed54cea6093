"""Where the last graph-replayed train step spends its chain time: kernels on the critical stream grouped by phase (forward /
backward), kernel template and grid size (the grid tells the resolution).  usage: python tools/step_breakdown.py <kernel_trace.csv>"""
import csv, sys, collections, re

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        g = (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], g))
rows.sort()
commits = [i for i, r in enumerate(rows) if "step_commit" in r[2]]
step = None
for a_, b_ in zip(commits[:-1], commits[1:]):
    if b_ - a_ > 300:
        cand = rows[a_ + 1:b_ + 1]
        span = max(r[1] for r in cand) - cand[0][0]
        if step is None or span < best:
            step, best = cand, span
fin = next(r for r in step if "elbo_finalize" in r[2])
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); n = n.replace("cgen::", "")
    return n[:60]
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g in step:
    if "wgrad" in n or "wred" in n:
        ph = "bg"
    else:
        ph = "fwd" if e <= fin[1] else "bwd"
    k = (ph, short(n), g[0] // max(1, g[1]))
    acc[k][0] += 1; acc[k][1] += (e - s) / 1e3
tot = collections.Counter()
for (ph, n, g), (c, t) in acc.items():
    tot[ph] += t
print("totals (us): " + "  ".join("%s %.0f" % kv for kv in tot.items()))
for (ph, n, g), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-4s %-62s wgs %5d  x%3d  %7.1f us  avg %6.1f" % (ph, n, g, c, t, t / c))
