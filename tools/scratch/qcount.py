import csv, sys, re, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
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
c = collections.Counter(); t = collections.Counter()
for s, e, n, q in step:
    ph = "fwd" if e <= fin[1] else "bwd"
    c[(ph, q)] += 1; t[(ph, q)] += (e - s) / 1e3
print("span %.3f ms" % ((max(r[1] for r in step) - step[0][0]) / 1e6))
for k in sorted(c): print(k, c[k], "%.0f us" % t[k])
