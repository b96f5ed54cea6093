import csv, sys, re
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
t0 = step[0][0]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    return n.replace("cgen::", "")[:44]
idx = [i for i, r in enumerate(step) if "reparam_kl_bwd" in r[2]]
i0 = idx[18]
for j in range(i0 - 14, i0 + 16):
    s, e, n, q = step[j]
    print("   %9.1f .. %9.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, short(n)))
