"""Idle gaps inside the last graph-replayed train step (no kernel resident): which kernel ended before and which started after,
grouped.  usage: python tools/step_gaps.py <kernel_trace.csv>"""
import csv, sys, collections, re
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
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); n = re.sub(r"<.*$", "", n)
    return n.replace("cgen::", "")[:28]
t0 = step[0][0]
cur_e, cur_n, cur_q = step[0][1], step[0][2], step[0][3]
acc = collections.defaultdict(lambda: [0, 0.0])
big = []
for s, e, n, q in step[1:]:
    if s > cur_e:
        k = (short(cur_n), short(n), "same queue" if q == cur_q else "other queue")
        acc[k][0] += 1; acc[k][1] += (s - cur_e) / 1e3
        big.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e6, short(cur_n), short(n), q == cur_q))
    if e > cur_e:
        cur_e, cur_n, cur_q = e, n, q
print("gaps: %d, total %.1f us" % (sum(v[0] for v in acc.values()), sum(v[1] for v in acc.values())))
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-28s -> %-28s %-11s x%3d  %6.1f us  avg %5.2f" % (k[0], k[1], k[2], c, t, t / c))
print("largest:")
for g, at, a, b, sq in sorted(big, reverse=True)[:10]:
    print("  %6.1f us at %.3f ms  %s -> %s (%s)" % (g, at, a, b, "same queue" if sq else "other queue"))
