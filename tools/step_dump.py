"""Kernel sequence of the LAST graph-replayed train step inside a time window (ms from the step start): start | duration | queue | kernel.
usage: python tools/step_dump.py <kernel_trace.csv> <from_ms> <to_ms>"""
import csv, sys, re
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
commits = [i for i, r in enumerate(rows) if "step_commit" in r[2]]
step = None
for a_, b_ in zip(commits[:-1], commits[1:]):
    if b_ - a_ > 300 and (step is None or (rows[b_][1] - rows[a_ + 1][0]) <= (rows[step[1]][1] - rows[step[0]][0])):
        step = (a_ + 1, b_)
seg = rows[step[0]:step[1] + 1]
t0 = seg[0][0]
lo, hi = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
for s, e, n, q in seg:
    if lo <= s - t0 <= hi:
        n = re.sub(r"\(.*", "", n).replace("void cgen::", "").replace("cgen::", "")
        print("%9.1f | %6.1f | q%-3s | %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:70]))
