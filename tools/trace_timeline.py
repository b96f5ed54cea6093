#!/usr/bin/env python3
"""Where the weight-gradient launches sit relative to the main chain inside one graphed step of a rocprofv3 kernel trace.
usage: trace_timeline.py <kernel_trace.csv> [step index from the end, default -3]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wred_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -3
seg = rows[idx[k - 1] + 1: idx[k] + 1]
t0 = int(seg[0]["Start_Timestamp"])
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
isw = lambda r: "wgrad" in r["Kernel_Name"] or "wred" in r["Kernel_Name"]
chain = [q for q in seg if not isw(q)]
for r in seg:
    if not isw(r) or (E(r) - S(r)) < 30000: continue
    busy = sum(min(E(q), E(r)) - max(S(q), S(r)) for q in chain if S(q) < E(r) and E(q) > S(r)) / 1e3
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cgen::", "")
    print("%-40s start %9.1f dur %7.1f wgs %6d  chain busy during it %7.1f us  q=%s" % (n[:40], (S(r) - t0) / 1e3, (E(r) - S(r)) / 1e3, int(r["Grid_Size_X"]) // 256, busy, r.get("Queue_Id")))
gap = 0; last = E(chain[0])
for q in chain[1:]:
    if S(q) > last: gap += S(q) - last
    last = max(last, E(q))
print("step wall %.2f ms; chain kernels %d, sum %.2f ms, idle between chain kernels %.2f ms, chain span %.2f ms" % ((E(seg[-1]) - t0) / 1e6, len(chain), sum(E(q) - S(q) for q in chain) / 1e6, gap / 1e6, (last - S(chain[0])) / 1e6))
