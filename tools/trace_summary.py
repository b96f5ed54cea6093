#!/usr/bin/env python3
"""Per-kernel time per step from a rocprofv3 --kernel-trace CSV of bench.py (steps = number of wred_kernel launches).
usage: trace_summary.py <kernel_trace.csv> [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cgen::", "").replace("cgen::", "")
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
steps = agg.get("wred_kernel", [1])[0] or 1
tot = sum(v[1] for v in agg.values())
print("steps %d, sum of kernel durations %.2f ms/step" % (steps, tot / 1e3 / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-58s calls/step %6.1f  ms/step %7.3f  avg us %7.1f" % (k[:58], v[0] / steps, v[1] / 1e3 / steps, v[1] / v[0]))
cls = collections.defaultdict(float)
for k, v in agg.items():
    c = ("px" if "conv_px" in k else "ws" if "conv_ws" in k else "tile" if "conv_tile" in k else "generic" if "conv_kernel" in k else
         "wgrad_tile" if "wgrad_tile" in k else "wgrad_gen" if "wgrad_kernel" in k else "wred" if "wred" in k else "other")
    cls[c] += v[1] / 1e3 / steps
print({k: round(v, 2) for k, v in sorted(cls.items(), key=lambda kv: -kv[1])})
