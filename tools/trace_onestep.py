import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wred_kernel" in r["Kernel_Name"]]
# take the interval between the last two wred launches (a graphed train step)
k = int(sys.argv[3]) if len(sys.argv) > 3 else -3
a, b = idx[k-1] + 1, idx[k] + 1
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
print("kernels %d, wall %.3f ms" % (len(seg), (t1 - t0) / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cgen::", "").replace("cgen::", "")
    agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("sum of durations %.3f ms" % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("%-58s calls %5d  ms %7.3f  avg us %7.1f" % (k[:58], v[0], v[1] / 1e3, v[1] / v[0]))
# streams
st = collections.defaultdict(float)
for r in seg: st[r.get("Stream_Id", r.get("Queue_Id"))] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(dict(st))
