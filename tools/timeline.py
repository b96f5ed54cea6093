"""Timeline of the LAST graph-replayed train step in a rocprofv3 kernel trace (csv): span, device-idle time, how long each
phase takes (forward | backward chain | tail after the last data-gradient), per-stream busy time.
usage: python tools/timeline.py <kernel_trace.csv> [out.txt]"""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
# steps are delimited by step_commit (one per step): the last interval between two commits that holds a whole step
commits = [i for i, r in enumerate(rows) if "step_commit" in r[2]]
step = None
for a_, b_ in zip(commits[:-1], commits[1:]):  # the shortest whole step = a graph replay (not the eager / profiled ones)
    if b_ - a_ > 300:
        cand = rows[a_ + 1:b_ + 1]
        span = max(r[1] for r in cand) - cand[0][0]
        if step is None or span < best:
            step, best = cand, span
assert step is not None, "no train step in the trace"
t0, t1 = step[0][0], max(r[1] for r in step)
out = []
P = out.append
P("kernels in the step: %d   span %.3f ms" % (len(step), (t1 - t0) / 1e6))
# device idle: union of busy intervals
busy = 0
cur_s, cur_e = step[0][0], step[0][1]
gaps = []
for s, e, *_ in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
P("device busy (union) %.3f ms, idle inside the step %.3f ms in %d gaps (mean %.2f us)" % (busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), (sum(g for g, _ in gaps) / max(1, len(gaps))) / 1e3))
# concurrency histogram: time with k kernels running
ev = []
for s, e, *_ in step:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
k = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[k] += t - last; last = t; k += d
P("time with k kernels resident: " + "  ".join("k=%d %.2f ms" % (kk, v / 1e6) for kk, v in sorted(hist.items())))
# phases
def first(sub):
    for r in step:
        if sub in r[2]:
            return r
    return None
def last(sub):
    z = None
    for r in step:
        if sub in r[2]:
            z = r
    return z
fin = first("elbo_finalize")
if fin:
    P("forward  (start .. elbo_finalize end): %.3f ms" % ((fin[1] - t0) / 1e6))
lastd = None
for r in step:
    n = r[2]
    if ("conv_" in n or "blk_kernel" in n) and "wgrad" not in n:
        lastd = r
ss = first("sumsq_partial")
if fin and lastd:
    P("backward chain (elbo_finalize end .. last fwd/dgrad conv end): %.3f ms" % ((lastd[1] - fin[1]) / 1e6))
    P("tail (last dgrad conv end .. step end): %.3f ms" % ((t1 - lastd[1]) / 1e6))
    tail = [r for r in step if r[1] > lastd[1]]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, *_ in tail:
        a = agg[n.split("(")[0][:70]]; a[0] += 1; a[1] += e - max(s, lastd[1])
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        P("    tail: %-70s x%-3d %.3f ms" % (n, c, d / 1e6))
P("kernels longer than 100 us (start .. end, ms from step start; queue):")
for s_, e_, n_, q_, st_ in step:
    if e_ - s_ > 100e3:
        P("    %7.3f .. %7.3f  q%s  %s" % ((s_ - t0) / 1e6, (e_ - t0) / 1e6, q_, n_.split("(")[0][:60]))
if fin:
    P("elbo_finalize ends at %.3f; last dgrad conv ends at %.3f" % ((fin[1] - t0) / 1e6, (lastd[1] - t0) / 1e6))
# time per kernel family over the step (sum of durations) and the sum of launch-to-launch gaps on the busiest stream
fam = collections.defaultdict(lambda: [0, 0])
for s, e, n, *_ in step:
    a = fam[n.split("(")[0].split("<")[0][:50]]; a[0] += 1; a[1] += e - s
for n, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
    P("  %-50s x%-4d sum %.3f ms  avg %.1f us" % (n, c, d / 1e6, d / c / 1e3))
# a window of consecutive kernels in the middle of the backward chain: start offset, duration, gap to the previous END on the
# same queue (negative = it started before the previous kernel of that queue had ended)
if fin:
    mid = [r for r in step if r[0] > fin[1]]
    mid = mid[len(mid) // 3: len(mid) // 3 + 40]
    last_end = {}
    P("window of 40 consecutive launches (backward, after a third of the chain): start us | dur us | gap to previous end on the same queue | queue | kernel (grid)")
    for s_, e_, n_, q_, st_ in mid:
        gap = (s_ - last_end[q_]) / 1e3 if q_ in last_end else float("nan")
        last_end[q_] = e_
        P("    %9.1f | %6.1f | %6.1f | q%s | %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, gap, q_, n_.split("(")[0][:48]))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
