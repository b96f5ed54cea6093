#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd sqlite database."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = cur.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, n, t, a, mn, mx in rows:
        short = name if len(name) <= 90 else name[:87] + "..."
        lines.append("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short, n, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    lines.append("TOTAL kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
