#!/usr/bin/env python3
"""rocprofv3 <tag>_kernel_stats.csv -> the compact text table kept under profiles/.  usage: stats_to_txt.py <csv> <header line> > out.txt"""
import csv, sys
print("# " + sys.argv[2])
print("%-103s %5s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in csv.DictReader(open(sys.argv[1])):
    print("%-103s %5d %12.1f %10.2f %7.2f" % (r["Name"][:103], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
