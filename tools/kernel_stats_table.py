#!/usr/bin/env python3
"""Print the kernel_stats.csv of a `rocprofv3 --kernel-trace --stats --output-format csv -d <dir>` run as a table:  tools/kernel_stats_table.py <dir> [title]"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
print(sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --kernel-trace --stats (durations in us)")
if not fs:
    print("no kernel_stats.csv")
    sys.exit(0)
print("%6s %10s %10s %10s %6s  kernel" % ("calls", "avg_us", "min_us", "max_us", "pct"))
for r in list(csv.DictReader(open(fs[0])))[:20]:
    print("%6s %10.1f %10.1f %10.1f %6.2f  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"]), r["Name"][:130]))
