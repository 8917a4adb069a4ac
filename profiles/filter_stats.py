#!/usr/bin/env python3
"""Keep only this library's kernels (skx::, rocPRIM, runtime fills/copies) from a rocprofv3
*_kernel_stats.csv and shorten the names.  usage: filter_stats.py in.csv > out.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_ms"])
for r in rows:
    n = r["Name"]
    if "skx::" in n or "rocprim" in n or "rocclr" in n:
        short = n.split("(")[0].replace("void ", "")[:90]
        w.writerow([short, r["Calls"], f"{float(r['AverageNs']) / 1e3:.1f}", f"{float(r['MinNs']) / 1e3:.1f}",
                    f"{float(r['MaxNs']) / 1e3:.1f}", f"{float(r['TotalDurationNs']) / 1e6:.3f}"])
