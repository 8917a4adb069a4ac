#!/usr/bin/env python3
"""Per-kernel mean of a rocprofv3 --pmc counter from its *_counter_collection.csv.
usage: summarize_pmc.py <counter_collection.csv> > summary.csv"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        key = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"])
        acc[key][1] += 1
print("kernel,counter,dispatches,mean_value,total_value")
for (k, c), (tot, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"\"{k}\",{c},{n},{tot / n:.1f},{tot:.1f}")
