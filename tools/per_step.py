#!/usr/bin/env python3
"""usage: per_step.py kernel_trace.csv -- duration (us) of the main kernels per launch, in launch order (one column per kernel)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = ["sketch_wave_kernel<16, 256", "scan_lean_kernel", "transpose_bits", "seg_sum", "chunk_prefix", "seg_prefix", "rank_seg_top", "k_merge", "top1_merge", "chunk_leader_part", "dict_insert"]
cols = {n: [] for n in names}
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    for n in names:
        if n in r["Kernel_Name"]:
            cols[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(" ".join(f"{n.split('<')[0][:12]:>12s}" for n in names))
for i in range(max(len(v) for v in cols.values())):
    print(" ".join(f"{cols[n][i]:12.0f}" if i < len(cols[n]) else " " * 12 for n in names))
