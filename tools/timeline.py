#!/usr/bin/env python3
"""usage: timeline.py kernel_trace.csv -- prints the kernels of the last full step (between two sketch kernels) with
start offset, duration and queue, from a rocprofv3 --kernel-trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows if "skx::" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"]]
ks.sort()
starts = [i for i, k in enumerate(ks) if "sketch_wave_kernel<16, 256" in k[2]]
a, b = starts[-4], starts[-2]  # two steps
t0 = ks[a][0]
print(f"step length {(ks[b][0] - t0) / 1e3:.1f} us")
for s, e, n, q in ks[a:b + 1]:
    short = n.split("(")[0].replace("void ", "").replace("skx::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {short}")
