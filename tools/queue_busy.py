#!/usr/bin/env python3
"""usage: queue_busy.py kernel_trace.csv [window_ms] [min_us] -- the last `window_ms` of a rocprofv3 --kernel-trace csv: busy time per HIP
queue, and the kernels of at least `min_us` in start order (offset, duration, queue, name)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 30e6
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
ks = [k for k in ks if "skx::" in k[2] or "rocclr" in k[2]]
t1 = ks[-1][1]
t0 = t1 - win
ks = [k for k in ks if k[0] >= t0]
busy = {}
for s, e, n, q in ks:
    busy[q] = busy.get(q, 0) + (e - s)
print("window %.1f ms; busy per queue (ms):" % (win / 1e6), {q: round(v / 1e6, 2) for q, v in sorted(busy.items())})
for s, e, n, q in ks:
    if (e - s) / 1e3 >= min_us:
        short = n.split("(")[0].replace("void ", "").replace("skx::", "")[:50]
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  q{q}  {short}")
