#!/usr/bin/env python3
"""usage: trace_summary.py kernel_trace.csv [bin_us]  -- the timed region of a bench run (from the last skx reset = the last
long gap before K sketch launches to the end) as: per-kernel totals, per-queue busy time, time with n queues busy, and a coarse
Gantt chart (one row per queue, one letter per bin: S sketch, s sketch follow-ups, D dictionary, C scan, T transpose, U seg/chunk
sums, P prefixes/leaders, R ranking, M merges, . idle)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
bin_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows if "skx::" in r["Kernel_Name"]]
ks.sort()
main = [i for i, k in enumerate(ks) if "sketch_wave_kernel<16, 128, true>" in k[2] or "sketch_wave_kernel<16, 256, true>" in k[2]]
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
first = main[-n_steps]
t0 = ks[first][0]
# (the batch_check kernel in front of the first main kernel belongs to the region too)
sel = [k for k in ks if k[0] >= t0 - 20000]
t0 = min(k[0] for k in sel)
t1 = max(k[1] for k in sel)
print(f"region: {n_steps} steps, {(t1 - t0) / 1e6:.3f} ms = {(t1 - t0) / 1e3 / n_steps:.1f} us per step")


def cls(n):
    n = n.replace("void ", "").replace("skx::", "")
    if n.startswith("sketch_wave_kernel<16, 128") or n.startswith("sketch_wave_kernel<16, 256"): return "S"
    if n.startswith(("sketch_", "batch_check", "count_scan", "publish", "dict_insert")): return "s"
    if n.startswith(("dict_", "pair_q", "window", "word_bands", "exceptions")): return "D"
    if n.startswith("scan_"): return "C"
    if n.startswith("transpose"): return "T"
    if n.startswith("seg_sum"): return "U"
    if n.startswith(("chunk_", "seg_prefix", "seg_lead")): return "P"
    if n.startswith("rank_"): return "R"
    if "merge" in n: return "M"
    return "o"


tot = defaultdict(lambda: [0, 0.0])
for s, e, n, q in sel:
    key = n.split("(")[0].replace("void ", "").replace("skx::", "")[:48]
    tot[key][0] += 1
    tot[key][1] += (e - s) / 1e3
print("kernel                                             launches   total_us   us/step")
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:50s} {c:8d} {us:10.1f} {us / n_steps:9.1f}")
queues = sorted({k[3] for k in sel})
busy = {q: 0.0 for q in queues}
events = []
for s, e, n, q in sel:
    busy[q] += (e - s) / 1e3
    events.append((s, 1)); events.append((e, -1))
print("queue busy (us, sum of kernel durations; kernels of one queue may overlap at their edges):", {q: round(v) for q, v in busy.items()})
events.sort()
level, last, hist = 0, t0, defaultdict(float)
for t, d in events:
    hist[level] += (t - last) / 1e3
    last = t
    level += d
print("time with n kernels in flight (us):", {k: round(v) for k, v in sorted(hist.items())})
nb = int((t1 - t0) / 1e3 / bin_us) + 1
for q in queues:
    line = ["."] * nb
    for s, e, n, qq in sel:
        if qq != q: continue
        c = cls(n)
        for b in range(int((s - t0) / 1e3 / bin_us), int((e - t0) / 1e3 / bin_us) + 1):
            if line[b] == "." or c in "SCR": line[b] = c
    print(f"q{q:>3s} " + "".join(line))
if len(sys.argv) > 5:  # dump every kernel of the region that starts inside [a_us, b_us)
    a_us, b_us = float(sys.argv[4]), float(sys.argv[5])
    for s, e, n, q in sel:
        if a_us <= (s - t0) / 1e3 < b_us:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n.split('(')[0].replace('void ', '').replace('skx::', '')[:60]}")
