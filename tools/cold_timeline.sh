#!/bin/bash
# usage (GPU box): tools/cold_timeline.sh <tag> [ancestor|truth]  -- the kernels of ONE lone first batch (tools/diag_cold.py) with their start offsets
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/kt_$TAG -o kt -- python3 tools/diag_cold.py 8 ${2:-ancestor} > $P/${TAG}_cold.txt 2> $P/${TAG}_cold.err
python3 - $(find $P/kt_$TAG -name "*kernel_trace.csv" | head -1) > $P/${TAG}_cold_timeline.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "skx::" in r["Kernel_Name"]]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("skx::", "")[:44], r.get("Queue_Id", "?")) for r in rows)
main = [i for i, k in enumerate(ks) if k[2].startswith("sketch_wave_kernel<16, 128")]
first = main[-5]  # (the last four pushes run with per-stage HIP events; the one before them is undisturbed)
t0 = ks[first - 1][0] if first > 0 and ks[first - 1][2].startswith("batch_check") else ks[first][0]
end = ks[main[-4]][0] - 1
prev_end = t0
for a, e, n, q in ks:
    if a < t0 or a > end: continue
    print("%9.1f %8.1f  gap %6.1f  q%s  %s" % ((a - t0) / 1e3, (e - a) / 1e3, (a - prev_end) / 1e3, q, n))
    prev_end = max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
PY
rm -rf $P/kt_$TAG
cat $P/${TAG}_cold.txt; cat $P/${TAG}_cold_timeline.txt
