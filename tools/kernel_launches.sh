#!/bin/bash
# usage (GPU box): tools/kernel_launches.sh <tag> <kernel name substring> [bench args...]  -- every launch of the matching kernels of a
# bench.py run in start order: start offset (ms), duration (us), grid size, name -- to see WHICH launches of a kernel are the long ones
TAG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $P/kl_$TAG -o kt -- python3 bench.py --reps 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > $P/${TAG}_bench.json 2> $P/${TAG}_bench.err
python3 - "$PAT" $(find $P/kl_$TAG -name "*kernel_trace.csv" | head -1) > $P/${TAG}_launches.txt <<'PY'
import csv, sys
pats = sys.argv[1].split(",")
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "skx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("skx::", "")
    if any(p in n for p in pats):
        print("%10.3f ms %9.1f us  grid %8s  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                    r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Queue_Id", "?"), n[:60]))
PY
rm -rf $P/kl_$TAG
cat $P/${TAG}_launches.txt
