#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/kt_tl -o kt -- python3 bench.py --reps 1 --cpu-seconds 0 --no-extra-legs --no-check --steps 10 --warmup 3 > /dev/null 2>&1
F=$(find $P/kt_tl -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $F > $P/timeline_p3.txt; tail -120 $P/timeline_p3.txt | cut -c1-150
rm -rf $P/kt_tl
