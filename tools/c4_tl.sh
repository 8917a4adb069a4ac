#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 600 python3 bench.py --config c4 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py c4 | cut -c1-330
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $P/kt_tl -o kt -- python3 bench.py --config c4 --cpu-seconds 0 --no-extra-legs --no-check --steps 6 --warmup 2 > /dev/null 2>&1
F=$(find $P/kt_tl -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $F > $P/timeline_c4.txt; tail -70 $P/timeline_c4.txt | cut -c1-100
rm -rf $P/kt_tl
