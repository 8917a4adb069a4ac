#!/bin/bash
# usage (GPU box): tools/prof_cold.sh <tag>  -- kernel trace of lone first batches (tools/diag_cold.py)
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_$TAG -o kt -- python3 tools/diag_cold.py 8 > $P/${TAG}_cold.txt 2> $P/${TAG}_cold.err
python3 profiles/filter_stats.py $(find $P/kt_$TAG -name "*kernel_stats.csv") > $P/${TAG}_cold_kernel_stats.csv; rm -rf $P/kt_$TAG
cat $P/${TAG}_cold.txt; grep -E "skx::|^kernel" $P/${TAG}_cold_kernel_stats.csv | cut -c1-110
