#!/bin/bash
# usage (on the GPU box, repo root): tools/prof_kernels.sh <tag> <bench args...>
# runs bench.py under rocprofv3 --kernel-trace --stats and leaves a filtered summary in gpurun_out/prof/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_$TAG -o kt -- python3 bench.py --cpu-seconds 0 "$@" > $P/${TAG}_bench.json 2> $P/${TAG}_bench.err
python3 profiles/filter_stats.py $(find $P/kt_$TAG -name "*kernel_stats.csv") > $P/${TAG}_kernel_stats.csv
rm -rf $P/kt_$TAG
grep -E "skx::|^kernel" $P/${TAG}_kernel_stats.csv | cut -c1-160
