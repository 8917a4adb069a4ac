#!/bin/bash
# usage (GPU box): tools/prof_kt.sh <tag> [bench args...]   -- rocprofv3 kernel-trace stats of bench.py, filtered to this library
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_$TAG -o kt -- python3 bench.py --reps 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > $P/${TAG}_bench_under_rocprof.json 2> $P/${TAG}_bench.err
python3 profiles/filter_stats.py $(find $P/kt_$TAG -name "*kernel_stats.csv") > $P/${TAG}_kernel_stats.csv; rm -rf $P/kt_$TAG
cat $P/${TAG}_kernel_stats.csv | cut -c1-130
