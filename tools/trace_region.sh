#!/bin/bash
# usage (GPU box): tools/trace_region.sh <tag> [lib.so] [bench args]  -- kernel trace of ONE timed region (20 steps from a fresh table) -> tools/trace_summary.py
TAG=$1; LIB=${2:-sketchy_amd/libsketchy_hip.so}; shift; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export SKX_LIB_PATH=$PWD/$LIB
P=gpurun_out/prof; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P/kt_$TAG -o kt -- python3 bench.py --reps 1 --cpu-seconds 0 --no-extra-legs --no-check --no-profile --steps ${NSTEPS:-20} --warmup 3 "$@" > /dev/null 2> $P/${TAG}_trace.err
F=$(find $P/kt_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $F ${BIN_US:-50} ${NSTEPS:-20} $DUMP_FROM $DUMP_TO > $P/${TAG}_region.txt; cat $P/${TAG}_region.txt
rm -rf $P/kt_$TAG
