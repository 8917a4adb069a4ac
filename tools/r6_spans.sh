#!/bin/bash
# usage (GPU box): tools/r6_spans.sh <tag> [bench args]  -- stage timeline by HIP events (experiments build), 20 steps from a fresh table
TAG=$1; shift
P=gpurun_out/prof; mkdir -p $P
SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so SKX_SPAN_DUMP=1 timeout 300 python3 bench.py --reps 1 --cpu-seconds 0 --no-extra-legs --no-check --profile-all --steps 20 --warmup 5 "$@" > $P/${TAG}_spans.json 2> $P/${TAG}_spans.err
python3 tools/span_timeline.py $P/${TAG}_spans.err 0.1 > $P/${TAG}_spans.txt; cat $P/${TAG}_spans.txt
