#!/bin/bash
# usage (GPU box, repo root): tools/ab_lib.sh [-c CONFIG] [-s STEPS] [-r ROUNDS] [-x "extra bench args"] label=path/to/lib.so ...
# the same bench.py invocation once per library build (SKX_LIB_PATH), alternating, ROUNDS times, on one box: how a kernel change
# is compared with the build it replaces (variants/libskx_base.so = a copy of the previous libsketchy_hip.so)
CFG=c2; STEPS=20; ROUNDS=2; EXTRA=""
while getopts "c:s:r:x:" o; do case $o in c) CFG=$OPTARG;; s) STEPS=$OPTARG;; r) ROUNDS=$OPTARG;; x) EXTRA=$OPTARG;; esac; done; shift $((OPTIND - 1))
for i in $(seq $ROUNDS); do
  for V in "$@"; do
    L=${V%%=*}; P=${V#*=}
    SKX_LIB_PATH=$PWD/$P timeout 600 python3 bench.py --config $CFG --steps $STEPS --cpu-seconds 0 --no-large-batch --no-end-to-end $EXTRA 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-330
  done
done
