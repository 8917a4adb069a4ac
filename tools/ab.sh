#!/bin/bash
# usage (GPU box, repo root): tools/ab.sh [-c CONFIG] [-s STEPS] [-w ancestor|truth] "label ENV=val ENV=val" "label2 ..." ...
# runs bench.py once per variant (same box, back to back) and prints one condensed line each -- how every A/B of
# DESIGN.md was measured.  Boxes differ by a few per cent: compare inside one call only, and repeat the baseline.
CFG=c2; STEPS=20; WL=ancestor
while getopts "c:s:w:" o; do case $o in c) CFG=$OPTARG;; s) STEPS=$OPTARG;; w) WL=$OPTARG;; esac; done; shift $((OPTIND - 1))
for V in "$@"; do
  set -- $V; L=$1; shift
  # (variants with knobs load the experiments build: the product library reads no environment variable)
  X=""; [ $# -gt 0 ] && X="SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so"
  env $X "$@" timeout 600 python3 bench.py --config $CFG --workload $WL --steps $STEPS --cpu-seconds 0 --no-end-to-end --no-truth-leg --oracle-steps none 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-220
done
