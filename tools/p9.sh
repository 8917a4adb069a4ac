#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --steps 20 --no-extra-legs 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-100; }
V=$GRAFT_REPO_ROOT/variants
for i in 1 2 3; do
run s2r2
run s0r3 SKX_LIB_PATH=$V/libskx_s0r3.so
run s2r3 SKX_LIB_PATH=$V/libskx_s2r3.so
done
