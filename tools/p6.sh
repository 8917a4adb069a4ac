#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --steps 12 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-200; }
run rb64
run rb32 SKX_RB=32
run rb48 SKX_RB=48
run rb96 SKX_RB=96
run rb128 SKX_RB=128
run rb64
