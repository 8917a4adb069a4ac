#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --steps 20 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-200; }
run split SKX_RANK_SPLIT=1
run nosplit SKX_RANK_SPLIT=0
run split SKX_RANK_SPLIT=1
run nosplit SKX_RANK_SPLIT=0
