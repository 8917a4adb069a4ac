#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py "$L"; }
run default_prio1
run default_prio0 SKX_SCAN_PRIO=0
for PAD in 8192 11264 13312 16384 21504; do
run p3_prio1_pad$PAD SKX_PIPELINE=3 SKX_SKETCH_LDS_PAD=$PAD
run p3_prio0_pad$PAD SKX_PIPELINE=3 SKX_SKETCH_LDS_PAD=$PAD SKX_SCAN_PRIO=0
done
run default_prio1
