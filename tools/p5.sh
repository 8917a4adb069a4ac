#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py "$L"; }
run base
run nt2 SKX_SCAN_NT=2
run nt3 SKX_SCAN_NT=3
run base
run pad8k SKX_SKETCH_LDS_PAD=8192
run pad0 SKX_SKETCH_LDS_PAD=0
run pad16k SKX_SKETCH_LDS_PAD=16384
