#!/bin/bash
run() { L=$1; shift; env SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so "$@" timeout 600 python3 bench.py --config c4 --steps 6 --warmup 2 --cpu-seconds 0 --no-check 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-150; }
run c4_base
run c4_ablate2 SKX_SCAN_ABLATE=2
run c4_ablate3 SKX_SCAN_ABLATE=3
run c4_ablate1 SKX_SCAN_ABLATE=1
