#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --steps 20 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-200; }
run pad11k
run pad16k SKX_SKETCH_LDS_PAD=16384
run pad21k SKX_SKETCH_LDS_PAD=21504
run pad8k SKX_SKETCH_LDS_PAD=8192
run pad11k
run pad16k SKX_SKETCH_LDS_PAD=16384
run pad21k SKX_SKETCH_LDS_PAD=21504
