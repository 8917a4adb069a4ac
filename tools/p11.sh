#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --steps 20 --no-extra-legs 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-110; }
run pad11k SKX_SKETCH_ROOM=2
run cap5_pad4k SKX_SKETCH_LDS_PAD_CAPPED=4096
run cap5_pad8k SKX_SKETCH_LDS_PAD_CAPPED=8192
run cap5_pad11k SKX_SKETCH_LDS_PAD_CAPPED=11264
run pad13k SKX_SKETCH_ROOM=2 SKX_SKETCH_LDS_PAD=13312
run pad16k SKX_SKETCH_ROOM=2 SKX_SKETCH_LDS_PAD=16384
run pad11k SKX_SKETCH_ROOM=2
