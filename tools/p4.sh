#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py "$L"; }
run p3
run p4 SKX_PIPELINE=4
run p4_pad0 SKX_PIPELINE=4 SKX_SKETCH_LDS_PAD=0
run p4_pad16k SKX_PIPELINE=4 SKX_SKETCH_LDS_PAD=16384
run p4_prio0 SKX_PIPELINE=4 SKX_SCAN_PRIO=0
run p4_spec0 SKX_PIPELINE=4 SKX_SPEC_INSERT=0
run p3_spec0 SKX_SPEC_INSERT=0
