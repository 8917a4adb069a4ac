#!/bin/bash
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 --steps 20 2>/dev/null | python3 tools/bench_line.py "$L" | cut -c1-200; }
run base
run rankhi SKX_PRIO_RANK=1
run base
run rankhi SKX_PRIO_RANK=1
run noprio SKX_PRIO=0
