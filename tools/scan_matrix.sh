#!/bin/bash
# experiment matrix for the scan kernel (on the GPU box): band height x flush mode x table slots x ablation
for RB in 64 128 256 512; do
 for CFG in "0 2048 0" "1 2048 0" "0 2048 1" "0 4096 0" "1 4096 0"; do
  set -- $CFG
  SKX_RB=$RB SKX_SCAN_FLUSH=$1 SKX_SCAN_SLOTS=$2 SKX_SCAN_ABLATE=$3 python bench.py --batch ${B:-4096} --steps 6 --warmup 1 --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py rb=$RB flush=$1 slots=$2 ablate=$3 | cut -c1-150
 done
done
