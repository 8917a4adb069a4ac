#!/bin/bash
# usage (GPU box): tools/scan_pmc_ab.sh  -- SQ/TCC counters of the scan kernel, baseline probe vs SKX_SCAN_FAST=1 (first two passes only)
sed -i 's/^ "TCC_HIT_sum.*$//; s/^ "TCC_EA0_WRREQ_STALL.*$//; s/^ "TCP_PENDING.*$//' tools/pmc_scan.sh
tools/pmc_scan.sh r02_scan_base SKX_SCAN_FAST=0 > /dev/null 2>&1
tools/pmc_scan.sh r02_scan_fast SKX_SCAN_FAST=1 > /dev/null 2>&1
for t in base fast; do echo "== $t"; sed 's/"[a-z ]*skx::\([a-z_0-9]*\)[^"]*"/\1/' gpurun_out/prof/r02_scan_${t}_pmc.csv | awk -F, '{printf "%-14s %-28s %4s %16s\n",$1,$2,$3,$4}'; done
for f in 0 1; do SKX_SCAN_FAST=$f timeout 300 python3 bench.py --cpu-seconds 0 --no-extra-legs 2>/dev/null | python3 tools/bench_line.py fast=$f; done
