#!/bin/bash
# per-kernel instruction counts of the bench's timed region (20 steps from a fresh table, no warm-up: the first batch's
# heavy ranking counts once in 20 as it does in the timed region; SQ_INSTS_*: wave instructions), summed per kernel name
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $P/pmc_all -o pmc -- python3 bench.py --steps 20 --warmup 0 --reps 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > /dev/null 2> $P/pmc_all.err
python3 profiles/summarize_pmc.py $(find $P/pmc_all -name "*counter_collection.csv") > $P/pmc_all_summary.csv
grep -E "skx::" $P/pmc_all_summary.csv | grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS" | sed 's/"[a-z ]*skx::\([a-z_0-9]*\)[^"]*"/\1/' | awk -F, '{printf "%-28s %-16s %4s %14s %16s\n",$1,$2,$3,$4,$5}' | sort -k2,2 -k5,5nr | head -70
rm -rf $P/pmc_all
