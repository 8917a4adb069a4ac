#!/bin/bash
# usage (GPU box): KPAT=<kernel regex> tools/pmc_kernel.sh <tag> [bench args]   -- SQ/TCC counter means per launch
TAG=$1; shift
KPAT=${KPAT:-scan_kernel}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
 "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
)
i=0
: > $P/${TAG}_pmc.csv
for C in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $C --output-format csv -d $P/pmck_${TAG}_$i -o pmc -- python3 bench.py --steps 3 --warmup 1 --reps 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > /dev/null 2> $P/${TAG}_pmck_$i.err
  F=$(find $P/pmck_${TAG}_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$F" ]; then python3 profiles/summarize_pmc.py $F | grep -E "$KPAT" >> $P/${TAG}_pmc.csv; fi
  rm -rf $P/pmck_${TAG}_$i
done
sed 's/"[a-z ]*skx::\([a-z_0-9]*\)[^"]*"/\1/' $P/${TAG}_pmc.csv | awk -F, '{printf "%-22s %-34s %4s %16s\n",$1,$2,$3,$4}'
