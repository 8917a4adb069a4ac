#!/bin/bash
# usage (GPU box, repo root): tools/pmc_scan.sh <tag> [env assignments...] -- collects PMC passes for the bench and
# prints per-kernel means for kernels matching $KPAT (default scan_kernel)
set -e
TAG=$1; shift
KPAT=${KPAT:-scan_kernel}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
# (a pass with GRBM_GUI_ACTIVE + TA_* counters aborted rocprofv3 and hung the run: left out)
)
i=0
: > $P/${TAG}_pmc.csv
for C in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 120 env "$@" rocprofv3 --pmc $C --output-format csv -d $P/pmc_${TAG}_$i -o pmc -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extra-legs --no-check > /dev/null 2> $P/${TAG}_pmc_$i.err || true
  F=$(find $P/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then python3 profiles/summarize_pmc.py $F | grep -E "$KPAT" >> $P/${TAG}_pmc.csv || true; fi
  rm -rf $P/pmc_${TAG}_$i
done
cat $P/${TAG}_pmc.csv | cut -c1-200
