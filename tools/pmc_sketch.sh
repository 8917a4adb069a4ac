#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sketch.sh <tag> [path/to/lib.so [bench args]]  -- SQ counter means per launch of the main sketch kernel
# (rocprofv3 serialises kernels under --pmc: the kernel runs alone, at its stand-alone occupancy)
TAG=$1; LIB=${2:-sketchy_amd/libsketchy_hip.so}; shift; shift
KPAT=${KPAT:-sketch_wave_kernel}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export SKX_LIB_PATH=$PWD/$LIB
P=gpurun_out/prof; mkdir -p $P
[ -n "$ONLY_INSTS" ] && PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES") || PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_ANY GRBM_GUI_ACTIVE"
)
i=0
: > $P/${TAG}_pmc.csv
for C in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --output-format csv -d $P/pmcs_${TAG}_$i -o pmc -- python3 bench.py --steps 4 --warmup 1 --reps 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > /dev/null 2> $P/${TAG}_pmcs_$i.err
  F=$(find $P/pmcs_${TAG}_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$F" ]; then python3 profiles/summarize_pmc.py $F | grep -E "$KPAT" >> $P/${TAG}_pmc.csv; else tail -3 $P/${TAG}_pmcs_$i.err; fi
  rm -rf $P/pmcs_${TAG}_$i
done
sed 's/"[a-z ]*skx::\([a-z_0-9]*\)<\([^>]*\)>[^"]*"/\1<\2>/' $P/${TAG}_pmc.csv | awk -F, '{printf "%-40s %-28s %4s %16s\n",$1,$2,$3,$4}'
