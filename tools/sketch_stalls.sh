#!/bin/bash
# usage (GPU box, repo root): tools/sketch_stalls.sh <tag>  -- where the main sketch kernel's issue slots go.
# (1) tries the instruction-level trace (rocprofv3 --att on the kernel alone); this image ships no trace decoder library, the attempt and its
#     message are recorded; (2) SQ counters of the kernel alone (rocprofv3 serialises kernels under --pmc): busy / wait / active cycles per
#     instruction class, LDS conflicts, instruction mix -> fractions of the kernel's wave-cycles (tools/sketch_stalls.py)
TAG=$1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
{ echo "== rocprofv3 --att attempt"; ls /opt/rocm/lib | grep -i -E "decoder|att" ; timeout 200 rocprofv3 --att --att-target-cu 1 --kernel-include-regex "sketch_wave_kernel" --output-format csv -d $P/att_$TAG -o att -- python3 bench.py --steps 2 --warmup 0 --reps 1 --cpu-seconds 0 --no-extra-legs --no-check 2>&1 | grep -v "^{" | tail -15; echo "rc=$?"; find $P/att_$TAG -type f 2>/dev/null | head -20; } > $P/${TAG}_att_attempt.txt 2>&1
rm -rf $P/att_$TAG
cat $P/${TAG}_att_attempt.txt | tail -25
tools/pmc_sketch.sh ${TAG}_sketch > /dev/null 2>&1
python3 tools/sketch_stalls.py $P/${TAG}_sketch_pmc.csv | tee $P/${TAG}_sketch_stalls.txt
