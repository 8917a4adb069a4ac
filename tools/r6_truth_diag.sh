#!/bin/bash
# usage (GPU box): tools/r6_truth_diag.sh <tag>  -- truth-strain workload: stage timeline by HIP events (experiments build) and per-kernel VALU instructions
TAG=$1
P=gpurun_out/prof; mkdir -p $P
SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so SKX_SPAN_DUMP=1 timeout 300 python3 bench.py --workload truth --reps 1 --cpu-seconds 0 --no-extra-legs --no-check --profile-all --steps 20 --warmup 5 > $P/${TAG}_spans.json 2> $P/${TAG}_spans.err
python3 tools/span_timeline.py $P/${TAG}_spans.err 0.1 > $P/${TAG}_spans_truth.txt; cat $P/${TAG}_spans_truth.txt
tools/pmc_all.sh --workload truth > $P/${TAG}_c2truth_insts.txt 2>&1
cp $P/pmc_all_summary.csv $P/${TAG}_c2truth_insts_per_kernel.csv
python3 tools/valu_insts.py $P/${TAG}_c2truth_insts_per_kernel.csv c2truth_b98304 ${TAG}_c2truth
cp profiles/valu_insts.json $P/${TAG}_valu_insts.json
