#!/bin/bash
# usage (GPU box, repo root): tools/pmc_round.sh <tag>  -- per-kernel VALU wave instructions per step (C2 both workloads, C4) -> profiles/valu_insts.json
# (keys as bench.py's _profile_key: c2truth_b98304 = the default workload, SURVEY 8(d)'s; c2_b98304 = the ancestor stream), and the multiply micro-benchmark
TAG=$1
mkdir -p gpurun_out/prof
for V in "c2truth --workload truth" "c2 --workload ancestor" "c4 --config c4 --workload ancestor" "c4truth --config c4 --workload truth"; do
  set -- $V; K=$1; shift
  tools/pmc_all.sh "$@" > gpurun_out/prof/${TAG}_${K}_insts.txt 2>&1
  cp gpurun_out/prof/pmc_all_summary.csv gpurun_out/prof/${TAG}_${K}_insts_per_kernel.csv
  python3 tools/valu_insts.py gpurun_out/prof/${TAG}_${K}_insts_per_kernel.csv ${K}_b98304 ${TAG}_${K}
done
cp profiles/valu_insts.json gpurun_out/prof/${TAG}_valu_insts.json
hipcc --offload-arch=gfx950 -O3 tools/ubench/mul_rates.hip -o /tmp/mul_rates 2>/dev/null && /tmp/mul_rates > gpurun_out/prof/${TAG}_mul_rates.txt; cat gpurun_out/prof/${TAG}_mul_rates.txt
