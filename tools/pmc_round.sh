#!/bin/bash
# usage (GPU box, repo root): tools/pmc_round.sh <tag>  -- per-kernel VALU wave instructions per step (C2 and C4) -> profiles/valu_insts.json,
# and the multiply micro-benchmark
TAG=$1
mkdir -p gpurun_out/prof
tools/pmc_all.sh > gpurun_out/prof/${TAG}_c2_insts.txt 2>&1
cp gpurun_out/prof/pmc_all_summary.csv gpurun_out/prof/${TAG}_c2_insts_per_kernel.csv
python3 tools/valu_insts.py gpurun_out/prof/${TAG}_c2_insts_per_kernel.csv c2_b98304 ${TAG}_c2
tools/pmc_all.sh --config c4 > gpurun_out/prof/${TAG}_c4_insts.txt 2>&1
cp gpurun_out/prof/pmc_all_summary.csv gpurun_out/prof/${TAG}_c4_insts_per_kernel.csv
python3 tools/valu_insts.py gpurun_out/prof/${TAG}_c4_insts_per_kernel.csv c4_b98304 ${TAG}_c4
cp profiles/valu_insts.json gpurun_out/prof/${TAG}_valu_insts.json
hipcc --offload-arch=gfx950 -O3 tools/ubench/mul_rates.hip -o /tmp/mul_rates 2>/dev/null && /tmp/mul_rates > gpurun_out/prof/${TAG}_mul_rates.txt; cat gpurun_out/prof/${TAG}_mul_rates.txt
