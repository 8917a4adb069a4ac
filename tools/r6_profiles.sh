#!/bin/bash
# usage (GPU box, repo root): tools/r6_profiles.sh <tag>  -- the round's committed profiles: kernel stats + FETCH / WRITE_SIZE per workload and config, VALU instructions per kernel
TAG=$1
tools/prof_round.sh ${TAG}_c2truth --workload truth --steps 20 --warmup 5 > gpurun_out/${TAG}_prof_c2truth.log 2>&1
tools/prof_round.sh ${TAG}_c2 --workload ancestor --steps 20 --warmup 5 > gpurun_out/${TAG}_prof_c2.log 2>&1
tools/prof_round.sh ${TAG}_c4 --config c4 --workload ancestor --steps 8 --warmup 2 > gpurun_out/${TAG}_prof_c4.log 2>&1
tools/prof_round.sh ${TAG}_c4truth --config c4 --workload truth --steps 8 --warmup 2 > gpurun_out/${TAG}_prof_c4truth.log 2>&1
tools/pmc_round.sh ${TAG} > gpurun_out/${TAG}_pmc_round.log 2>&1
python3 tools/kernel_resources.py > gpurun_out/prof/${TAG}_kernel_resources.txt 2>&1
cp profiles/scan_traffic.json gpurun_out/prof/${TAG}_scan_traffic_all.json; cp profiles/valu_insts.json gpurun_out/prof/${TAG}_valu_insts_all.json
tail -3 gpurun_out/${TAG}_prof_*.log; tail -5 gpurun_out/${TAG}_pmc_round.log
