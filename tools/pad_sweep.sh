#!/bin/bash
# usage (GPU box): tools/pad_sweep.sh <tag> -- the sketch kernel's LDS pad (how many of its workgroups fit a CU beside the scan) x prefilter
TAG=$1
X=SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so
for PF in 0 1; do
  for PAD in 0 8192 11264 13312 16384 19456 22528; do
    env $X SKX_KMER_PREFILTER=$PF SKX_SKETCH_LDS_PAD=$PAD timeout 600 python3 bench.py --steps 20 --reps 3 --cpu-seconds 0 --no-extra-legs 2>/dev/null | python3 tools/bench_line.py "pf=$PF pad=$PAD" | cut -c1-120
  done
done | tee gpurun_out/${TAG}_pad_sweep.txt
