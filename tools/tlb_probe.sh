#!/bin/bash
# usage (GPU box): tools/tlb_probe.sh <tag> -- the pure-streaming ablation of the scan (SKX_SCAN_ABLATE=2, serial pipeline: nothing
# beside it) at 3.2 / 6 / 12 GB of matrix, default allocation vs physically contiguous memory (SKX_MAT_ALLOC=1)
TAG=$1
X=SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so
for CFG in c2 c6g c4; do
  for A in 0 1; do
    for AB in 2 0; do
      env $X SKX_PIPELINE=1 SKX_SCAN_ABLATE=$AB SKX_MAT_ALLOC=$A timeout 600 python3 bench.py --config $CFG --steps 6 --warmup 2 --reps 1 --cpu-seconds 0 --no-extra-legs --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$CFG alloc=$A ablate=$AB  scan_ms=%.4f  GB/s=%.0f  frac=%.3f  bytes=%.2f GB' % (r['avg_launch_ms'], r['achieved'], r['frac'], r['algorithmic_bytes_per_launch']/1e9))"
    done
  done
done | tee gpurun_out/${TAG}_tlb_probe.txt
