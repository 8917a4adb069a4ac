#!/bin/bash
# usage (GPU box): tools/batch_sweep.sh <tag> -- reads/s of the C2 / C4 stream at several batch sizes (one reference scan is amortised over a batch)
TAG=$1
for CFG in c2 c4; do
  for B in ${BATCHES:-98304 196608 393216}; do
    for PF in 1 0; do
      S=20; [ $CFG = c4 ] && S=8
      env SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so SKX_PASS_READS=1048576 SKX_KMER_PREFILTER=$PF timeout 900 python3 bench.py --config $CFG --batch $B --steps $S --warmup 2 --reps 3 --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py "$CFG pf=$PF" | cut -c1-200
    done
  done
done | tee gpurun_out/${TAG}_batch_sweep.txt
