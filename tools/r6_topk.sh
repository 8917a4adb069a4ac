#!/bin/bash
# usage (GPU box, repo root): tools/r6_topk.sh <tag>  -- the top-k tests, then bench.py at --top 16 / 5 / 2 on both workloads with the first and
# the last timed batch compared row by row with the oracle (20 batches from a fresh table, steady state, a lone batch)
TAG=$1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_species.py tests/test_gpu_patterns.py tests/test_gpu_shared_ancestor.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log
for T in 16 5 2; do for W in truth ancestor; do
  timeout 900 python3 bench.py --top $T --workload $W --no-truth-leg --no-end-to-end --no-large-batch --cpu-seconds 0 --steps 20 > gpurun_out/${TAG}_top${T}_$W.json 2> gpurun_out/${TAG}_top${T}_$W.err || tail -5 gpurun_out/${TAG}_top${T}_$W.err
  python3 tools/bench_line.py top$T-$W < gpurun_out/${TAG}_top${T}_$W.json | cut -c1-420
done; done | tee gpurun_out/${TAG}_topk.txt
