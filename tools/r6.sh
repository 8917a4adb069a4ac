#!/bin/bash
# usage (GPU box, repo root): tools/r6.sh <tag> [tests...]  -- the given GPU tests (default: all), then the two bench workloads without
# their side legs (reads/s from a fresh table, steady state, cold)
TAG=$1; shift
mkdir -p gpurun_out
T="$@"; [ -z "$T" ] && T=tests
timeout 2400 python3 -m pytest $T -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -15 gpurun_out/${TAG}_pytest.log
for W in truth ancestor; do
  timeout 900 python3 bench.py --workload $W --no-truth-leg --no-end-to-end --no-large-batch --cpu-seconds 0 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err; echo "bench $W rc=$?"
  python3 tools/bench_line.py c2 < gpurun_out/${TAG}_bench_$W.json; tail -3 gpurun_out/${TAG}_bench_$W.err
done
