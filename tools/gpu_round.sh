#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round.sh <tag> [pytest args...]  -- the GPU test suite, then the default bench (+ c4)
TAG=$1; shift
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -x -q "$@" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -8 gpurun_out/${TAG}_pytest.log
timeout 600 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python3 tools/bench_line.py c2 < gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
timeout 900 python3 bench.py --config c4 --steps 8 --warmup 2 > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err; echo "bench c4 rc=$?"
python3 tools/bench_line.py c4 < gpurun_out/${TAG}_bench_c4.json; tail -3 gpurun_out/${TAG}_bench_c4.err
tools/prof_kt.sh ${TAG}_c4 --config c4 --steps 6 --warmup 2 | grep -E "rank_seg|seg_sum|transpose|scan_lean|sketch_|seg_prefix|chunk_|merge|word_bands" | grep -v rocprim
tools/prof_kt.sh ${TAG}_c2 --steps 12 --warmup 2 | grep -E "skx::" 
