#!/bin/bash
# usage (GPU box, repo root): tools/r6_final.sh <tag>  -- the round's closing run: the driver's bench command on the fresh box, C4 both workloads, then the
# GPU test suite (a bench that FOLLOWS the suite on one box runs its end-to-end leg at half the rate: page cache full of the tests' files)
TAG=$1
mkdir -p gpurun_out
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err; echo "bench rc=$?"
python3 tools/bench_line.py c2 < gpurun_out/${TAG}_bench_driver.json | cut -c1-600; tail -2 gpurun_out/${TAG}_bench_driver.err
timeout 1500 python3 bench.py --config c4 --steps 8 --warmup 2 --no-end-to-end > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err; echo "bench c4 rc=$?"
python3 tools/bench_line.py c4 < gpurun_out/${TAG}_bench_c4.json | cut -c1-600
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
