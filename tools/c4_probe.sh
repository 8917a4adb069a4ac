#!/bin/bash
TAG=$1
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "submit or errors" 2>&1 | tail -3
timeout 600 python3 bench.py --config c4 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | tee gpurun_out/${TAG}_c4.json | python3 tools/bench_line.py c4
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_c4.json')); print(d.get('pass_stats')); print(d.get('value_host_fed'))"
timeout 300 python3 bench.py --cpu-seconds 0 2>/dev/null | tee gpurun_out/${TAG}_c2.json | python3 tools/bench_line.py c2
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_c2.json')); print(d.get('pass_stats')); print(d.get('value_host_fed'))"
tools/prof_kt.sh ${TAG}_c4 --config c4 --steps 6 --warmup 2 | grep -E "rank_seg|seg_sum|transpose|scan_lean|sketch_|seg_prefix|chunk_|merge"
