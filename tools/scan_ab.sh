#!/bin/bash
# usage (GPU box): tools/scan_ab.sh <tag>  -- parity subset, then the bench under scan-kernel variants (serial pipeline: clean kernel times)
TAG=$1; shift
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_species.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -6 gpurun_out/${TAG}_pytest.log
run() { L=$1; shift; env SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so "$@" timeout 300 python3 bench.py --cpu-seconds 0 --no-extra-legs 2>/dev/null | python3 tools/bench_line.py "$L"; }
run serial_legacy SKX_PIPELINE=1 SKX_SCAN_LEAN=0
run serial_lean SKX_PIPELINE=1
run serial_lean_nt SKX_PIPELINE=1 SKX_SCAN_NT=1
run serial_lean_abl1 SKX_PIPELINE=1 SKX_SCAN_ABLATE=1
run serial_lean_abl2 SKX_PIPELINE=1 SKX_SCAN_ABLATE=2
run serial_legacy SKX_PIPELINE=1 SKX_SCAN_LEAN=0
run serial_lean SKX_PIPELINE=1
run serial_lean_filt33 SKX_PIPELINE=1 SKX_FILTER_LG=33
run lean
run lean_filt33 SKX_FILTER_LG=33
timeout 600 python3 bench.py --config c4 --steps 8 --warmup 2 --cpu-seconds 0 2>/dev/null | tee gpurun_out/${TAG}_c4.json | python3 tools/bench_line.py c4
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_c4.json')); print(d.get('pass_stats'))"
