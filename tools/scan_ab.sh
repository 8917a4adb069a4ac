#!/bin/bash
# usage (GPU box): tools/scan_ab.sh <tag>  -- parity subset, then the bench under scan-kernel variants
TAG=$1; shift
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_species.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
run() { L=$1; shift; env "$@" timeout 300 python3 bench.py --cpu-seconds 0 2>/dev/null | python3 tools/bench_line.py "$L"; }
run legacy SKX_SCAN_STREAM=0
run stream SKX_SCAN_STREAM=1
run stream SKX_SCAN_STREAM=1
run stream_run2 SKX_SCAN_RUN=2
run stream_run4 SKX_SCAN_RUN=4
run stream_run8 SKX_SCAN_RUN=8
run stream_run16 SKX_SCAN_RUN=16
run stream_bpc5 SKX_SCAN_BLOCKS_PER_CU=5
run stream_bpc7 SKX_SCAN_BLOCKS_PER_CU=7
run stream_abl1 SKX_SCAN_ABLATE=1
