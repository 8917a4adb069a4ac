#!/bin/bash
# usage (GPU box): tools/prof_round.sh <tag> [bench args...]   -- kernel-trace stats + FETCH/WRITE_SIZE (separate PMC passes) of bench.py
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_$TAG -o kt -- python3 bench.py --cpu-seconds 0 --no-extra-legs "$@" > $P/${TAG}_bench_under_rocprof.json 2> $P/${TAG}_bench.err
python3 profiles/filter_stats.py $(find $P/kt_$TAG -name "*kernel_stats.csv") > $P/${TAG}_kernel_stats.csv; rm -rf $P/kt_$TAG
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $P/pmc_${TAG}_$C -o pmc -- python3 bench.py --reps 1 --steps 4 --warmup 1 --cpu-seconds 0 --no-extra-legs --no-check "$@" > /dev/null 2> $P/${TAG}_pmc_$C.err
  python3 profiles/summarize_pmc.py $(find $P/pmc_${TAG}_$C -name "*counter_collection.csv") | grep -E "skx::|^kernel" > $P/${TAG}_pmc_$C.csv; rm -rf $P/pmc_${TAG}_$C
done
grep -E "skx::|^kernel" $P/${TAG}_kernel_stats.csv | cut -c1-120; cat $P/${TAG}_pmc_FETCH_SIZE.csv $P/${TAG}_pmc_WRITE_SIZE.csv | grep -E "scan_lean|scan_kernel|transpose"
# HBM bytes per scan launch -> profiles/scan_traffic.json (with the sha of the sources profiled); a copy travels back in gpurun_out/
# (key: <config>[truth]_b<batch> -- bench.py's _profile_key: the workload belongs to the key; truth is bench.py's default)
CFG=c2; B=98304; WL=truth; prev=""; for a in "$@"; do [ "$prev" = "--config" ] && CFG=$a; [ "$prev" = "--batch" ] && B=$a; [ "$prev" = "--workload" ] && WL=$a; prev=$a; done
[ "$WL" = "truth" ] && CFG=${CFG}truth
# (a dense dictionary -- C4's mixed stream -- is scanned by scan_kernel's split-array variant instead of the lean kernel)
KPAT=scan_lean_kernel; grep -q scan_lean_kernel $P/${TAG}_pmc_FETCH_SIZE.csv || KPAT="scan_kernel<"
python3 tools/scan_traffic.py $P/${TAG}_pmc_FETCH_SIZE.csv $P/${TAG}_pmc_WRITE_SIZE.csv ${CFG}_b$B $TAG "$KPAT" && cp profiles/scan_traffic.json $P/${TAG}_scan_traffic.json
