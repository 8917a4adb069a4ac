#!/bin/bash
# usage (GPU box): tools/bisect_bench.sh [bench args] -- the same bench line from every worktree under variants/ and from this tree, twice round
for i in 1 2; do
  for W in variants/base variants/w_*; do
    (cd $W && timeout 600 python3 bench.py --cpu-seconds 0 --no-extra-legs "$@" 2>/dev/null | python3 tools/bench_line.py "$(basename $W)" | cut -c1-110)
  done
  timeout 600 python3 bench.py --cpu-seconds 0 --no-extra-legs "$@" 2>/dev/null | python3 tools/bench_line.py "head" | cut -c1-110
done
