#!/bin/bash
# usage (GPU box): tools/ab_base.sh [bench args] -- the same bench.py invocation from the baseline worktree (variants/base) and from this
# tree, alternating, on one box
for i in 1 2; do
  (cd variants/base && timeout 600 python3 bench.py --cpu-seconds 0 "$@" 2>/dev/null | python3 tools/bench_line.py "base " | cut -c1-170)
  timeout 600 python3 bench.py --cpu-seconds 0 "$@" 2>/dev/null | python3 tools/bench_line.py "head " | cut -c1-170
done
