#!/bin/bash
# usage (GPU box, repo root): tools/r6_sparse_ab.sh <tag>  -- rare_to_mq's sparse writes on / off (experiments build): a lone truth-strain batch,
# 20 batches from a fresh table, the C4 truth-strain stream
TAG=$1
X="SKX_LIB_PATH=$PWD/sketchy_amd/libsketchy_hip_exp.so"
for V in 1 0 1 0; do
  echo "== SKX_RARE_SPARSE_WRITES=$V: lone batch (truth)"; env $X SKX_RARE_SPARSE_WRITES=$V timeout 300 python3 tools/diag_cold.py 8 truth 2>&1 | head -2 | cut -c1-200
done
tools/ab.sh -w truth "sparse1 SKX_RARE_SPARSE_WRITES=1" "sparse0 SKX_RARE_SPARSE_WRITES=0" "sparse1 SKX_RARE_SPARSE_WRITES=1" "sparse0 SKX_RARE_SPARSE_WRITES=0"
tools/ab.sh -c c4 -s 8 -w truth "c4-sparse1 SKX_RARE_SPARSE_WRITES=1" "c4-sparse0 SKX_RARE_SPARSE_WRITES=0"
