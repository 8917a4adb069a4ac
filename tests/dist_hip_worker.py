"""One rank of tests/test_gpu_configs.py::test_two_ranks_score_their_shards_on_the_hip_path (started by
torch.distributed.run, both ranks on device 0 of a 1-GPU box): shard the reads, score the shard on the HIP path,
reduce the tables, then redo the shard seeded with the totals of the earlier shards."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402,F401  (first: its HIP runtime is the one the process keeps)
from helpers import workload  # noqa: E402
from sketchy_amd import api, shard  # noqa: E402

out_dir = sys.argv[1]
rank, _, world = shard.env_rank()
shard.init_process_group("gloo")
ref, bases, offsets = workload(700, 300, 900, read_len=700, rng_seed=123)
lo, hi = shard.shard_range(len(offsets) - 1, rank, world)
R = api.ReferenceSketch(ref["ref"], ref["col_len"], device=0)
S = api.SumOfSharedHashes(R, top=2, max_batch_reads=hi - lo, max_batch_bases=int(offsets[hi] - offsets[lo]))
S.push(bases, offsets[lo:hi + 1])
mine = S.table()
prefix = shard.exclusive_prefix_tables(mine)
red = shard.TableReducer(0)
red.allreduce(S)
reduced = S.table()
S.reset()
S.table_add(prefix)
again = S.push(bases, offsets[lo:hi + 1])
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), reduced=reduced, idx=again["topk_idx"], sums=again["topk_sum"], how=red.how)
shard.barrier()
red.close()
import torch.distributed as dist  # noqa: E402
dist.destroy_process_group()
