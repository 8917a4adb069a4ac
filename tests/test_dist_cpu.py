"""Multi-GPU host logic on CPU: world_size-2 gloo processes.

The path shards over READS (reference replicated); the only exchange is the sum all-reduce of the
u64 running table.  Here each rank's shard is scored by the CPU oracle and the tables are reduced
through sketchy_amd.shard exactly as bench.py does it (RCCL is unavailable without GPUs, so the
TableReducer must fall back to its gloo host path -- also what bench.py does if RCCL fails)."""
import os
import socket

import numpy as np
import pytest

from helpers import workload
from oracle import oracle as orc


class FakeStream:
    """Stand-in for api.SumOfSharedHashes holding a host table (no GPU in this test)."""

    def __init__(self, table):
        self._t = table.copy()

    def table(self):
        return self._t.copy()

    def table_add(self, add):
        self._t = self._t + add

    def allreduce(self, comm):  # pragma: no cover - RCCL path, needs GPUs
        raise AssertionError("RCCL path must not be taken on CPU")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sketchy_amd import shard
    dist = shard.init_process_group("gloo")
    assert shard.env_rank() == (rank, rank, world)
    ref, bases, offsets = workload(40, 128, 61, read_len=500, genome_len=40000, rng_seed=77)
    n = len(offsets) - 1
    lo, hi = shard.shard_range(n, rank, world)
    mine = orc.stream(16, 0, 128, ref["ref"], ref["col_len"], bases, offsets[lo:hi + 1], top_k=2)
    # (1) final table: sum all-reduce
    total = shard.allreduce_table_host(mine["cum"])
    # (2) exactness extension: earlier shards' totals as this shard's starting table
    prefix = shard.exclusive_prefix_tables(mine["cum"])
    again = orc.stream(16, 0, 128, ref["ref"], ref["col_len"], bases, offsets[lo:hi + 1], top_k=2, cum=prefix)
    # (3) the reducer bench.py uses: falls back to gloo on CPU, identical on every rank
    red = shard.TableReducer(device=rank)
    fs = FakeStream(mine["cum"])
    red.allreduce(fs)
    shard.barrier()
    assert shard.max_over_ranks(float(rank)) == float(world - 1)
    assert shard.sum_over_ranks_int(1) == world
    np.savez(os.path.join(tmpdir, f"rank{rank}.npz"), total=total, idx=again["topk_idx"], sums=again["topk_sum"],
             lo=lo, hi=hi, reduced=fs.table(), how=red.how)
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from sketchy_amd import shard
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            cuts = [shard.shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_gloo_allreduce_matches_single_stream(tmp_path, world):
    """World sizes 2 and 8 (the driver's node): every rank scores its contiguous shard with the oracle; the summed tables and --
    with the exclusive-prefix extension -- every per-read row equal the single stream's (src/sketchy.rs:326, :341: u64 sums)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ref, bases, offsets = workload(40, 128, 61, read_len=500, genome_len=40000, rng_seed=77)
    single = orc.stream(16, 0, 128, ref["ref"], ref["col_len"], bases, offsets, top_k=2)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for p in parts:
        np.testing.assert_array_equal(p["total"], single["cum"])       # integer sums: exact, order-free
        np.testing.assert_array_equal(p["reduced"], single["cum"])
        assert str(p["how"]) == "gloo-host"
    np.testing.assert_array_equal(np.concatenate([p["idx"] for p in parts]), single["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([p["sums"] for p in parts]), single["topk_sum"])


def test_deadline_ends_a_rank_that_waits_for_ever(tmp_path):
    """shard.deadline: a block that outlives its deadline ends the PROCESS with exit code 86 and says which rank was stuck in
    what (the first N-rank bring-up must fail loudly instead of hanging); a block that finishes in time is left alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from sketchy_amd import shard\n"
            "with shard.deadline(30, 'a quick block'):\n    pass\n"
            "print('first block done', flush=True)\n"
            "with shard.deadline(1, 'the rendezvous nobody joins'):\n    time.sleep(30)\n"
            "print('never printed')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60,
                         env=dict(os.environ, RANK="3", LOCAL_RANK="3", WORLD_SIZE="8"))
    assert out.returncode == 86, (out.returncode, out.stderr[-500:])
    assert "first block done" in out.stdout and "never printed" not in out.stdout
    assert "rank 3 of 8" in out.stderr and "the rendezvous nobody joins" in out.stderr
