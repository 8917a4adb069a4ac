"""BASELINE.json configs on the GPU box beyond the small parity cases: C1 (5 000 genomes x s=1000) against the
oracle and through size-independent properties at 100k reads; a plain `python bench.py --gpus 2` (self-launching,
both ranks on the one GPU of the box); RCCL through the C ABI on a one-rank communicator.

C2 at full batch size lives in test_gpu_fullsize.py (it shares that module's 3.2 GB reference)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import workload
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_np(table, k):
    order = np.lexsort((np.arange(len(table)), -table.astype(np.int64)))  # sum desc, index asc
    return order[:k].astype(np.uint32)


@pytest.fixture(scope="module")
def c1(gpu):
    from sketchy_amd import api
    n_reads = 98304 + 8192
    ref, bases, offsets = workload(5000, 1000, n_reads, rng_seed=101)
    R = api.ReferenceSketch(ref["ref"], ref["col_len"])
    yield dict(ref=ref, bases=bases, offsets=offsets, R=R)
    R.close()


def test_c1_8k_reads_vs_oracle(c1):
    """BASELINE configs[1] (N=5 000, s=1 000, k=16): 8 192 reads, every row and the table against the oracle
    (src/sketchy.rs:337-349), in one push and in uneven cuts."""
    from sketchy_amd import api
    ref, bases, offsets, R = c1["ref"], c1["bases"], c1["offsets"], c1["R"]
    n = 8192
    exp = orc.stream(16, 0, 1000, ref["ref"], ref["col_len"], bases, offsets[:n + 1], top_k=3)
    S = api.SumOfSharedHashes(R, top=3, max_batch_reads=n, max_batch_bases=int(offsets[n]))
    got = S.push(bases, offsets[:n + 1])
    np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"])
    np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    S1 = api.SumOfSharedHashes(R, top=1, max_batch_reads=3000, max_batch_bases=3000 * 1500)
    parts = [S1.push(bases, offsets[a:b + 1]) for a, b in ((0, 1), (1, 3000), (3000, 5001), (5001, 8000), (8000, 8192))]
    np.testing.assert_array_equal(np.concatenate([p["topk_idx"][:, 0] for p in parts]), exp["topk_idx"][:, 0])
    np.testing.assert_array_equal(np.concatenate([p["topk_sum"][:, 0] for p in parts]), exp["topk_sum"][:, 0])
    np.testing.assert_array_equal(S1.table(), exp["cum"])


def test_c1_100k_reads_properties(c1):
    """The whole ~100k-read C1 stream as ONE pass (B = 98 304, the geometry bench.py times) against the same reads in
    4 096-read pushes, continuing the table the 8 192 oracle-checked reads left; plus the properties any size offers."""
    from sketchy_amd import api
    bases, offsets, R = c1["bases"], c1["offsets"], c1["R"]
    n0, B = 8192, 98304
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=int(offsets[n0 + B] - offsets[n0]))
    S.push(bases, offsets[:n0 + 1])
    t0 = S.table()
    big = S.push(bases, offsets[n0:])
    t_big = S.table()
    assert S.reads == n0 + B
    C = api.SumOfSharedHashes(R, top=1, max_batch_reads=4096, max_batch_bases=4096 * 1500)
    C.table_add(t0)
    cuts = [C.push(bases, offsets[a:a + 4097]) for a in range(n0, n0 + B, 4096)]
    np.testing.assert_array_equal(np.concatenate([c["topk_idx"] for c in cuts]), big["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([c["topk_sum"] for c in cuts]), big["topk_sum"])
    np.testing.assert_array_equal(C.table(), t_big)
    sums = big["topk_sum"][:, 0].astype(np.int64)
    assert (np.diff(sums) >= 0).all() and sums[-1] == t_big.max()
    assert big["topk_idx"][-1, 0] == _rank_np(t_big, 1)[0]
    idx, sm = S.rank(10)
    np.testing.assert_array_equal(idx, _rank_np(t_big, 10))
    np.testing.assert_array_equal(sm, t_big[idx])
    # shards add up: table(second part alone) = table(all) - table(first part)
    S.reset()
    S.push(bases, offsets[n0:])
    np.testing.assert_array_equal(S.table(), t_big - t0)


def test_rccl_one_rank_communicator_through_the_c_abi(gpu):
    """skx_comm_* / skx_stream_allreduce on a 1-rank RCCL communicator: ncclCommInitRank + ncclAllReduce(u64, sum, in
    place) run for real (a second rank needs a second GPU); the sum over one rank is the table itself."""
    from sketchy_amd import api
    ref, bases, offsets = workload(300, 256, 200, read_len=600, rng_seed=111)
    R = api.ReferenceSketch(ref["ref"], ref["col_len"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=200, max_batch_bases=len(bases))
    S.push(bases, offsets)
    before = S.table()
    assert before.any()
    comm = api.Comm(0, 0, 1, api.Comm.unique_id())
    assert comm.n_ranks == 1
    S.allreduce(comm)
    S.allreduce(comm)
    np.testing.assert_array_equal(S.table(), before)
    # a batch enqueued but not yet flushed belongs to the table the collective sums (regression: the all-reduce used to run
    # before the batch's passes were queued, so the reduced table missed it and the batch was added on top afterwards)
    S2 = api.SumOfSharedHashes(R, top=1, max_batch_reads=200, max_batch_bases=len(bases))
    d_b, d_o = api.DeviceBuffer.from_numpy(bases), api.DeviceBuffer.from_numpy(offsets)
    S2.enqueue_device(d_b.ptr, d_o.ptr, 200, int(offsets[-1]), None, None)
    S2.allreduce(comm)
    np.testing.assert_array_equal(S2.table(), before)
    d_b.free(); d_o.free()
    comm.close()


def _run(cmd, timeout=600, **env):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _bench_two_ranks(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--share-gpu", "--config", "c1", "--batch", "8192",
                          "--steps", "3", "--warmup", "1", "--reps", "2", "--cpu-seconds", "0", "--no-extra-legs", *extra],
                         capture_output=True, text=True, timeout=900, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines


def test_bench_self_launches_two_ranks(gpu):
    """`python bench.py --gpus 2` from a plain invocation (no torchrun, no WORLD_SIZE): the parent starts the ranks
    itself; both share the box's one GPU, which RCCL refuses -- the table goes through gloo on the host, and the line SAYS
    so: rccl_ranks = 0, allreduce.how = gloo-host (accepted only because --allow-host-allreduce is given)."""
    out, lines = _bench_two_ranks("--allow-host-allreduce")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and "parity_error" not in j
    assert j["parity"]["last_timed_step_vs_4096_read_cuts"] is True
    assert j["config"]["rccl_ranks"] == 0
    ar = j["config"]["allreduce"]
    assert ar["how"] == "gloo-host" and ar["ranks"] == 0 and ar["world"] == 2 and ar["ms"] > 0
    assert len(j["values_per_rank"]) == 2 and len(j["values_all"]) == 2 and j["reps"] == 2
    assert min(j["values_all"]) <= j["value"] <= max(j["values_all"])
    assert j["value"] > 0


def test_bench_fails_when_rccl_did_not_reduce_the_table(gpu):
    """The same run WITHOUT --allow-host-allreduce: a multi-GPU line whose table was not reduced by RCCL over all ranks must
    not come out green (on the driver's 8-GPU node a broken RCCL bring-up would otherwise look like a valid scaling point)."""
    out, lines = _bench_two_ranks()
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[0])
    assert "not reduced by RCCL" in j["parity_error"] and j["config"]["rccl_ranks"] == 0


def test_two_ranks_score_their_shards_on_the_hip_path(gpu, tmp_path):
    """World-size-2 run where every rank scores its shard of the reads through libsketchy_hip (not a stand-in), the
    tables are reduced through shard.TableReducer and the exactness extension (earlier shards' totals via
    skx_stream_table_add) reproduces the single-stream rows: compared with the oracle's single stream."""
    port = 29000 + os.getpid() % 2000
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(ROOT, "tests", "dist_hip_worker.py"), str(tmp_path)])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    ref, bases, offsets = workload(700, 300, 900, read_len=700, rng_seed=123)
    single = orc.stream(16, 0, 300, ref["ref"], ref["col_len"], bases, offsets, top_k=2)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for p in parts:
        np.testing.assert_array_equal(p["reduced"], single["cum"])
    np.testing.assert_array_equal(np.concatenate([p["idx"] for p in parts]), single["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([p["sums"] for p in parts]), single["topk_sum"])


def test_bench_eight_ranks_on_one_gpu(gpu):
    """`python bench.py --gpus 8` the way the driver launches it on its node, here with all eight ranks on the box's one GPU:
    eight references + eight streams must fit, every rank checks its own shard, rank 0 prints ONE line whose allreduce block
    says which transport summed the table (RCCL refuses ranks that share a device: gloo-host, accepted only by flag)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--share-gpu", "--allow-host-allreduce", "--config", "c1",
                          "--batch", "4096", "--steps", "3", "--warmup", "1", "--reps", "2", "--cpu-seconds", "0", "--no-extra-legs"],
                         capture_output=True, text=True, timeout=1500, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and "parity_error" not in j and len(j["values_per_rank"]) == 8
    assert j["config"]["allreduce"]["world"] == 8 and j["config"]["allreduce"]["how"] == "gloo-host"
    # every rank says on stderr what its reducer ended up with
    assert sum("table reducer ready" in ln for ln in out.stderr.splitlines()) == 8, out.stderr[-3000:]


@pytest.mark.parametrize("workload_kind", ["ancestor", "truth"])
def test_bench_eight_ranks_multi_species(gpu, workload_kind):
    """BASELINE configs[4]'s shape (five species resident, log-normal read lengths, a stream MIXED over all five, sharded over
    eight ranks) at reduced size, all ranks on the box's one GPU: the multi-species path has met eight ranks once.  Every rank
    checks its own shard's rows (4 096-read cuts); rank 0 also compares every row of its first and last step with the oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--share-gpu", "--allow-host-allreduce", "--config", "c4s",
                          "--workload", workload_kind, "--batch", "2048", "--steps", "3", "--warmup", "1", "--reps", "2", "--cpu-seconds", "0",
                          "--no-extra-legs"],
                         capture_output=True, text=True, timeout=1500, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and "parity_error" not in j and len(j["values_per_rank"]) == 8
    assert j["config"]["n_species"] == 5 and j["config"]["allreduce"]["world"] == 8
    assert j["parity"]["last_timed_step_vs_4096_read_cuts"] is True


def test_bench_multi_species_whole_steps_vs_oracle(gpu):
    """the same small five-species config on one rank: EVERY row of the first and of the last timed step and the final table
    against the oracle (orc_stream_fast, one run per species over the mixed stream)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for kind in ("ancestor", "truth"):
        out = subprocess.run([sys.executable, "bench.py", "--config", "c4s", "--workload", kind, "--batch", "8192", "--steps", "4", "--warmup", "1",
                              "--reps", "2", "--cpu-seconds", "0", "--no-extra-legs"],
                             capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        o = j["oracle_whole_steps"]
        assert o["timed_rows_match_oracle"] is True and o["final_table_matches_oracle"] is True and o["rows_compared"] == 2 * 8192 * 5


def test_comm_watchdog_ends_a_rank_whose_peer_never_arrives(gpu):
    """Option comm_timeout_ms: ncclCommInitRank for a 2-rank communicator with only rank 0 present blocks for ever; with the
    watchdog the PROCESS ends with SKX_COMM_TIMEOUT_EXIT (86) after the stated time and says which rank was stuck where."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from sketchy_amd import api\n"
            "api.set_option('comm_timeout_ms', 4000)\n"
            "api.Comm(0, 0, 2, api.Comm.unique_id())\n"
            "print('never printed')\n" % ROOT)
    out = _run([sys.executable, "-c", code], timeout=240)
    assert out.returncode == 86, (out.returncode, out.stdout[-500:], out.stderr[-1500:])
    assert "never printed" not in out.stdout
    assert "rank 0 of 2" in out.stderr and "ncclCommInitRank" in out.stderr
