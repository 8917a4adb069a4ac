"""Full-size (BASELINE config C2: 40 000 genomes x s=10 000, 3.2 GB of hashes) checks through
size-independent properties, plus a small oracle sample at full size.  The reference matrix is
generated in a SUBPROCESS with torch on the GPU (this process must not import torch after the HIP
library: two HIP runtimes in one process)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, S_, B = 40000, 10000, 4096
B_FULL = 98304  # the batch bench.py times: the whole ~100k-read C2 stream as one pass


@pytest.fixture(scope="module")
def c2(gpu):
    d = tempfile.mkdtemp(prefix="skx_c2_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sketchy_amd import synth\n"
        "ref = synth.make_reference(%d, %d, rng_seed=1, device='cuda' if torch.cuda.is_available() else 'numpy')\n"
        "bases, offsets = synth.make_reads(ref['genome'], %d, 1500, rng_seed=4242)\n"
        "np.save(%r + '/ref.npy', ref['ref']); np.save(%r + '/bases.npy', bases); np.save(%r + '/offsets.npy', offsets)\n"
    ) % (ROOT, N, S_, 2 * B + B_FULL, d, d, d)
    subprocess.check_call([sys.executable, "-c", code])
    ref = np.load(d + "/ref.npy", mmap_mode="r")
    out = dict(ref=np.ascontiguousarray(ref), bases=np.load(d + "/bases.npy"), offsets=np.load(d + "/offsets.npy"))
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
    from sketchy_amd import api
    out["R"] = api.ReferenceSketch(out["ref"])
    yield out
    out["R"].close()


def _rank_np(table, k):
    order = np.lexsort((np.arange(len(table)), -table.astype(np.int64)))  # sum desc, index asc
    return order[:k].astype(np.uint32)


def test_shard_invariance_determinism_and_rank(c2):
    from sketchy_amd import api
    R, bases, offsets = c2["R"], c2["bases"], c2["offsets"]
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=B * 1500)
    a = S.push(bases, offsets[:B + 1])
    ta = S.table()
    b = S.push(bases, offsets[B:2 * B + 1])
    tab = S.table()
    assert S.reads == 2 * B
    # running sums only grow, and the emitted top row is the table's leader at that read
    sums = np.concatenate([a["topk_sum"][:, 0], b["topk_sum"][:, 0]])
    assert (np.diff(sums.astype(np.int64)) >= 0).all()
    assert sums[B - 1] == ta.max() and a["topk_idx"][-1, 0] == _rank_np(ta, 1)[0]
    assert sums[-1] == tab.max() and b["topk_idx"][-1, 0] == _rank_np(tab, 1)[0]
    idx, sm = S.rank(25)
    np.testing.assert_array_equal(idx, _rank_np(tab, 25))
    np.testing.assert_array_equal(sm, tab[idx])
    # shards add up exactly (integer sums): table(second half alone) == table(all) - table(first half)
    S.reset()
    b2 = S.push(bases, offsets[B:2 * B + 1])
    np.testing.assert_array_equal(S.table(), tab - ta)
    # determinism: same input, same bytes
    S.reset()
    b3 = S.push(bases, offsets[B:2 * B + 1])
    np.testing.assert_array_equal(b2["topk_idx"], b3["topk_idx"])
    np.testing.assert_array_equal(b2["topk_sum"], b3["topk_sum"])
    # seeding the second shard with the first shard's totals reproduces the single-stream rows
    S.reset()
    S.table_add(ta)
    b4 = S.push(bases, offsets[B:2 * B + 1])
    np.testing.assert_array_equal(b4["topk_idx"], b["topk_idx"])
    np.testing.assert_array_equal(b4["topk_sum"], b["topk_sum"])
    # different batch cuts give the same stream (B=4096 vs 8 x 512)
    S2 = api.SumOfSharedHashes(R, top=1, max_batch_reads=512, max_batch_bases=512 * 1500)
    parts = [S2.push(bases, offsets[i:i + 513]) for i in range(0, B, 512)]
    np.testing.assert_array_equal(np.concatenate([p["topk_idx"] for p in parts]), a["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([p["topk_sum"] for p in parts]), a["topk_sum"])
    np.testing.assert_array_equal(S2.table(), ta)


def test_full_batch_one_pass_vs_4096_read_cuts_and_oracle_sample(c2):
    """B = 98 304 reads in ONE push (96 chunks of 1024 reads, one scan, the chunk-level pruning at that scale -- what
    bench.py times) gives the rows and the table of the same reads pushed 4 096 at a time, and its first rows are the
    oracle's (src/sketchy.rs:337-349) -- checked through the very same stream."""
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = c2["R"], c2["ref"], c2["bases"], c2["offsets"]
    n0 = 2 * B
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B_FULL, max_batch_bases=int(offsets[n0 + B_FULL] - offsets[n0]))
    big = S.push(bases, offsets[n0:n0 + B_FULL + 1])
    t_big = S.table()
    n = 6
    exp = orc.stream(16, 0, S_, ref, np.full(N, S_, np.uint32), bases, offsets[n0:n0 + n + 1], top_k=1)
    np.testing.assert_array_equal(big["topk_idx"][:n], exp["topk_idx"])
    np.testing.assert_array_equal(big["topk_sum"][:n], exp["topk_sum"])
    C = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=B * 1500)
    cuts = [C.push(bases, offsets[a:a + B + 1]) for a in range(n0, n0 + B_FULL, B)]
    np.testing.assert_array_equal(np.concatenate([c["topk_idx"] for c in cuts]), big["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([c["topk_sum"] for c in cuts]), big["topk_sum"])
    np.testing.assert_array_equal(C.table(), t_big)
    # a second full batch continues the table (steady state: pruned ranking against a clear leader)
    big2 = S.push(bases, offsets[n0:n0 + B_FULL + 1])
    cuts2 = [C.push(bases, offsets[a:a + B + 1]) for a in range(n0, n0 + B_FULL, B)]
    np.testing.assert_array_equal(np.concatenate([c["topk_idx"] for c in cuts2]), big2["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([c["topk_sum"] for c in cuts2]), big2["topk_sum"])
    np.testing.assert_array_equal(C.table(), S.table())


def test_top5_rows_are_consistent_with_top1(c2):
    from sketchy_amd import api
    R, bases, offsets = c2["R"], c2["bases"], c2["offsets"]
    S1 = api.SumOfSharedHashes(R, top=1, max_batch_reads=1024, max_batch_bases=1024 * 1500)
    S5 = api.SumOfSharedHashes(R, top=5, max_batch_reads=1024, max_batch_bases=1024 * 1500)
    a = S1.push(bases, offsets[:1025])
    b = S5.push(bases, offsets[:1025])
    np.testing.assert_array_equal(a["topk_idx"][:, 0], b["topk_idx"][:, 0])
    np.testing.assert_array_equal(a["topk_sum"][:, 0], b["topk_sum"][:, 0])
    assert (np.diff(b["topk_sum"].astype(np.int64), axis=1) <= 0).all()      # rows sorted by sum desc
    ties = b["topk_sum"][:, :-1] == b["topk_sum"][:, 1:]
    assert (b["topk_idx"][:, :-1][ties] < b["topk_idx"][:, 1:][ties]).all()  # ties in reference order
    np.testing.assert_array_equal(S1.table(), S5.table())


def test_self_intersection_is_sketch_size(c2):
    """docs/index.md:145-149: a sketch shares all of its s hashes with itself."""
    R, ref = c2["R"], c2["ref"]
    pick = np.array([0, 1, 255, 256, 20000, N - 1])
    common = R.common_hashes(ref[pick])
    assert (common[np.arange(len(pick)), pick] == S_).all()
    assert (common <= S_).all()


def test_oracle_sample_at_full_size(c2):
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = c2["R"], c2["ref"], c2["bases"], c2["offsets"]
    n = 6
    exp = orc.stream(16, 0, S_, ref, np.full(N, S_, np.uint32), bases, offsets[:n + 1], top_k=3, want_shared=True)
    S = api.SumOfSharedHashes(R, top=3, max_batch_reads=n, max_batch_bases=n * 1500)
    got = S.push(bases, offsets[:n + 1], want_shared=True)
    np.testing.assert_array_equal(got["shared"], exp["shared"])
    np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"])
    np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])


def test_c2_footprint_and_four_streams_on_one_reference(c2):
    """C2 at full size: the reference (3.2 GB of hashes + filter + k-mer table) and ONE full-batch stream take less than 8 GB
    of device memory together (rounds 1-2: ~75 GB -- rows and bit matrices sized for debug outputs and pair counts); four
    streams share the reference, each scores its own reads."""
    from sketchy_amd import api
    R, bases, offsets = c2["R"], c2["bases"], c2["offsets"]
    assert R.kmer_filter == (0, 0)   # (policy "kmer_prefilter": off by default)
    free0, total = api.device_mem(0)
    streams = []
    for i in range(4):
        streams.append(api.SumOfSharedHashes(R, top=1, max_batch_reads=B_FULL, max_batch_bases=B_FULL * 1500))
        if i == 0:
            free1, _ = api.device_mem(0)
    free4, _ = api.device_mem(0)
    one = free0 - free1
    ref_bytes = R.pass_bytes + 64 * (1 << 20)  # matrix + filter + k-mer table (upper bound)
    assert one + ref_bytes < 8 * (1 << 30), (one, ref_bytes)
    assert free0 - free4 < 4 * one + (1 << 30)
    tables = []
    for i, S in enumerate(streams):
        S.push(bases, offsets[i * 2048:(i + 1) * 2048 + 1])
        tables.append(S.table())
    S0 = streams[0]
    S0.reset()
    S0.push(bases, offsets[3 * 2048:4 * 2048 + 1])
    np.testing.assert_array_equal(S0.table(), tables[3])
    assert not np.array_equal(tables[0], tables[1])
    print("footprint: one stream %.2f GB, reference <= %.2f GB" % (one / 2**30, ref_bytes / 2**30))
    for S in streams:
        S.close()


@pytest.mark.parametrize("split,big", [("0", "0"), ("1", "0"), ("1", "1")])
def test_forced_scan_kernel_variants(gpu, split, big):
    """The split-array and big-table scan variants are picked automatically only for dense full-size passes;
    force each of them (env, read once per process) on small ragged inputs and compare with the oracle.
    (SKX_RARE_MAX=0: no rare-hash index, so every query hash of these small panels really goes through the scan.)"""
    from helpers import exp_env
    env = exp_env(SKX_SCAN_SPLIT=split, SKX_SCAN_BIG=big, SKX_RARE_MAX=0)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py")], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "variant ok" in out.stdout


@pytest.mark.parametrize("knobs", [{"SKX_PASS_READS": "64"}, {"SKX_PASS_READS": "100", "SKX_PIPELINE": "1"},
                                   {"SKX_PASS_READS": "37", "SKX_PIPELINE": "2"}, {"SKX_NO_FILTER": "1"},
                                   {"SKX_TOP1_WIDE": "1"}, {"SKX_RANK_LIVE": "0"}, {"SKX_RANK_LANES": "1"},
                                   {"SKX_RANK_LANES": "4", "SKX_PASS_READS": "100"}, {"SKX_SPEC_INSERT": "0"}, {"SKX_PIPELINE": "4"},
                                   {"SKX_TWO_LEVEL": "1"}, {"SKX_TWO_LEVEL": "0"}, {"SKX_TWO_LEVEL": "1", "SKX_RANK_LIVE": "0"},
                                   {"SKX_SCAN_NT": "0"}, {"SKX_SCAN_NT": "6"}, {"SKX_SCAN_NT": "0", "SKX_PASS_READS": "100"},
                                   {"SKX_SCAN_BIGSLICE": "1"}, {"SKX_SCAN_BIGSLICE": "1", "SKX_PASS_READS": "64"},
                                   {"SKX_CAND": "0"}, {"SKX_CAND": "0", "SKX_PASS_READS": "64"}, {"SKX_RARE_MAX": "0"}, {"SKX_RARE_MAX": "3"},
                                   {"SKX_RARE_MAX": "0", "SKX_CAND": "0"}, {"SKX_TABLE_LEGACY": "0"}, {"SKX_TABLE_LEGACY": "100"},
                                   {"SKX_RARE_MAX": "100000"}, {"SKX_RARE_MAX": "100000", "SKX_RARE_CHUNK": "1"}, {"SKX_RARE_DIRECT": "0"},
                                   {"SKX_RARE_MAX": "100000", "SKX_RARE_DIRECT": "0"}, {"SKX_TAIL_SCALE": "1"}, {"SKX_FIRST_GROUP": "1"},
                                   {"SKX_RARE_MAX": "100000", "SKX_LONG_ROWS_T": "0"}, {"SKX_STATIC_DENSE": "0"},
                                   {"SKX_STATIC_DENSE": "0", "SKX_PASS_READS": "64"}, {"SKX_RARE_MAX": "3", "SKX_STATIC_DENSE": "0"},
                                   {"SKX_RARE_MAX": "3", "SKX_PASS_READS": "37"}, {"SKX_PATTERNS": "0"}, {"SKX_LONG_ROWS": "0"}])
def test_forced_pass_partition_and_pipeline_depth(gpu, knobs):
    """Several passes per push (the path that needs the per-read pair offsets on the host), the pipeline depths, the
    unfiltered dictionary, the ranking without its per-word live flags, the ranking lanes forced to one / four, the
    pair gather on the scan stream, the ranking's counts in two levels / one and the lean scan's results as slabs (SKX_SCAN_NT=0) /
    straight into M (6; the default), the lean scan's instance for large slices (SKX_SCAN_BIGSLICE=1: 510 entries, three-entry
    probe), and round 5's machinery switched off or forced -- no compact ranking on the candidates (SKX_CAND=0), no rare-hash index or
    one for nearly every hash (SKX_RARE_MAX), the table always / never out of the ranking chains (SKX_TABLE_LEGACY), the rare rows into the
    group-major matrix one rank group per turn (SKX_RARE_CHUNK=1) or through M and the transpose as before (SKX_RARE_DIRECT=0), the list
    walks of a closing pass at their ordinary size (SKX_TAIL_SCALE=1), first groups of one batch, no transposed bit rows (SKX_LONG_ROWS_T=0: the candidates' long-list rows found row by row),
    round 6's static dense dictionary off (SKX_STATIC_DENSE=0: per-pass dictionaries, the scan behind them) or with nearly every hash in it
    (SKX_RARE_MAX=3), no patterns, no bit rows at all -- all give the oracle's rows."""
    from helpers import exp_env
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py")], env=exp_env(**knobs),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "variant ok" in out.stdout


def test_the_product_library_reads_no_environment_knob(gpu):
    """SKX_SCAN_ABLATE=1 (no result write-back) and SKX_NO_FILTER make the EXPERIMENTS build compute something else --
    the product library must not even look: same parity run, knobs set, default library -> the oracle's rows; the
    experiments build under SKX_SCAN_ABLATE=1 -> wrong rows (which proves the knob is live there, i.e. that this test
    would notice a product build that still read it)."""
    from helpers import exp_env
    script = os.path.join(ROOT, "tests", "variant_check.py")
    # (SKX_RARE_MAX=0 for the experiments run: without the rare-hash index every query hash of these small panels goes through the scan --
    # with it none does, and an ablated scan would change nothing)
    env = dict(os.environ, SKX_SCAN_ABLATE="1", SKX_NO_FILTER="1", SKX_PIPELINE="1", SKX_PASS_READS="37", SKX_RARE_MAX="0")
    env.pop("SKX_LIB_PATH", None)
    out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "variant ok" in out.stdout, out.stdout + out.stderr
    bad = subprocess.run([sys.executable, script], env=exp_env(SKX_SCAN_ABLATE=1, SKX_RARE_MAX=0), capture_output=True, text=True)
    assert bad.returncode != 0, "the experiments build ignored SKX_SCAN_ABLATE=1"
