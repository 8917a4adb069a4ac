"""Bit-exact parity of the HIP path (through the C ABI) with the CPU oracle.

The oracle restates the reference's _sum_of_shared_hashes loop (src/sketchy.rs:317-356);
integer work throughout, so every comparison is exact equality.
"""
import numpy as np
import pytest

from helpers import assert_stream_equal, pack_reads, workload
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def run_both(ref, bases, offsets, top, k=16, seed=0, col_len=None, batches=1, want_shared=True, want_sketches=True, s=None):
    """s: the size reads are sketched with when it is not the matrix width (the reference: |sketch 0|)"""
    from sketchy_amd import api
    hashes = ref["ref"] if isinstance(ref, dict) else ref
    n, stride = hashes.shape
    s = stride if s is None else s
    col_len = np.full(n, stride, np.uint32) if col_len is None else col_len
    exp = orc.stream(k, seed, s, hashes, col_len, bases, offsets, top_k=max(top, 1), want_shared=True, want_sketches=True)
    R = api.ReferenceSketch(hashes, col_len, k=k, seed=seed, s=s)
    n_reads = len(offsets) - 1
    S = api.SumOfSharedHashes(R, top=top, max_batch_reads=max(1, n_reads), max_batch_bases=max(1, len(bases)))
    # feed in `batches` pushes to exercise table continuity
    cuts = np.linspace(0, n_reads, batches + 1).astype(int)
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            parts.append(S.push(bases, offsets[a:b + 1], want_shared=want_shared, want_sketches=want_sketches))
    got = {}
    for key in ("topk_idx", "topk_sum", "shared", "sketches", "sketch_len"):
        vals = [p[key] for p in parts if p.get(key) is not None]
        got[key] = np.concatenate(vals) if vals else None
    got["cum"] = S.table()
    assert S.reads == n_reads
    return got, exp, R, S


def check(ref, bases, offsets, top, **kw):
    got, exp, R, S = run_both(ref, bases, offsets, top, **kw)
    np.testing.assert_array_equal(got["cum"], exp["cum"], err_msg="running table")
    assert_stream_equal(got, exp, top)
    return got, exp, R, S


def test_c0_plumbing_config(gpu):
    """BASELINE configs[0]: 1k synthetic 1.5 kb reads vs 500-genome s=1000 k=16 sketch."""
    ref, bases, offsets = workload(500, 1000, 1000)
    got, exp, R, S = check(ref, bases, offsets, top=5)
    assert exp["shared"].max() > 0  # the workload really shares hashes
    idx, sm = S.rank(10)
    order = orc.stable_rank(exp["cum"])[:10]
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_array_equal(sm, exp["cum"][order])


def test_c0_truth_strain_workload(gpu):
    """C0 on SURVEY.md 8(d)'s generator: SNP clone tree with real 16-mer hashes, reads from ONE truth strain --
    strain-specific matches occur and a leader emerges (the regime of a real sample), per-read counts included."""
    from helpers import workload_snp
    ref, bases, offsets = workload_snp(500, 1000, 1000, rng_seed=3)
    got, exp, R, S = check(ref, bases, offsets, top=5)
    lin, ti = ref["lineage"], ref["truth_index"]
    assert lin[exp["topk_idx"][-1, 0]] == lin[ti] and exp["cum"][ti] == exp["cum"].max()
    sh = exp["shared"].astype(np.int64).sum(axis=0)
    assert (sh[lin == lin[ti]] < sh[ti]).any()        # hashes only the truth strain (and not all its mates) holds


@pytest.mark.parametrize("mode", ["pool", "snp"])
def test_c1_full_config_every_row(gpu, mode):
    """BASELINE configs[1] at its full size: 100 000 reads x 1.5 kb vs 5 000 genomes x s=1000, every row and the table
    against the fast checker (orc_stream_fast, pinned against the literal loop in tests/test_oracle.py); the stream is
    enqueued in uneven batches that share passes."""
    from helpers import workload_snp
    from sketchy_amd import api
    from test_gpu_enqueue import _enqueue_stream
    n = 100000
    if mode == "snp":
        ref, bases, offsets = workload_snp(5000, 1000, n, rng_seed=17)
    else:
        ref, bases, offsets = workload(5000, 1000, n, rng_seed=17)
    exp = orc.stream_fast(16, 0, 1000, ref["ref"], ref["col_len"], bases, offsets, top_k=1)
    R = api.ReferenceSketch(ref["ref"])
    cuts = [0, 16384, 16385, 50000, 82768, n]
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=33615, max_batch_bases=int(np.max(np.diff(offsets[cuts].astype(np.int64)))))
    idx, val = _enqueue_stream(S, bases, offsets, cuts, 1)
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    if mode == "snp":
        assert exp["topk_idx"][-1, 0] == ref["truth_index"]


@pytest.mark.parametrize("top", [1, 3])
def test_candidates_shrink_and_grow_mid_stream(gpu, top):
    """Round 5: the per-read ranking of a batch runs on its CANDIDATES (genomes whose value at the end of the batch reaches the
    top-th best value at its start) when a species has at most 1024 of them, else on every genome.  Three unrelated families of
    900 genomes: a fresh table has 2 700 candidates (full ranking), reads of family 0 soon leave its 900 (compact), then reads
    of family 1 make it catch up -- both families are candidates again (full), until family 1 leads alone (compact).  Every
    row and the table against the oracle; both kinds of batches must have occurred.  Enqueued in uneven batches that share passes."""
    from sketchy_amd import api
    from test_gpu_enqueue import _enqueue_stream
    fams = [workload(900, 400, 2600, read_len=500, rng_seed=4000 + i) for i in range(3)]
    hashes = np.concatenate([f[0]["ref"] for f in fams])
    perm = np.random.default_rng(9).permutation(len(hashes))
    hashes = np.ascontiguousarray(hashes[perm])

    def reads_of(i, a, b):
        bs, of = fams[i][1], fams[i][2]
        return [bs[int(of[j]):int(of[j + 1])].tobytes() for j in range(a, b)]
    reads = reads_of(0, 0, 1200) + reads_of(1, 0, 2600) + reads_of(2, 0, 300) + reads_of(1, 0, 400)
    bases, offsets = pack_reads(reads)
    exp = orc.stream_fast(16, 0, 400, hashes, None, bases, offsets, top_k=top)
    R = api.ReferenceSketch(hashes)
    n = len(reads)
    cuts = list(range(0, n, 300)) + [n]
    S = api.SumOfSharedHashes(R, top=top, max_batch_reads=300, max_batch_bases=int(np.max(np.diff(offsets[cuts].astype(np.int64)))))
    idx, val = _enqueue_stream(S, bases, offsets, cuts, top)
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    st = S.stats()
    assert st["batches_compact"] > 0 and st["batches_full"] > 0, st
    fam_of = perm // 900
    lead = fam_of[exp["topk_idx"][:, 0]]
    assert lead[1100] == 0 and lead[-1] == 1


def test_multiple_pushes_continue_the_table(gpu):
    ref, bases, offsets = workload(300, 500, 257, rng_seed=7)
    check(ref, bases, offsets, top=3, batches=5)


def test_top1_default_and_table_only(gpu):
    ref, bases, offsets = workload(130, 400, 100, rng_seed=3)
    check(ref, bases, offsets, top=1)
    got, exp, R, S = run_both(ref, bases, offsets, top=0)
    np.testing.assert_array_equal(got["cum"], exp["cum"])


@pytest.mark.parametrize("k,seed", [(16, 42), (21, 0), (11, 7), (32, 1), (15, 3), (17, 9), (8, 0)])
def test_other_kmer_sizes_and_seeds(gpu, k, seed):
    ref, bases, offsets = workload(70, 300, 60, read_len=400, k=k, seed=seed, rng_seed=11 + k)
    check(ref, bases, offsets, top=2, k=k, seed=seed)


def test_s_larger_than_read(gpu):
    """s=10000-style regime: m < s, the sketch is every distinct k-mer hash of the read."""
    ref, bases, offsets = workload(64, 3000, 80, read_len=700, genome_len=400000, rng_seed=5)
    got, exp, _, _ = check(ref, bases, offsets, top=4)
    assert (exp["sketch_len"] < 3000).all()


def test_edge_reads(gpu):
    ref, _, _ = workload(40, 200, 1, read_len=300, rng_seed=21)
    g = ref["genome"].tobytes()
    reads = [
        b"",                                   # empty read still counts and ranks
        b"ACGT",                               # shorter than k
        g[100:115],                            # exactly k-1
        g[100:116],                            # exactly k
        g[1000:1400].lower(),                  # lower case is folded
        g[2000:2200] + b"N" + g[2201:2500],    # N breaks windows
        g[3000:3100] + b"\n" + g[3100:3200] + b"\r\n" + g[3200:3300] + b" \t",  # whitespace is removed
        g[4000:4300].replace(b"T", b"U"),      # U -> T
        b"RYKMSWBDHV" * 20,                    # IUPAC -> N: no valid k-mer
        b"A" * 500,                            # one distinct k-mer
        b"ACGTACGTACGTACGTACGT" * 10,          # palindromic / repeated k-mers
        g[5000:5000 + 2063],                   # the longest read the wave sketcher takes (2048 k-mers)
        g[7000:7300] + b"-" + g[7301:7600] + b"." + g[7601:7700] + b"~",
    ]
    bases, offsets = pack_reads(reads)
    check(ref, bases, offsets, top=3)


def test_ragged_columns_and_ties(gpu):
    rng = np.random.default_rng(9)
    ref, bases, offsets = workload(90, 256, 50, read_len=600, rng_seed=31)
    hashes = ref["ref"].copy()
    # duplicate genomes -> exact ties, resolved by reference order (stable sort, src/sketchy.rs:348)
    hashes[10] = hashes[3]
    hashes[77] = hashes[3]
    hashes[40] = hashes[41]
    col_len = rng.integers(0, 257, size=90).astype(np.uint32)
    col_len[3] = col_len[10] = col_len[77] = 256
    col_len[5] = 0  # an empty reference sketch
    col_len[6] = 1
    check(hashes, bases, offsets, top=7, col_len=col_len)


def test_no_shared_hashes_all_ties(gpu):
    """Reads unrelated to the reference: every sum is 0, rows are the first genomes in order."""
    ref, _, _ = workload(33, 128, 1, rng_seed=41)
    rng = np.random.default_rng(1)
    reads = [bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 300)]) for _ in range(20)]
    bases, offsets = pack_reads(reads)
    got, exp, _, _ = check(ref, bases, offsets, top=5)
    np.testing.assert_array_equal(got["topk_idx"], np.tile(np.arange(5, dtype=np.uint32), (20, 1)))


def test_single_genome_and_top_equals_n(gpu):
    ref, bases, offsets = workload(1, 64, 10, read_len=300, genome_len=30000, rng_seed=51)
    check(ref, bases, offsets, top=1)
    ref, bases, offsets = workload(5, 64, 10, read_len=300, genome_len=30000, rng_seed=52)
    check(ref, bases, offsets, top=5)


def test_genome_counts_across_tile_boundaries(gpu):
    for n in (255, 256, 257, 513):
        ref, bases, offsets = workload(n, 96, 24, read_len=500, genome_len=40000, rng_seed=60 + n)
        check(ref, bases, offsets, top=2, want_sketches=False)


def test_sentinel_like_reference_hashes(gpu):
    """Reference hashes at the very top of the u64 range (the values the kernels use as padding /
    empty markers) must still be matched exactly."""
    ref, bases, offsets = workload(20, 64, 12, read_len=400, genome_len=30000, rng_seed=71)
    hashes = ref["ref"].copy()
    sk = orc.sketch(bases[int(offsets[0]):int(offsets[1])].tobytes(), 16, 0, 10 ** 6)
    hashes[2, -2:] = [0xFFFFFFFFFFFFFFFE, 0xFFFFFFFFFFFFFFFF]
    hashes[4, -1] = 0xFFFFFFFFFFFFFFFF
    hashes[7, -1] = int(sk[-1])  # a genome that contains the first read's largest hash
    hashes[7] = np.sort(hashes[7])
    assert len(np.unique(hashes[7])) == 64
    check(hashes, bases, offsets, top=3)


def test_errors(gpu):
    from sketchy_amd import _lib, api
    ref, bases, offsets = workload(8, 32, 4, read_len=200, genome_len=30000, rng_seed=81)
    R = api.ReferenceSketch(ref["ref"])
    with pytest.raises(_lib.SketchyHipError) as e:  # the reference panics on top > N (src/sketchy.rs:391)
        api.SumOfSharedHashes(R, top=9)
    assert e.value.code == _lib.ERR_INVALID
    bad = ref["ref"].copy()
    bad[3, 5], bad[3, 6] = bad[3, 6], bad[3, 5]
    with pytest.raises(_lib.SketchyHipError) as e:
        api.ReferenceSketch(bad)
    assert e.value.code == _lib.ERR_UNSORTED
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=2, max_batch_bases=10 ** 6)
    with pytest.raises(_lib.SketchyHipError) as e:
        S.push(bases, offsets)
    assert e.value.code == _lib.ERR_CAPACITY
    with pytest.raises(_lib.SketchyHipError) as e:
        api.ReferenceSketch(ref["ref"], k=33)
    assert e.value.code == _lib.ERR_INVALID


def test_table_add_reset_and_shard_invariance(gpu):
    """Reads shard across GPUs; the table is an integer sum, so shard tables add up exactly and a
    shard seeded with the totals of earlier shards reproduces the single-stream ranks."""
    from sketchy_amd import api
    ref, bases, offsets = workload(150, 300, 120, read_len=800, rng_seed=91)
    exp = orc.stream(16, 0, 300, ref["ref"], ref["col_len"], bases, offsets, top_k=3)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=3, max_batch_reads=120, max_batch_bases=len(bases))
    a = S.push(bases, offsets[:61])
    t0 = S.table()
    S.reset()
    assert S.reads == 0 and not S.table().any()
    b = S.push(bases, offsets[60:])
    t1 = S.table()
    np.testing.assert_array_equal(t0 + t1, exp["cum"])
    # second shard again, offset by the first shard's totals -> identical per-read rows
    S.reset()
    S.table_add(t0)
    b2 = S.push(bases, offsets[60:])
    np.testing.assert_array_equal(np.concatenate([a["topk_idx"], b2["topk_idx"]]), exp["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([a["topk_sum"], b2["topk_sum"]]), exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])


def test_common_hashes_operator(gpu):
    """`sketchy shared`: all-pairs _common_hashes; self-vs-self = sketch size (docs/index.md:145-149)."""
    from sketchy_amd import api
    ref, _, _ = workload(60, 500, 1, rng_seed=101)
    R = api.ReferenceSketch(ref["ref"])
    q = ref["ref"][:17]
    got = R.common_hashes(q)
    exp = np.array([[orc.common_hashes(ref["ref"][g], q[i]) for g in range(60)] for i in range(17)], np.uint32)
    np.testing.assert_array_equal(got, exp)
    assert (np.diag(got[:, :17]) == 500).all()


def test_sketch_reads_operator(gpu):
    from sketchy_amd import api
    ref, bases, offsets = workload(4, 64, 40, read_len=1500, genome_len=50000, rng_seed=111)
    for s in (10, 1000, 5000):
        sk, sl = api.sketch_reads(bases, offsets, k=16, seed=42, s=s)
        for r in range(40):
            e = orc.sketch(bases[int(offsets[r]):int(offsets[r + 1])].tobytes(), 16, 42, s)
            assert sl[r] == len(e)
            np.testing.assert_array_equal(sk[r, :len(e)], e)


def test_frozen_golden_fixture(gpu):
    """tests/golden/stream_small.npz (inputs + expected outputs frozen by tests/golden/make_golden.py)."""
    import os
    from sketchy_amd import api
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stream_small.npz"))
    R = api.ReferenceSketch(z["hashes"], z["col_len"], k=int(z["k"]), seed=int(z["seed"]))
    S = api.SumOfSharedHashes(R, top=4, max_batch_reads=len(z["offsets"]) - 1, max_batch_bases=len(z["bases"]))
    got = S.push(z["bases"], z["offsets"], want_shared=True, want_sketches=True)
    for key in ("topk_idx", "topk_sum", "shared", "sketches", "sketch_len"):
        np.testing.assert_array_equal(got[key], z[key], err_msg=key)
    np.testing.assert_array_equal(S.table(), z["cum"])


def _wrap(seq: bytes, width=60) -> bytes:
    return b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + b"\n"


def test_long_reads_mixed_lengths(gpu):
    """BASELINE configs[4] style: log-normal read lengths (200 .. 50 000); reads with more than 2048 k-mers
    take the block-per-read kernels.  s=300 < #k-mers (truncation matters) and s=5000 > most reads."""
    from sketchy_amd import synth
    for s, n_gen, rng_seed in ((300, 70, 5), (5000, 40, 6)):
        ref = synth.make_reference(n_gen, s, genome_len=max(120000, 280 * s), rng_seed=rng_seed, device="numpy")
        bases, offsets = synth.make_reads(ref["genome"], 60, 3000, err=0.03, rng_seed=rng_seed + 100, lognormal_sigma=1.0,
                                          min_len=200, max_len=50000)
        lens = np.diff(offsets.astype(np.int64))
        assert lens.max() > 2063 and lens.min() < 2063
        check(ref, bases, offsets, top=3)
        check(ref, bases, offsets, top=1, batches=3, want_sketches=False)  # production (in-range only) sketch path


def test_long_read_edge_cases(gpu):
    ref, _, _ = workload(30, 400, 1, read_len=300, genome_len=150000, rng_seed=141)
    g = ref["genome"].tobytes()
    reads = [
        g[0:2064],                                   # one base past the wave sketcher's limit
        g[1000:2063 + 1000],                         # exactly at the limit
        _wrap(g[5000:45000]),                        # multi-line FASTA style: newlines are removed
        g[50000:70000].lower(),
        g[70000:75000] + b"N" * 40 + g[75040:90000],  # N run inside a long read
        b"ACGT" * 3000,                              # 12 kb of a 4-periodic sequence: 4 distinct k-mers... (2 canonical)
        b"N" * 5000,
        g[90000:140000],                             # 50 kb
        b"",
        g[200:1700],
    ]
    bases, offsets = pack_reads(reads)
    check(ref, bases, offsets, top=2)
    check(ref, bases, offsets, top=1, want_sketches=False)


def _split_reads():
    """Reads around and far beyond the 8192-base split point of the production sketcher (kLongSplit = 4 chunks of 2048 raw
    bytes): a read cut into segments must give the serial loop's hashes whatever sits on the chunk borders."""
    ref, _, _ = workload(40, 500, 1, read_len=300, genome_len=400000, rng_seed=909)
    g = ref["genome"].tobytes()
    ws = b" \t\r\n" * 50                                   # 200 bytes of whitespace
    reads = [
        g[0:8192],                                            # at the limit: one wave, four chunks
        g[100:100 + 8193],                                    # one base beyond: five segments, the last holds ONE raw byte
        g[9000:9000 + 10240],                                 # a multiple of the chunk size
        g[20000:20000 + 8192 + 15],                           # the last segment is shorter than k
        _wrap(g[30000:90000]),                                # FASTA lines of 70 + newline: borders fall between, on and behind newlines
        g[100000:104000] + ws + g[104000:112000],             # 200 removed bytes in a row, across a chunk border: the carry scan loops
        g[120000:124090] + b"\n" * 6 + g[124090:131000],      # whitespace right AT a border (raw offset 4096)
        g[140000:142040] + b"N" * 20 + g[142060:150500],      # an N run over the first border
        (g[150000:156143] + b"n" + g[156144:160000]).lower(), # lower case, an n one base before a border
        g[160000:260000],                                     # 100 kb: 49 segments
        g[200000:399000],                                     # 199 kb: 98 segments (more than one wave's worth of counts in the merge)
        b"ACGT" * 3000,                                       # 12 kb with two distinct canonical k-mers
        b"N" * 9000,
        b"\n" * 9000,                                         # nothing but removed bytes
        g[1000:2500], b"", g[5000:5000 + 2064],
    ]
    return ref, reads


def test_long_reads_split_over_waves(gpu):
    """Production sketch path (no debug sketches): reads beyond 8192 bases are hashed by one wave per 2048-byte chunk and
    merged (sketch_merge_kernel) -- rows, per-read counts and table against the oracle; in one batch, in several, through
    the device-resident entry points and as 4-bit packed input."""
    from sketchy_amd import api
    ref, reads = _split_reads()
    bases, offsets = pack_reads(reads)
    n = len(reads)
    exp = orc.stream(16, 0, 500, ref["ref"], np.full(40, 500, np.uint32), bases, offsets, top_k=2, want_shared=True)
    assert exp["shared"][10].max() > 20  # the 199 kb read really shares hashes
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=2, max_batch_reads=n, max_batch_bases=len(bases))
    got = S.push(bases, offsets, want_shared=True)
    np.testing.assert_array_equal(got["shared"], exp["shared"])
    np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"])
    np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    st = S.stats()
    n_long = sum(1 for r in reads if len(r) > 8192)
    assert st["reads_split_over_waves"] == n_long and st["reads_block_sketcher"] == 0
    assert st["read_segments"] == sum((len(r) + 2047) // 2048 for r in reads if len(r) > 8192)
    # the debug path (full sketches: the block sketcher takes the long reads) agrees, and so do three batches + device pushes
    S.reset()
    full = S.push(bases, offsets, want_shared=True, want_sketches=True)
    np.testing.assert_array_equal(full["shared"], exp["shared"])
    S.reset()
    parts = [_push_device(S, bases, offsets[a:b + 1], 2) for a, b in ((0, 5), (5, 11), (11, n))]
    np.testing.assert_array_equal(np.concatenate([p_["topk_idx"] for p_ in parts]), exp["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([p_["topk_sum"] for p_ in parts]), exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    # 4-bit packed input (no whitespace in that format: the packer drops it), starting on an odd nibble
    packed, poff = api.pack_reads(bases, offsets, first_nibble=1)
    P = api.SumOfSharedHashes(R, top=2, max_batch_reads=n, max_batch_bases=int(poff[-1]) + 2)
    P.set_packed_input(True)
    gp = P.push(packed, poff, want_shared=True)
    np.testing.assert_array_equal(gp["shared"], exp["shared"])
    np.testing.assert_array_equal(gp["topk_idx"], exp["topk_idx"])
    np.testing.assert_array_equal(P.table(), exp["cum"])
    assert P.stats()["reads_split_over_waves"] > 0


def test_long_reads_split_other_k_and_dense_segments(gpu):
    """k = 21 / 11 / 32 (the carry in front of a segment is k - 1 codes) and a reference dense enough that segments overflow
    their 64-entry slots: those reads fall through to the block sketcher, same rows."""
    from sketchy_amd import api
    for k, seed in ((21, 3), (11, 0), (32, 5)):
        ref, _, _ = workload(20, 200, 1, read_len=300, k=k, seed=seed, genome_len=150000, rng_seed=250 + k)
        g = ref["genome"].tobytes()
        bases, offsets = pack_reads([g[0:9000], g[10000:10500], _wrap(g[20000:60000]), g[60000:60000 + 8193 + k]])
        check(ref, bases, offsets, top=2, k=k, seed=seed, want_sketches=False)
    ref, _, _ = workload(12, 6000, 1, read_len=300, genome_len=40000, rng_seed=261)  # ~15 % of the hash space is in range
    g = ref["genome"].tobytes()
    bases, offsets = pack_reads([g[0:20000], g[100:1600], g[15000:39000]])
    got, exp, R, S = check(ref, bases, offsets, top=1, want_sketches=False)
    assert S.stats()["reads_block_sketcher"] == 2 and S.stats()["reads_split_over_waves"] == 2


def synth_reference_with_many_keys(n_keys=90000):
    """two columns of hashes of REAL 16-mers (a random sequence's), more of them than a 32 KB table takes at 4 bits per key"""
    from sketchy_amd import synth
    g = synth.random_genome(40 * n_keys, np.random.default_rng(5))
    h = np.unique(synth.canonical_kmer_hashes(g, 16, 0))[:n_keys]
    return np.stack([h, h])


@pytest.mark.parametrize("seed", [0, 42])
def test_kmer_prefilter_gives_the_oracles_rows(gpu, seed):
    """k = 16 references carry a Bloom table over the canonical 16-mers whose hash can meet them; the production sketcher
    hashes only the windows that pass it -- for reads with at most s windows (no truncation possible); longer reads take
    the plain loop.  Rows, per-read counts and table against the oracle with the table on (default) and off, around the
    s + 15 base boundary, with reads that are both split over waves and prefiltered (8193 .. s + 15 bases), N / lower case /
    whitespace / U, and a dense reference where most windows survive the table (the survivor queue drains mid-chunk)."""
    from sketchy_amd import api
    ref, _, _ = workload(60, 9000, 1, read_len=300, seed=seed, genome_len=300000, rng_seed=7000 + seed)
    g = ref["genome"].tobytes()
    rng = np.random.default_rng(12)
    reads = [g[a:a + n] for a, n in ((int(rng.integers(0, 280000)), int(n)) for n in
                                     [15, 16, 17, 64, 300, 1500, 1500, 2047, 2048, 2049, 2063, 2064, 4000, 8192, 8193, 9000, 9014, 9015, 9016, 12000])]
    reads += [g[1000:2500].lower(), g[3000:3700] + b"N" + g[3701:4500], _wrap(g[5000:13000]), g[20000:21500].replace(b"T", b"U"),
              b"ACGT" * 500, b"", g[50000:50040] + b"\n\n" + g[50040:50300]]
    bases, offsets = pack_reads(reads)
    exp = orc.stream(16, seed, 9000, ref["ref"], np.full(60, 9000, np.uint32), bases, offsets, top_k=2, want_shared=True)
    assert exp["shared"].max() > 3
    try:
        for on in (1, 0):
            api.set_option("kmer_prefilter", on)
            R = api.ReferenceSketch(ref["ref"], seed=seed)
            keys, nbytes = R.kmer_filter
            assert (keys > 0 and nbytes >= 4096) if on else (keys == 0 and nbytes == 0)
            S = api.SumOfSharedHashes(R, top=2, max_batch_reads=len(reads), max_batch_bases=len(bases))
            got = S.push(bases, offsets, want_shared=True)
            np.testing.assert_array_equal(got["shared"], exp["shared"], err_msg=f"prefilter {on}")
            np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"], err_msg=f"prefilter {on}")
            np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"], err_msg=f"prefilter {on}")
            np.testing.assert_array_equal(S.table(), exp["cum"])
            S.reset()
            full = S.push(bases, offsets, want_shared=True, want_sketches=True)   # debug path: plain loop, full sketches
            np.testing.assert_array_equal(full["shared"], exp["shared"])
            S.close(); R.close()
    finally:
        api.set_option("kmer_prefilter", 0)
    # dense: s = 20 000 of a 30 kb genome -- two thirds of all windows are reference k-mers
    ref, bases, offsets = workload(12, 20000, 60, read_len=1500, seed=seed, genome_len=30000, rng_seed=7100 + seed)
    try:
        api.set_option("kmer_prefilter", 1)
        got, exp, R, S = check(ref, bases, offsets, top=1, seed=seed, want_sketches=False)
    finally:
        api.set_option("kmer_prefilter", 0)
    assert R.kmer_filter[0] > 10000 and exp["shared"].max() > 256
    # policy 2: only a table that fits the L1 caches (32 KB at 8, else 4, bits per key) is built
    if seed == 0:
        small, _, _ = workload(6, 2000, 1, read_len=300, rng_seed=7200)
        large = synth_reference_with_many_keys()
        try:
            api.set_option("kmer_prefilter", 2)
            Rs, Rl = api.ReferenceSketch(small["ref"]), api.ReferenceSketch(large)
        finally:
            api.set_option("kmer_prefilter", 0)
        assert 2000 <= Rs.kmer_filter[0] < 8000 and 0 < Rs.kmer_filter[1] <= 32768, Rs.kmer_filter
        assert Rl.kmer_filter == (0, 0)
        Rs.close(); Rl.close()


def test_long_reads_other_k(gpu):
    for k, seed in ((21, 3), (11, 0)):
        ref, _, _ = workload(20, 200, 1, read_len=300, k=k, seed=seed, genome_len=100000, rng_seed=150 + k)
        g = ref["genome"].tobytes()
        bases, offsets = pack_reads([g[0:9000], g[10000:10500], g[20000:60000]])
        check(ref, bases, offsets, top=2, k=k, seed=seed)


def test_sketch_reads_operator_long(gpu):
    from sketchy_amd import api, synth
    rng = np.random.default_rng(3)
    g = synth.random_genome(120000, rng).tobytes()
    reads = [g[0:30000], g[30000:31000], _wrap(g[40000:100000]), g[100000:102063], g[100000:102064]]
    bases, offsets = pack_reads(reads)
    for s in (50, 1000, 100000):
        sk, sl = api.sketch_reads(bases, offsets, k=16, seed=0, s=s)
        for r, rd in enumerate(reads):
            e = orc.sketch(rd, 16, 0, s)
            assert sl[r] == len(e), (s, r)
            np.testing.assert_array_equal(sk[r, :len(e)], e)


def test_dense_reference_most_read_hashes_in_range(gpu):
    """A reference whose sketches cover most of the hash space (s close to the genome's k-mer count): nearly
    every read hash can match, so the in-range fast sketcher overflows its small buffer and the read is redone
    by the full-size variant; pairs per read are in the hundreds."""
    ref, bases, offsets = workload(12, 20000, 40, read_len=1500, genome_len=30000, rng_seed=171)
    got, exp, _, _ = check(ref, bases, offsets, top=2, want_sketches=False, batches=2)
    assert exp["shared"].max() > 256
    check(ref, bases, offsets, top=1)


def test_dense_reference_long_reads_take_the_block_sketcher(gpu):
    """Long reads against a reference that covers most of the hash space: thousands of in-range hashes per read, more
    than a wave's 2048 slots -- the production (in-range) sketch path hands them to the block sketcher, which must
    compact mid-read (16 384 slots) and, with s = 20 000 > 12 288, select in two passes."""
    from sketchy_amd import synth
    ref = synth.make_reference(12, 20000, genome_len=30000, rng_seed=181, device="numpy")
    bases, offsets = synth.make_reads(ref["genome"], 12, 9000, err=0.02, rng_seed=182, lognormal_sigma=0.8, min_len=300, max_len=29000)
    assert np.diff(offsets.astype(np.int64)).max() > 18000
    got, exp, _, _ = check(ref, bases, offsets, top=1, want_sketches=False, want_shared=True)
    assert exp["shared"].max() > 2048
    check(ref, bases, offsets, top=2)


def test_leader_changes_mid_pass_and_ties(gpu):
    """The per-read top-1 replay prunes genomes that cannot lead within a 64-read segment, against a bound taken
    from the genome that led when the pass began.  Here the leader moves from one clone family to another in the
    middle of a single pass (and back in a second push), with duplicated genomes giving exact ties for the lead."""
    refA, basesA, offsA = workload(300, 500, 1100, rng_seed=61)
    refB, basesB, offsB = workload(300, 500, 1100, rng_seed=67)
    hashes = np.concatenate([refA["ref"], refB["ref"]])
    hashes[450] = hashes[310]          # ties inside family B (resolved by reference order)
    hashes[7] = hashes[299]            # and inside family A, across rank groups of 512 genomes? (same group here)
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(hashes))  # families interleaved over words and groups
    hashes = np.ascontiguousarray(hashes[perm])
    ra = [basesA[int(offsA[i]):int(offsA[i + 1])].tobytes() for i in range(1100)]
    rb = [basesB[int(offsB[i]):int(offsB[i + 1])].tobytes() for i in range(1100)]
    # A leads, B overtakes mid-pass, then they alternate read by read, then A catches up again
    reads = ra[:150] + rb[:500] + [x for p in zip(ra[150:350], rb[500:700]) for x in p] + ra[350:1100]
    bases, offsets = pack_reads(reads)
    got, exp, _, _ = check(hashes, bases, offsets, top=1, want_shared=False, want_sketches=False)
    lead = exp["topk_idx"][:, 0]
    fam = (perm[lead] >= 300).astype(int)
    assert fam[100] == 0 and fam[600] == 1 and fam[-1] == 0 and (np.diff(fam) != 0).sum() >= 2
    check(hashes, bases, offsets, top=1, batches=3, want_shared=False, want_sketches=False)
    check(hashes, bases, offsets, top=4, batches=2, want_shared=False, want_sketches=False)


def _push_device(S, bases, offsets, top):
    """skx_stream_push_device with everything resident on the device; returns rows like push()."""
    from sketchy_amd import api
    n = len(offsets) - 1
    d_b = api.DeviceBuffer.from_numpy(bases if len(bases) else np.zeros(1, np.uint8))
    d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets, np.uint64))
    d_i = api.DeviceBuffer(max(1, n * top) * 4)
    d_s = api.DeviceBuffer(max(1, n * top) * 8)
    try:
        S.push_device(d_b.ptr, d_o.ptr, n, int(offsets[-1] - offsets[0]), d_i.ptr, d_s.ptr)
        S.sync()
        return dict(topk_idx=d_i.to_numpy(np.uint32, (n, top)), topk_sum=d_s.to_numpy(np.uint64, (n, top)))
    finally:
        for d in (d_b, d_o, d_i, d_s):
            d.free()


def test_push_device_matches_oracle_short_and_long_reads(gpu):
    """The device-resident entry point looks at the offsets on the device (no host copy): short reads take the
    one-synchronisation fast path, a batch with long reads the fallback that fetches the offsets after all."""
    from sketchy_amd import api, synth
    ref, bases, offsets = workload(200, 400, 300, rng_seed=71)
    exp = orc.stream(16, 0, 400, ref["ref"], np.full(200, 400, np.uint32), bases, offsets, top_k=2)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=2, max_batch_reads=300, max_batch_bases=len(bases))
    got = _push_device(S, bases, offsets[:181], 2)
    got2 = _push_device(S, bases, offsets[180:], 2)  # offsets need not start at 0
    np.testing.assert_array_equal(np.concatenate([got["topk_idx"], got2["topk_idx"]]), exp["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([got["topk_sum"], got2["topk_sum"]]), exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    # mixed lengths
    ref = synth.make_reference(60, 300, genome_len=120000, rng_seed=72, device="numpy")
    bases, offsets = synth.make_reads(ref["genome"], 50, 3000, err=0.03, rng_seed=73, lognormal_sigma=1.0, min_len=200, max_len=40000)
    assert np.diff(offsets.astype(np.int64)).max() > 2063
    exp = orc.stream(16, 0, 300, ref["ref"], np.full(60, 300, np.uint32), bases, offsets, top_k=1)
    R2 = api.ReferenceSketch(ref["ref"])
    S2 = api.SumOfSharedHashes(R2, top=1, max_batch_reads=50, max_batch_bases=len(bases))
    got = _push_device(S2, bases, offsets, 1)
    np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"])
    np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"])
    np.testing.assert_array_equal(S2.table(), exp["cum"])


def test_push_device_errors_leave_the_table_alone(gpu):
    from sketchy_amd import api, _lib
    ref, bases, offsets = workload(40, 200, 20, rng_seed=75)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=20, max_batch_bases=len(bases))
    bad = offsets.copy()
    bad[7], bad[8] = bad[8], bad[7]  # not monotonic
    with pytest.raises(_lib.SketchyHipError) as e:
        _push_device(S, bases, bad, 1)
    assert e.value.code == _lib.ERR_INVALID
    S_small = api.SumOfSharedHashes(R, top=1, max_batch_reads=20, max_batch_bases=1000)
    with pytest.raises(_lib.SketchyHipError) as e:
        _push_device(S_small, bases, offsets, 1)
    assert e.value.code == _lib.ERR_CAPACITY
    assert S.reads == 0 and not S.table().any()
    # n_bases is what the caller vouches for from offsets[0] on: offsets reaching past it are refused, not followed
    d_b, d_o = api.DeviceBuffer.from_numpy(bases), api.DeviceBuffer.from_numpy(offsets)
    with pytest.raises(_lib.SketchyHipError) as e:
        S.push_device(d_b.ptr, d_o.ptr, 20, int(offsets[10]), None, None)
    assert e.value.code == _lib.ERR_INVALID
    d_b.free(); d_o.free()
    assert S.reads == 0 and not S.table().any()
    _push_device(S, bases, offsets, 1)  # the stream is still usable
    assert S.reads == 20


def test_submit_pipeline_matches_push(gpu):
    """skx_stream_submit / wait / drain (page-locked buffers, copy of batch i+1 overlapping batch i, processing one
    call behind submission): rows and table of the same batches pushed synchronously; uneven batches, a wait in the
    middle, reuse of the stream afterwards."""
    import ctypes as C
    from sketchy_amd import api
    ref, bases, offsets = workload(300, 400, 930, read_len=700, rng_seed=191)
    exp = orc.stream(16, 0, 400, ref["ref"], ref["col_len"], bases, offsets, top_k=2)
    n = len(offsets) - 1
    R = api.ReferenceSketch(ref["ref"], ref["col_len"])
    S = api.SumOfSharedHashes(R, top=2, max_batch_reads=256, max_batch_bases=256 * 700)
    hb, ho = api.HostBuffer(len(bases)), api.HostBuffer((n + 1) * 8)
    hi, hs = api.HostBuffer(n * 2 * 4), api.HostBuffer(n * 2 * 8)
    hb.view(np.uint8)[:] = bases
    ho.view(np.uint64)[:] = offsets
    hi.view(np.uint32)[:] = 0xFFFFFFFF
    at = lambda buf, off: C.c_void_p(buf.ptr.value + off)
    cuts = [0, 256, 300, 301, 557, 800, 930]
    tickets = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        tickets.append(S.submit(hb.ptr, at(ho, a * 8), b - a, at(hi, a * 8), at(hs, a * 16)))
        if len(tickets) == 3:
            S.wait(tickets[0])  # the first batch is complete although two more are queued behind it
            np.testing.assert_array_equal(hi.view(np.uint32).reshape(n, 2)[:256], exp["topk_idx"][:256])
    S.wait(tickets[3])
    np.testing.assert_array_equal(hs.view(np.uint64).reshape(n, 2)[:557], exp["topk_sum"][:557])
    S.drain()
    np.testing.assert_array_equal(hi.view(np.uint32).reshape(n, 2), exp["topk_idx"])
    np.testing.assert_array_equal(hs.view(np.uint64).reshape(n, 2), exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.reads == n
    # the synchronous entry point still works on the same stream after a drain
    S.reset()
    got = S.push(bases, offsets[:201])
    np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"][:200])
    for h in (hb, ho, hi, hs):
        h.free()


@pytest.mark.parametrize("top", [2, 16, 17, 40])
def test_topk_fast_and_generic_paths(gpu, top):
    """2..16 rows: pruned per-group kernel; 17..64: generic per-word kernel.  Duplicated genomes give ties that span
    rank groups of 512 genomes."""
    ref, bases, offsets = workload(1100, 160, 150, read_len=700, rng_seed=91)
    hashes = ref["ref"].copy()
    hashes[1000] = hashes[5]
    hashes[600] = hashes[5]
    check(hashes, bases, offsets, top=top, batches=2, want_shared=False, want_sketches=False)


@pytest.mark.parametrize("top", [2, 16])
@pytest.mark.parametrize("n_lead,spread", [(2, 0), (5, 1), (8, 0), (9, 1), (16, 0), (40, 0), (64, 0), (65, 0), (130, 1), (130, 0)])
def test_topk_with_a_given_number_of_leaders(gpu, top, n_lead, spread):
    """The pruned top-k ranking (rank_seg_topk_kernel) orders a rank group's candidates in one of four ways -- one candidate, up to eight
    (counted one by one), up to 64 (one per lane), more (read-by-read replay) -- and topk_merge_kernel merges the groups' lists in one of
    two.  Here n_lead variants of a SECOND ancestor sit among 1 100 variants of the first, packed into one rank group or spread over all
    three; the reads come from the second ancestor, so after the first batch exactly those n_lead genomes are ahead (with exact ties among
    them), and batches two and three rank them through the path their number selects (src/sketchy.rs:348, :389-400: sum desc, index asc)."""
    if n_lead < top:
        pytest.skip("fewer leaders than rows: the zero genomes tie for the rest and everything is a candidate")
    ref, _, _ = workload(1100, 160, 10, read_len=700, rng_seed=91)
    lead, bases, offsets = workload(n_lead, 160, 600, read_len=700, rng_seed=92)
    hashes = ref["ref"].copy()
    pos = (np.arange(n_lead) * (1100 // n_lead if spread else 1) + 7) % 1100
    hashes[pos] = lead["ref"]
    if n_lead >= 3:
        hashes[pos[-1]] = hashes[pos[0]]   # an exact tie between the lowest and the highest index
    check(hashes, bases, offsets, top=top, batches=3, want_shared=False, want_sketches=False)


def test_common_hashes_extreme_query_values(gpu):
    """Arbitrary query hashes reach the dictionary builder through `shared`: 0, the all-ones value (the hash set's
    empty marker), the table markers, heavy duplication across queries, and a query the reference never meets."""
    from sketchy_amd import api
    top = 0xFFFFFFFFFFFFFFFF
    ref = np.array([[0, 5, 9, top - 1, top],
                    [1, 5, 7, 8, top],
                    [0, 1, 2, 3, 4],
                    [10, 11, 12, top - 2, top - 1]], np.uint64)
    R = api.ReferenceSketch(ref)
    q = np.array([[0, 5, top - 1, top, 0],
                  [top, 0, 0, 0, 0],
                  [6, 13, 14, 15, 16],
                  [0, 1, 5, 9, top]], np.uint64)
    q_len = np.array([4, 1, 5, 5], np.uint32)
    got = R.common_hashes(q, q_len)
    exp = np.array([[orc.common_hashes(ref[g], q[i, :q_len[i]]) for g in range(4)] for i in range(4)], np.uint32)
    np.testing.assert_array_equal(got, exp)
    assert got[1].tolist() == [1, 1, 0, 0] and not got[2].any()
    # many queries sharing the same few hashes (the hash set sees heavy duplication)
    q2 = np.tile(np.array([[0, 5, 9, top - 1, top]], np.uint64), (300, 1))
    got2 = R.common_hashes(q2)
    assert (got2 == np.array([5, 2, 1, 1], np.uint32)).all()


def test_randomized_workloads(gpu):
    """40 random small workloads against the oracle: reference shape, ragged and duplicated columns, reads with
    ambiguity codes / lower case / white space / lengths around k, random `top` and batch cuts.
    SKX_TEST_SEED selects another stream of cases (sweeps beyond the committed one)."""
    import os
    from sketchy_amd import synth
    rng = np.random.default_rng(int(os.environ.get("SKX_TEST_SEED", "2024")))
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNRYKM-\n ", np.uint8)
    for case in range(40):
        n = int(rng.integers(1, 700))
        s = int(rng.choice([1, 7, 64, 200, 513]))
        k = int(rng.choice([16, 16, 16, 11, 21, 32]))
        seed = int(rng.choice([0, 0, 42, 7]))
        ref = synth.make_reference(n, s, k=k, hash_seed=seed, genome_len=max(3000, 40 * s), rng_seed=int(rng.integers(1, 10 ** 6)), device="numpy")
        hashes = ref["ref"].copy()
        col_len = np.full(n, s, np.uint32)
        if n > 3 and rng.random() < 0.5:
            hashes[rng.integers(0, n)] = hashes[rng.integers(0, n)]          # exact ties
        if rng.random() < 0.5:
            col_len = rng.integers(0, s + 1, size=n).astype(np.uint32)        # ragged, possibly empty columns
        n_reads = int(rng.integers(1, 200))
        bases, offsets = synth.make_reads(ref["genome"], n_reads, int(rng.choice([30, 150, 600])), err=0.03, rng_seed=int(rng.integers(1, 10 ** 6)),
                                          lognormal_sigma=0.7, min_len=0, max_len=4000)
        reads = [bytearray(bases[int(offsets[i]):int(offsets[i + 1])].tobytes()) for i in range(n_reads)]
        for r in reads:                                                       # dirty some reads
            if len(r) and rng.random() < 0.3:
                for _ in range(int(rng.integers(1, 6))):
                    r[int(rng.integers(0, len(r)))] = int(alphabet[rng.integers(0, len(alphabet))])
        if rng.random() < 0.3:
            reads[int(rng.integers(0, n_reads))] = bytearray(b"")
        bases, offsets = pack_reads([bytes(r) for r in reads])
        top = int(rng.integers(0, min(n, 20) + 1))
        got, exp, R, S = check(hashes, bases, offsets, top=top, k=k, seed=seed, col_len=col_len, batches=int(rng.integers(1, 4)),
                               want_shared=bool(rng.random() < 0.3), want_sketches=bool(rng.random() < 0.3))
        if top:  # the same batch through the device-resident entry point (offsets checked on the device)
            from sketchy_amd import api
            Sd = api.SumOfSharedHashes(R, top=top, max_batch_reads=n_reads, max_batch_bases=max(1, len(bases)))
            gd = _push_device(Sd, bases, offsets, top)
            np.testing.assert_array_equal(gd["topk_idx"], exp["topk_idx"], err_msg=f"case {case} device push idx")
            np.testing.assert_array_equal(gd["topk_sum"], exp["topk_sum"], err_msg=f"case {case} device push sum")
            np.testing.assert_array_equal(Sd.table(), exp["cum"], err_msg=f"case {case} device push table")


def test_truncation_comes_before_the_membership_filter(gpu):
    """Regression (found by test_randomized_workloads): a read hash that some genome holds but that ranks beyond the
    read's bottom-s must not count, even though every smaller non-member hash is dropped by the filter."""
    ref, bases, offsets = workload(4, 8, 6, read_len=700, genome_len=30000, rng_seed=33)
    read0 = bases[int(offsets[0]):int(offsets[1])].tobytes()
    H = orc.sketch(read0, 16, 0, 10 ** 6)  # all distinct hashes of the read, ascending
    assert len(H) > 40
    top = np.uint64(0xFFFFFFFFFFFFFFF0)
    hashes = np.array([[H[0], H[7]], [H[3], H[9]], [H[1], top], [H[2], H[30]]], np.uint64)
    got, exp, _, _ = check(hashes, bases, offsets, top=2, want_sketches=False)   # production sketch path, s = 2
    assert exp["shared"][0].tolist() == [1, 0, 1, 0]
    check(hashes, bases, offsets, top=1, want_shared=False, want_sketches=False)


@pytest.mark.parametrize("first_len", [40, 300, 1])
def test_read_sketch_size_is_the_first_sketchs_length(gpu, first_len):
    """The reference sketches every read with s := number of hashes of the collection's FIRST sketch (src/sketchy.rs:82,
    :520-527); the other sketches may be longer (first_len = 40, 1: the read sketch is truncated to fewer hashes than a
    column holds -- a dense reference, so hundreds of a read's hashes are in range and the truncation really bites) or
    shorter (first_len = 300 = the stride: most columns are shorter than the read sketch).  skx_ref_create takes s and the
    column stride separately; rows, per-read counts, sketches and table against the oracle, debug and production paths."""
    ref, bases, offsets = workload(90, 300, 120, read_len=900, genome_len=3000, rng_seed=4100 + first_len)
    hashes = ref["ref"]
    rng = np.random.default_rng(first_len)
    col_len = rng.integers(100, 301, size=len(hashes)).astype(np.uint32)
    col_len[0] = first_len
    col_len[5] = 300
    got, exp, R, S = check(hashes, bases, offsets, top=3, col_len=col_len, s=first_len)
    assert R.s == first_len and R.stride == 300
    assert exp["sketch_len"].max() == first_len                       # every read sketch is cut to |sketch 0|
    full = orc.stream(16, 0, 300, hashes, col_len, bases, offsets, top_k=1, want_shared=True)
    if first_len < 300:
        assert (full["shared"].astype(np.int64) - exp["shared"]).max() > 0  # ... and the cut changes the counts
    check(hashes, bases, offsets, top=1, col_len=col_len, s=first_len, want_shared=False, want_sketches=False, batches=3)


def test_randomized_families_leader_switching(gpu):
    """Larger random cases for the pruned ranking: 2-4 clone families (different ancestors) shuffled over tiles and rank
    groups, reads arriving in family blocks of random length (the leader changes inside segments, across segments and
    across 1024-read chunks), top in {1, 2, 5}, random batch cuts.  SKX_TEST_SEED selects another stream of cases."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SKX_TEST_SEED", "77")))
    for case in range(3):
        n_fam = int(rng.integers(2, 5))
        s = int(rng.choice([64, 200]))
        fams = [workload(int(rng.integers(150, 700)), s, 900, read_len=int(rng.choice([300, 800])), rng_seed=int(rng.integers(1, 10 ** 6)))
                for _ in range(n_fam)]
        hashes = np.concatenate([f[0]["ref"] for f in fams])
        n = len(hashes)
        for _ in range(3):  # exact ties, possibly across families' positions
            hashes[rng.integers(0, n)] = hashes[rng.integers(0, n)]
        hashes = np.ascontiguousarray(hashes[rng.permutation(n)])
        pools = [[f[1][int(f[2][i]):int(f[2][i + 1])].tobytes() for i in range(900)] for f in fams]
        used = [0] * n_fam
        reads = []
        while len(reads) < 2300:
            f = int(rng.integers(0, n_fam))
            blk = int(rng.choice([1, 3, 20, 64, 150, 700]))
            take = pools[f][used[f]:used[f] + blk]
            used[f] += len(take)
            reads += take
            if all(u >= 900 for u in used):
                break
        bases, offsets = pack_reads(reads)
        for top in (1, int(rng.choice([2, 5]))):
            check(hashes, bases, offsets, top=top, batches=int(rng.integers(1, 4)), want_shared=False, want_sketches=False)


def test_sketch_normalisation_stress(gpu):
    """The wave sketcher classifies four bytes per lane through an LDS table and stores whole words while nothing has
    been removed; a removed byte (whitespace) switches the rest of the chunk to byte-wise compaction, and the window
    state of every lane's run is rebuilt from the code bytes.  Random reads of every small length (partial words at the
    end, runs starting at every alignment), whitespace / N / lower case / U / IUPAC at random places, lengths straddling
    the 256-byte groups and the 2048-byte chunks, whitespace right at a chunk border -- sketches (full, k = 16 and the
    generic k path) and production rows against the oracle."""
    from sketchy_amd import api
    rng = np.random.default_rng(2025)
    ref, _, _ = workload(24, 300, 1, read_len=300, genome_len=60000, rng_seed=401)
    g = ref["genome"].tobytes()
    reads = []
    for n in list(range(0, 70)) + [255, 256, 257, 258, 259, 511, 512, 513, 1023, 1025, 2047, 2048, 2049, 2062, 2063]:
        a = int(rng.integers(0, len(g) - n - 1))
        reads.append(bytearray(g[a:a + n]))
    for n in (300, 700, 2100, 4096 + 15, 4097, 6000, 9000):
        for _ in range(3):
            a = int(rng.integers(0, len(g) - n - 1))
            reads.append(bytearray(g[a:a + n]))
    dirt = [b"\n", b" ", b"\t", b"\r", b"N", b"n", b"R", b"-", b"u", b"U", b"a", b"c", b"g", b"t"]
    out = []
    for i, r in enumerate(reads):
        r = bytes(r)
        if i % 3 and len(r):
            pieces, pos = [], 0
            for cut in sorted(set(int(x) for x in rng.integers(0, len(r) + 1, size=int(rng.integers(1, 6))))):
                pieces.append(r[pos:cut])
                pieces.append(dirt[int(rng.integers(0, len(dirt)))] * int(rng.integers(1, 4)))
                pos = cut
            pieces.append(r[pos:])
            r = b"".join(pieces)
        out.append(r)
    # whitespace exactly around the 2048-byte chunk border and the 256-byte group borders of a long read
    base = g[100:100 + 5000]
    out.append(base[:2047] + b"\n" + base[2047:])
    out.append(base[:2048] + b"\n" + base[2048:])
    out.append(base[:255] + b" " + base[255:256] + b"\t" + base[256:4000])
    out.append(b"\n" * 300 + base[:600] + b"\n" * 2100 + base[600:900])
    bases, offsets = pack_reads(out)
    for k, seed in ((16, 0), (16, 42), (21, 0), (11, 7)):
        sk, sl = api.sketch_reads(bases, offsets, k=k, seed=seed, s=200)
        for r in range(len(out)):
            e = orc.sketch(out[r], k, seed, 200)
            assert sl[r] == len(e), (k, seed, r, len(out[r]))
            np.testing.assert_array_equal(sk[r, :len(e)], e, err_msg=f"k={k} seed={seed} read {r}")
    check(ref, bases, offsets, top=2)                                              # debug outputs: full sketches
    check(ref, bases, offsets, top=1, want_sketches=False, want_shared=False)      # production (in-range) sketch path
