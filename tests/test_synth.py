"""The synthetic generator (product side) hashes exactly like the path under test and produces
workloads with real sharing."""
import numpy as np

from helpers import workload
from oracle import oracle as orc
from sketchy_amd import synth


def test_vectorised_hashing_matches_the_oracle():
    rng = np.random.default_rng(5)
    g = synth.random_genome(3000, rng)
    for k, seed in ((16, 0), (16, 42), (21, 7), (11, 3), (32, 1), (1, 0)):
        a = synth.canonical_kmer_hashes(g, k, seed)
        b, _ = orc.kmer_hashes(g.tobytes(), k, seed)
        np.testing.assert_array_equal(a, b)


def test_reference_is_valid_and_shared():
    ref, bases, offsets = workload(60, 200, 50, read_len=1500, rng_seed=3)
    h = ref["ref"]
    assert h.shape == (60, 200) and h.dtype == np.uint64
    assert (h[:, 1:] > h[:, :-1]).all()
    exp = orc.stream(16, 0, 200, h, ref["col_len"], bases, offsets, top_k=1, want_shared=True)
    assert exp["shared"].sum() > 0
    assert len(offsets) == 51 and offsets[-1] == len(bases) == 50 * 1500


def test_deterministic():
    a = workload(20, 64, 10, read_len=300, genome_len=30000, rng_seed=9)
    b = workload(20, 64, 10, read_len=300, genome_len=30000, rng_seed=9)
    np.testing.assert_array_equal(a[0]["ref"], b[0]["ref"])
    np.testing.assert_array_equal(a[1], b[1])


def test_mixed_read_lengths():
    ref, _, _ = workload(4, 64, 1, genome_len=60000, rng_seed=13)
    bases, offsets = synth.make_reads(ref["genome"], 200, 1500, lognormal_sigma=0.8, min_len=200, max_len=50000)
    lens = np.diff(offsets.astype(np.int64))
    assert lens.min() >= 200 and lens.max() <= 50000 and len(set(lens)) > 50
    assert set(np.unique(bases)) <= set(b"ACGT")
