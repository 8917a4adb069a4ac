"""Pins the CPU oracle: public MurmurHash3 known answers, the derived vectors of SURVEY.md 8(c)
(tests/golden/kats.json), the frozen stream fixture, and C-vs-Python cross checks.

The reference holds no tests or vectors for this path and cannot be run here (Rust, no
toolchain/crates): parity with upstream is UNPINNED; these anchors are what exists.
"""
import json
import os
import struct

import numpy as np
import pytest

from helpers import pack_reads, unpack_reads, workload
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLD, "kats.json")))


@pytest.mark.parametrize("fn", [orc.murmur3_x64_128, orc.py_murmur3_x64_128])
def test_murmur3_known_answers(fn):
    for v in KATS["murmur3_x64_128"]:
        h1, h2 = fn(v["key"].encode(), v["seed"])
        assert f"{h1:016x}" == v["h1"]
        if "h2" in v:
            assert f"{h2:016x}" == v["h2"]
    h1, h2 = fn(b"foo", 0)
    s1, s2 = struct.unpack("<qq", struct.pack("<QQ", h1, h2))
    assert [s1, s2] == KATS["mmh3_hash64_foo_signed"]


@pytest.mark.parametrize("fn", [orc.murmur3_x64_128, orc.py_murmur3_x64_128])
def test_smhasher_verification_value(fn):
    """SMHasher VerificationTest: keys {0..i-1}, seed 256-i, hash the 256 results with seed 0."""
    buf = b""
    for i in range(256):
        buf += struct.pack("<QQ", *fn(bytes(range(i)), 256 - i))
    h1, _ = fn(buf, 0)
    assert f"{h1 & 0xFFFFFFFF:08X}" == KATS["smhasher_verification_x64_128"]


def test_murmur3_all_tail_lengths_c_vs_python():
    rng = np.random.default_rng(0)
    for n in list(range(0, 50)) + [63, 64, 65, 255]:
        key = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        for seed in (0, 42, 0xDEADBEEFCAFEF00D):
            assert orc.murmur3_x64_128(key, seed) == orc.py_murmur3_x64_128(key, seed)


def test_canonical_kmers_vector():
    v = KATS["canonical_kmers"]
    got = orc.py_canonical_kmers(v["read"].encode(), v["k"])
    assert [(p, km.decode(), int(rc)) for p, km, rc in got] == [(p, km, rc) for p, km, rc, _ in v["kmers"]]
    hashes, is_rc = orc.kmer_hashes(v["read"].encode(), v["k"], v["seed"])
    assert [f"{h:016x}" for h in hashes] == [h for _, _, _, h in v["kmers"]]
    assert list(is_rc) == [rc for _, _, rc, _ in v["kmers"]]
    for impl in ("heap", "sort"):
        assert [f"{h:016x}" for h in orc.sketch(v["read"].encode(), v["k"], v["seed"], 4, impl)] == v["bottom4"]
    assert [f"{h:016x}" for h in orc.py_sketch(v["read"].encode(), v["k"], v["seed"], 4)] == v["bottom4"]


def test_normalise_vector_and_palindrome():
    v = KATS["normalise"]
    assert orc.normalize(v["read"].encode()).decode() == v["normalised"]
    assert orc.py_normalize(v["read"].encode()).decode() == v["normalised"]
    assert [p for p, _, _ in orc.py_canonical_kmers(v["read"].encode(), 16)] == v["valid_starts"]
    for impl in ("heap", "sort"):
        assert [f"{h:016x}" for h in orc.sketch(v["read"].encode(), 16, 0, 100, impl)] == v["sketch"]
    p = KATS["palindrome"]
    (_, km, rc), = orc.py_canonical_kmers(p["kmer"].encode(), 16)
    assert km.decode() == p["kmer"] and int(rc) == p["is_rc"]
    _, is_rc = orc.kmer_hashes(p["kmer"].encode(), 16, 0)
    assert list(is_rc) == [1]


def test_normalise_every_byte():
    allb = bytes(range(256))
    assert orc.normalize(allb) == orc.py_normalize(allb)
    out = orc.normalize(allb)
    assert len(out) == 256 - 4 and set(out) <= set(b"ACGTN-")


def test_rank_vector_and_stability():
    r = KATS["rank"]
    assert list(orc.stable_rank(np.array(r["sums"], np.uint64))) == r["order"]
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 64, 1000):
        sums = rng.integers(0, 5, n).astype(np.uint64)
        exp = sorted(range(n), key=lambda i: -int(sums[i]))
        assert list(orc.stable_rank(sums)) == exp


def test_heap_sketcher_equals_net_semantics_and_python():
    """finch's heap+map push sequence (faithful) == sort/distinct/truncate == pure Python."""
    ref, bases, offsets = workload(2, 64, 30, read_len=700, genome_len=30000, rng_seed=17)
    reads = unpack_reads(bases, offsets) + [b"A" * 300, b"ACGT" * 100, b"", b"ACG", b"N" * 50]
    for rd in reads:
        for s in (1, 5, 100, 685, 5000):
            a = orc.sketch(rd, 16, 0, s, "heap")
            b = orc.sketch(rd, 16, 0, s, "sort")
            np.testing.assert_array_equal(a, b)
            if s in (5, 5000):
                assert [int(x) for x in a] == orc.py_sketch(rd, 16, 0, s)
            assert (np.diff(a.astype(object)) > 0).all() if len(a) > 1 else True


def test_sketch_is_order_independent_and_strand_symmetric():
    ref, bases, offsets = workload(2, 64, 4, read_len=500, genome_len=30000, rng_seed=19, err=0)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for rd in unpack_reads(bases, offsets):
        a = orc.sketch(rd, 16, 0, 200)
        b = orc.sketch(rd.translate(comp)[::-1], 16, 0, 200)
        np.testing.assert_array_equal(a, b)  # canonical k-mers: reverse complement sketches identically


def test_common_hashes_properties():
    rng = np.random.default_rng(5)
    a = np.unique(rng.integers(0, 1000, 300)).astype(np.uint64)
    b = np.unique(rng.integers(0, 1000, 200)).astype(np.uint64)
    assert orc.common_hashes(a, a) == len(a)  # docs/index.md:145-149: self vs self = sketch size
    assert orc.common_hashes(a, b) == orc.common_hashes(b, a) == len(np.intersect1d(a, b)) == orc.py_common(a, b)
    assert orc.common_hashes(a, np.zeros(0, np.uint64)) == 0
    assert orc.common_hashes(a, b) <= min(len(a), len(b))


def test_stream_c_vs_python_small():
    ref, bases, offsets = workload(12, 48, 15, read_len=400, genome_len=30000, rng_seed=23)
    hashes = ref["ref"].copy()
    hashes[7] = hashes[1]
    col_len = ref["col_len"].copy()
    col_len[4] = 9
    exp = orc.stream(16, 0, 48, hashes, col_len, bases, offsets, top_k=3, want_shared=True)
    cols = [[int(x) for x in hashes[g, :col_len[g]]] for g in range(12)]
    rows, shared, cum = orc.py_stream(16, 0, 48, cols, unpack_reads(bases, offsets), top_k=3)
    np.testing.assert_array_equal(exp["shared"], np.array(shared, np.uint32))
    np.testing.assert_array_equal(exp["cum"], np.array(cum, np.uint64))
    for r, row in enumerate(rows):
        assert [i for i, _ in row] == list(exp["topk_idx"][r])
        assert [s for _, s in row] == list(exp["topk_sum"][r])
    # cumulative table == sum of per-read vectors; prefix property of the top rows
    np.testing.assert_array_equal(exp["shared"].sum(axis=0), exp["cum"])


def test_frozen_stream_fixture():
    z = np.load(os.path.join(GOLD, "stream_small.npz"))
    exp = orc.stream(int(z["k"]), int(z["seed"]), int(z["s"]), z["hashes"], z["col_len"], z["bases"], z["offsets"],
                     top_k=4, want_shared=True, want_sketches=True)
    for key in ("cum", "topk_idx", "topk_sum", "shared", "sketches", "sketch_len"):
        np.testing.assert_array_equal(exp[key], z[key], err_msg=key)


def test_stream_rejects_top_above_n():
    ref, bases, offsets = workload(3, 16, 2, read_len=200, genome_len=30000, rng_seed=29)
    with pytest.raises(ValueError):
        orc.stream(16, 0, 16, ref["ref"], ref["col_len"], bases, offsets, top_k=4)
