"""SURVEY.md 8(d)'s generator (items 1-3): SNP clone tree whose sketches hold REAL canonical k-mer hashes, reads from one
truth strain.  CPU checks: every row is the oracle's sketch of that strain's genome, the tree's sharing structure, and the
regime the survey asks for -- the truth strain's lineage, and within it the strain, accumulates fastest."""
import numpy as np
import pytest

from helpers import workload_snp
from oracle import oracle as orc


@pytest.fixture(scope="module")
def small():
    return workload_snp(500, 1000, 600, rng_seed=11)


def test_truth_row_is_the_oracle_sketch_of_the_truth_genome(small):
    ref, _, _ = small
    sk = orc.sketch(ref["truth_genome"].tobytes(), 16, 0, 1000)
    np.testing.assert_array_equal(sk, ref["ref"][ref["truth_index"]])
    r = ref["ref"]
    assert (r[:, 1:] > r[:, :-1]).all()                   # columns strictly ascending (distinct hashes)
    anc = orc.sketch(ref["genome"].tobytes(), 16, 0, 1000)
    share = np.array([len(np.intersect1d(anc, row)) for row in r[:50]])
    assert 700 < share.mean() < 900                       # 1 % lineage SNPs kill ~15 % of the 16-mers


def test_torch_hashes_equal_the_oracles():
    import subprocess
    import sys
    from helpers import ROOT
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sketchy_amd import synth\n"
        "from oracle import oracle as orc\n"
        "rng = np.random.default_rng(3)\n"
        "for k, seed in ((16, 0), (16, 42), (21, 7), (31, 1), (32, 0), (11, 9)):\n"
        "    g = np.frombuffer(b'ACGT', np.uint8)[rng.integers(0, 4, 600)]\n"
        "    g[100:100 + k] = np.frombuffer(b'ACGT' * 8, np.uint8)[:k]   # a palindrome for even k\n"
        "    want, _ = orc.kmer_hashes(g.tobytes(), k, seed)\n"
        "    codes = torch.from_numpy(np.searchsorted(np.frombuffer(b'ACGT', np.uint8), g).astype(np.uint8))\n"
        "    got = synth._t_window_hashes(codes, torch.arange(600 - k + 1), k, seed).numpy().view(np.uint64)\n"
        "    assert np.array_equal(got, want), (k, seed)\n"
        "    assert np.array_equal(synth.canonical_kmer_hashes(g, k, seed), want)\n"
        "print('ok')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_truth_strain_accumulates_fastest(small):
    """item 3's regime: after a few hundred reads the truth strain's lineage leads every other lineage, and the truth
    strain is the best (or ties the best) of its lineage -- and the oracle's top row says so."""
    ref, bases, offsets = small
    out = orc.stream_fast(16, 0, 1000, ref["ref"], ref["col_len"], bases, offsets, top_k=1)
    lin, ti = ref["lineage"], ref["truth_index"]
    cum = out["cum"].astype(np.int64)
    mine = lin == lin[ti]
    assert cum[mine].min() > cum[~mine].max()
    assert cum[ti] == cum.max()
    assert lin[out["topk_idx"][-1, 0]] == lin[ti]
    # strain-specific hits exist: the truth strain is strictly ahead of some of its lineage mates
    assert (cum[mine] < cum[ti]).any()
    st = out["stats"]
    assert 0.5 < st["pairs"] / 600 < 40


def test_ancestor_reads_and_other_k(small):
    ref, bases, offsets = workload_snp(60, 200, 40, read_len=500, k=21, seed=42, genome_len=60000, rng_seed=5, source="ancestor")
    exp = orc.stream(21, 42, 200, ref["ref"], ref["col_len"], bases, offsets, top_k=2)
    got = orc.stream_fast(21, 42, 200, ref["ref"], ref["col_len"], bases, offsets, top_k=2, n_threads=3)
    np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"])
    np.testing.assert_array_equal(got["cum"], exp["cum"])
    assert exp["cum"].max() > 0
