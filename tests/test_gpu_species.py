"""Several reference collections ("species") resident in one skx_ref and scored in ONE pass per batch
(BASELINE.json configs[4]): every species keeps its own table and its own (sum desc, index asc) rows.  The oracle is
what the reference does for that deployment -- one `sketchy predict` run per species over the same reads
(src/sketchy.rs:81-82, :317-356) -- so each species is compared with its own orc.stream."""
import numpy as np
import pytest

from helpers import workload_species
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _oracle(refs, bases, offsets, top, k=16, seed=0, col_lens=None):
    out = []
    for i, r in enumerate(refs):
        cl = r["col_len"] if col_lens is None else col_lens[i]
        out.append(orc.stream(k, seed, r["ref"].shape[1], r["ref"], cl, bases, offsets, top_k=max(top, 1), want_shared=True))
    return out


def _check(refs, bases, offsets, top, batches=1, col_lens=None, want_shared=True):
    from sketchy_amd import api
    exp = _oracle(refs, bases, offsets, top, col_lens=col_lens)
    R = api.ReferenceSketch([r["ref"] for r in refs], [r["col_len"] for r in refs] if col_lens is None else col_lens)
    assert R.n_species == len(refs) and R.n_genomes == sum(r["ref"].shape[0] for r in refs)
    n = len(offsets) - 1
    S = api.SumOfSharedHashes(R, top=top, max_batch_reads=n, max_batch_bases=max(1, len(bases)))
    cuts = np.linspace(0, n, batches + 1).astype(int)
    parts = [S.push(bases, offsets[a:b + 1], want_shared=want_shared) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    if top:
        ti = np.concatenate([p["topk_idx"] for p in parts]).reshape(n, len(refs), top)
        ts = np.concatenate([p["topk_sum"] for p in parts]).reshape(n, len(refs), top)
        for i, e in enumerate(exp):
            np.testing.assert_array_equal(ts[:, i], e["topk_sum"], err_msg=f"species {i} sums")
            np.testing.assert_array_equal(ti[:, i], e["topk_idx"], err_msg=f"species {i} rows")
    if want_shared:
        sh = np.concatenate([p["shared"] for p in parts])
        np.testing.assert_array_equal(sh, np.concatenate([e["shared"] for e in exp], axis=1))
    np.testing.assert_array_equal(S.table(), np.concatenate([e["cum"] for e in exp]))
    return R, S, exp


@pytest.mark.parametrize("top", [1, 3, 17])
def test_three_species_rows_shared_and_table(gpu, top):
    """Species sizes straddling tile (256) and rank-group (512) boundaries; pruned top-1, pruned top-k and the generic
    path; reads from all three ancestors in one stream."""
    refs, bases, offsets = workload_species([130, 700, 513], 300, 240, read_len=800, genome_len=90000, rng_seed=201)
    R, S, exp = _check(refs, bases, offsets, top, batches=3)
    assert all(e["shared"].max() > 0 for e in exp)
    idx, sm = S.rank(5)
    for i, e in enumerate(exp):
        order = orc.stable_rank(e["cum"])[:5]
        np.testing.assert_array_equal(idx[i], order)
        np.testing.assert_array_equal(sm[i], e["cum"][order])


def test_species_with_ties_ragged_columns_and_one_genome(gpu):
    rng = np.random.default_rng(5)
    refs, bases, offsets = workload_species([1, 60, 515, 2], 128, 90, read_len=500, genome_len=40000, rng_seed=211)
    refs[2]["ref"][300] = refs[2]["ref"][7]       # exact ties inside a species: reference order decides
    refs[2]["ref"][514] = refs[2]["ref"][7]
    refs[1]["ref"][:] = refs[2]["ref"][:60]       # the same genomes in two species: each ranks on its own
    col_lens = [r["col_len"].copy() for r in refs]
    col_lens[2] = rng.integers(0, 129, size=515).astype(np.uint32)
    col_lens[2][[7, 300, 514]] = 128
    col_lens[2][9] = 0
    _check(refs, bases, offsets, top=1, col_lens=col_lens)
    from sketchy_amd import _lib, api
    R = api.ReferenceSketch([r["ref"] for r in refs], col_lens)
    with pytest.raises(_lib.SketchyHipError) as e:   # top_k is bounded by the smallest species (src/sketchy.rs:391 per run)
        api.SumOfSharedHashes(R, top=2)
    assert e.value.code == _lib.ERR_INVALID


def test_species_table_add_reset_and_common_hashes(gpu):
    from sketchy_amd import api
    refs, bases, offsets = workload_species([300, 90, 1025], 200, 150, read_len=600, genome_len=60000, rng_seed=221)
    R, S, exp = _check(refs, bases, offsets, top=2, want_shared=False)
    total = S.table()
    half = (len(offsets) - 1) // 2
    S.reset()
    a = S.push(bases, offsets[:half + 1])
    ta = S.table()
    S.reset()
    S.table_add(ta)
    b = S.push(bases, offsets[half:])
    np.testing.assert_array_equal(S.table(), total)
    for i, e in enumerate(exp):
        np.testing.assert_array_equal(np.concatenate([a["topk_idx"][:, i], b["topk_idx"][:, i]]), e["topk_idx"])
        np.testing.assert_array_equal(np.concatenate([a["topk_sum"][:, i], b["topk_sum"][:, i]]), e["topk_sum"])
    # all-pairs operator over the concatenated genomes: a sketch shares everything with itself (docs/index.md:145-149)
    q = np.stack([refs[0]["ref"][5], refs[1]["ref"][0], refs[2]["ref"][1024]])
    common = R.common_hashes(q)
    assert common.shape == (3, 300 + 90 + 1025)
    assert common[0, 5] == 200 and common[1, 300] == 200 and common[2, 300 + 90 + 1024] == 200
    for qi in range(3):
        col = 0
        for r in refs:
            for g in (0, r["ref"].shape[0] - 1):
                assert common[qi, col + g] == orc.common_hashes(r["ref"][g], q[qi])
            col += r["ref"].shape[0]


def test_c4_shrunk_mixed_read_lengths(gpu):
    """BASELINE configs[4] at 1/20 scale: five species (2000, 2000, 1500, 1250, 750 genomes, s=1000), a log-normal
    200..50 000-base stream (long reads go through the chunked sketcher), every read scored against all five."""
    refs, bases, offsets = workload_species([2000, 2000, 1500, 1250, 750], 1000, 400, read_len=1500, rng_seed=231,
                                            lognormal_sigma=1.0)
    lens = np.diff(offsets.astype(np.int64))
    assert lens.max() > 2063 and lens.min() < 1000
    _check(refs, bases, offsets, top=1, batches=2, want_shared=False)
    _check(refs, bases, offsets, top=5, want_shared=True)
