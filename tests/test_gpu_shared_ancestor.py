"""Species that SHARE hashes, with more than 1 024 genomes each: two SNP clone trees grown from the same ancestor (most of a sketch is the
ancestor's hashes: held by nearly every genome of BOTH species) beside an unrelated third tree, resident in one skx_ref.

What only this arrangement reaches: a hash that is dense in one species and present in another cannot live in one
species' segment of the static dense dictionary -- skx_ref_create gives it a genome list whatever its length (`n_forced_rare`), the static
dictionary keeps the hashes that are dense in exactly one species, and the rare-row walks (gains, candidates, patterns) carry lists of
thousands of genomes across species borders.  Every species must still score as its own `sketchy predict` run over the same mixed
stream (src/sketchy.rs:81-82, :337-349, :425-438): every row of every batch, the tables, and the per-read shared counts."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECIES, S_, B, NB = [2400, 1800, 1300], 1000, 8192, 14


def _generate(d):
    # (in a child: torch's bundled HIP runtime and the library's do not share a process, as in the other GPU tests)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sketchy_amd import synth\n"
        "a = synth.make_reference(%d, %d, rng_seed=3, device='cuda', mode='snp', n_lineages=12)\n"
        "b = synth.make_reference(%d, %d, rng_seed=3, device='cuda', mode='snp', n_lineages=9, div_lineage=0.013, div_strain=0.0007)\n"
        "c = synth.make_reference(%d, %d, rng_seed=9, device='cuda', mode='snp', n_lineages=8)\n"
        "assert np.array_equal(a['genome'], b['genome'])\n"
        "srcs = [torch.from_numpy(x).to('cuda') for x in (a['truth_genome'], b['truth_genome'], c['truth_genome'])]\n"
        "share = [0.5, 0.4, 0.1]\n"
        "parts, lens = [], []\n"
        "for i in range(%d):\n"
        "    n = [int(%d * f) for f in share]; n[0] += %d - sum(n)\n"
        "    pr = [synth.make_reads_torch(srcs[j], n[j], 1200, rng_seed=(700 + i) * 7 + j, lognormal_sigma=0.6, device='cuda') for j in range(3)]\n"
        "    bb, oo = synth.mix_reads_torch(pr, rng_seed=700 + i)\n"
        "    parts.append(bb.cpu().numpy()); lens.append(np.diff(oo.cpu().numpy().astype(np.int64)))\n"
        "bases = np.concatenate(parts); offsets = np.zeros(1 + sum(len(x) for x in lens), np.uint64)\n"
        "offsets[1:] = np.cumsum(np.concatenate(lens)).astype(np.uint64)\n"
        "for i, r in enumerate((a, b, c)): np.save(%r + '/ref%%d.npy' %% i, r['ref'])\n"
        "np.save(%r + '/bases.npy', bases); np.save(%r + '/offsets.npy', offsets)\n"
        "np.save(%r + '/truth.npy', np.array([a['truth_index'], b['truth_index'], c['truth_index']]))\n"
    ) % (ROOT, SPECIES[0], S_, SPECIES[1], S_, SPECIES[2], S_, NB, B, B, d, d, d, d)
    subprocess.check_call([sys.executable, "-c", code])


@pytest.fixture(scope="module")
def kin(gpu):
    d = tempfile.mkdtemp(prefix="skx_kin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        _generate(d)
        out = dict(refs=[np.load(f"{d}/ref{i}.npy") for i in range(3)], bases=np.load(d + "/bases.npy"), offsets=np.load(d + "/offsets.npy"),
                   truth=np.load(d + "/truth.npy"))
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    from sketchy_amd import api
    out["R"] = api.ReferenceSketch(out["refs"])
    yield out
    out["R"].close()


def test_the_two_trees_share_their_ancestors_hashes(kin):
    a, b, c = kin["refs"]
    ha, hb = np.unique(a), np.unique(b)
    common = np.intersect1d(ha, hb)
    held_a = np.isin(a, common).sum(axis=1)
    print("distinct hashes:", len(ha), len(hb), "common:", len(common), "per genome of A:", held_a.min(), held_a.max(),
          "rare index:", kin["R"].rare_index, "static:", kin["R"].static_dense, "patterns:", kin["R"].patterns)
    assert held_a.min() > S_ // 2                       # most of every sketch of A is also in B
    assert len(np.intersect1d(ha, np.unique(c))) < 50   # the third collection is unrelated (a few chance 16-mers)
    # hashes dense (more than 1 024 holders, the default `rare_hash_genomes`) in one species and present in the other: these get lists
    ua, ca = np.unique(a, return_counts=True)
    ub, cb = np.unique(b, return_counts=True)
    both = np.intersect1d(ua[ca > 1024], ub)
    assert len(both) > 100
    assert kin["R"].rare_index["postings"] > int(ca[np.isin(ua, both)].sum())   # ... and their lists are in the index, whatever their length


@pytest.mark.parametrize("top,reuse", [(1, 0), (4, 0), (9, 0), (1, 1)])
def test_rows_and_tables_through_shared_passes(kin, top, reuse):
    from oracle import oracle as orc
    from sketchy_amd import api
    R, refs, bases, offsets = kin["R"], kin["refs"], kin["bases"], kin["offsets"]
    api.set_option("reuse_membership", reuse)
    try:
        S = api.SumOfSharedHashes(R, top=top, max_batch_reads=B, max_batch_bases=int(np.max(offsets[B::B] - offsets[:-B:B])))
    finally:
        api.set_option("reuse_membership", 0)
    d_b = api.DeviceBuffer.from_numpy(bases)
    bufs = []
    try:
        for i in range(NB):
            a = i * B
            d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:a + B + 1]))
            d_i, d_s = api.DeviceBuffer(B * 3 * top * 4), api.DeviceBuffer(B * 3 * top * 8)
            bufs.append((d_o, d_i, d_s))
            S.enqueue_device(d_b.ptr, d_o.ptr, B, int(offsets[a + B] - offsets[a]), d_i.ptr, d_s.ptr)
        S.sync()
        st = S.stats()
        got = [(t[1].to_numpy(np.uint32, (B, 3, top)), t[2].to_numpy(np.uint64, (B, 3, top))) for t in bufs]
        table = S.table()
    finally:
        d_b.free()
        for t in bufs:
            for x in t:
                x.free()
        S.close()
    print("stats:", st)
    assert st["passes_shared"] >= 2, st
    assert st["batches_compact"] >= 2, st   # every species has a leader after the first pass: the candidates' compact problems were used
    col = 0
    for sp, ref in enumerate(refs):
        cum = None
        for i in range(NB):
            a = i * B
            e = orc.stream_fast(16, 0, S_, ref, None, bases, offsets[a:a + B + 1], top_k=top, cum=cum, rows=True)
            cum = e["cum"]
            np.testing.assert_array_equal(got[i][1][:, sp], e["topk_sum"], err_msg=f"species {sp}, batch {i}: sums")
            np.testing.assert_array_equal(got[i][0][:, sp], e["topk_idx"], err_msg=f"species {sp}, batch {i}: genomes")
        np.testing.assert_array_equal(table[col:col + SPECIES[sp]], cum, err_msg=f"table of species {sp}")
        assert int(np.argmax(cum)) == int(kin["truth"][sp])
        col += SPECIES[sp]


def test_shared_counts_of_a_synchronous_push(kin):
    """the per-read x per-genome matrix (src/sketchy.rs:425-438) through the full ranking: the forced lists go into the group-major matrix"""
    from oracle import oracle as orc
    from sketchy_amd import api
    R, refs, bases, offsets = kin["R"], kin["refs"], kin["bases"], kin["offsets"]
    n = 256
    S = api.SumOfSharedHashes(R, top=2, max_batch_reads=n, max_batch_bases=int(offsets[n]))
    got = S.push(bases, offsets[:n + 1], want_shared=True)
    S.close()
    col = 0
    for sp, ref in enumerate(refs):
        e = orc.stream(16, 0, S_, ref, np.full(SPECIES[sp], S_, np.uint32), bases[:int(offsets[n])], offsets[:n + 1], top_k=2, want_shared=True)
        np.testing.assert_array_equal(got["shared"][:, col:col + SPECIES[sp]], e["shared"], err_msg=f"species {sp}")
        np.testing.assert_array_equal(got["topk_sum"][:, sp], e["topk_sum"])
        np.testing.assert_array_equal(got["topk_idx"][:, sp], e["topk_idx"])
        col += SPECIES[sp]
