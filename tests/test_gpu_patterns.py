"""Long genome lists as (pattern, exceptions) -- skx_kernels.hip, "long lists as (pattern, exceptions)", DESIGN.md 2.10.

A SURVEY 8(d) reference in small (lineages of 200 strains, real 16-mer hashes): most of its long lists are "the lineage's strains
minus the odd one", so skx_ref_create stores them as a shared pattern plus a few exceptions, and a pass adds such rows up per pattern
(gain) and maps them to the pattern's row of the compact matrices (candidates).  Every row of every batch and the table must still be
the oracle's (src/sketchy.rs:337-349, :425-438) -- with the sample's own lineage as the candidates (its strains ARE the exceptions of
its lineage-level hashes), for top-1 and top-3, through shared passes and through synchronous pushes."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N, S_, B, NB = 6000, 2000, 16384, 14
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generate(d):
    # (in a child: torch's bundled HIP runtime and the library's do not share a process, as in the other GPU tests)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sketchy_amd import synth\n"
        "ref = synth.make_reference(%d, %d, rng_seed=3, device='cuda', mode='snp', n_lineages=30)\n"
        "src = torch.from_numpy(ref['truth_genome']).to('cuda')\n"
        "parts, lens = [], []\n"
        "for i in range(%d):\n"
        "    b, o = synth.make_reads_torch(src, %d, 1500, rng_seed=4100 + i, device='cuda')\n"
        "    parts.append(b.cpu().numpy()); lens.append(np.diff(o.cpu().numpy().astype(np.int64)))\n"
        "bases = np.concatenate(parts); offsets = np.zeros(1 + sum(len(x) for x in lens), np.uint64)\n"
        "offsets[1:] = np.cumsum(np.concatenate(lens)).astype(np.uint64)\n"
        "np.save(%r + '/ref.npy', ref['ref']); np.save(%r + '/bases.npy', bases); np.save(%r + '/offsets.npy', offsets)\n"
        "np.save(%r + '/truth.npy', np.array([ref['truth_index']]))\n"
    ) % (ROOT, N, S_, NB, B, d, d, d, d)
    subprocess.check_call([sys.executable, "-c", code])


@pytest.fixture(scope="module")
def snp(gpu):
    d = tempfile.mkdtemp(prefix="skx_pat_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        _generate(d)
        out = dict(ref=np.load(d + "/ref.npy"), bases=np.load(d + "/bases.npy"), offsets=np.load(d + "/offsets.npy"),
                   truth=int(np.load(d + "/truth.npy")[0]))
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    from sketchy_amd import api
    out["R"] = api.ReferenceSketch(out["ref"])
    yield out
    out["R"].close()


def test_most_long_lists_are_stored_as_patterns(snp):
    p = snp["R"].patterns
    print("patterns:", p, "rare index:", snp["R"].rare_index)
    assert p["long_lists"] > 1000
    assert 10 <= p["patterns"] <= 2000                    # ~ one per lineage (30), plus unions of lineages that share a SNP
    assert p["pattern_lists"] >= 0.8 * p["long_lists"]    # tools/diag_patterns.py: 97 % within 14 genomes of their lineage's list


@pytest.mark.parametrize("top,reuse", [(1, 0), (3, 0), (16, 0), (1, 1)])
def test_rows_and_table_through_shared_passes(snp, top, reuse):
    """(reuse = 1: policy "reuse_membership" -- the static dense rows of M are scanned for once per buffer set and kept)"""
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = snp["R"], snp["ref"], snp["bases"], snp["offsets"]
    assert R.static_dense[0] and R.static_dense[1] > 100     # 6 000 genomes: the hashes of the common ancestor are held by more than 1 024
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
            d_i, d_s = api.DeviceBuffer(B * top * 4), api.DeviceBuffer(B * top * 8)
            bufs.append((d_o, d_i, d_s))
            S.enqueue_device(d_b.ptr, d_o.ptr, B, int(offsets[a + B] - offsets[a]), d_i.ptr, d_s.ptr)
        S.sync()
        st = S.stats()
        got = [(t[1].to_numpy(np.uint32, (B, top)), t[2].to_numpy(np.uint64, (B, top))) for t in bufs]
        table = S.table()
    finally:
        d_b.free()
        for t in bufs:
            for x in t:
                x.free()
        S.close()
    print("stats:", st)
    assert st["passes_shared"] >= 2, st
    assert st["batches_compact"] >= 2, st   # the pattern rows of the compact problems were really used
    cum = None
    for i in range(NB):
        a = i * B
        e = orc.stream_fast(16, 0, S_, ref, None, bases, offsets[a:a + B + 1], top_k=top, cum=cum, rows=True)
        cum = e["cum"]
        np.testing.assert_array_equal(got[i][1], e["topk_sum"], err_msg=f"sums of batch {i}")
        np.testing.assert_array_equal(got[i][0], e["topk_idx"], err_msg=f"genomes of batch {i}")
    np.testing.assert_array_equal(table, cum, err_msg="final table")
    assert int(np.argmax(cum)) == snp["truth"]


def test_rows_and_shared_counts_through_pushes(snp):
    """synchronous pushes (one batch per pass; the per-read x per-genome debug matrix forces the full ranking: the pattern rows go into
    the group-major matrix through their bit rows) against the literal oracle loop"""
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = snp["R"], snp["ref"], snp["bases"], snp["offsets"]
    n = 300
    S = api.SumOfSharedHashes(R, top=2, max_batch_reads=n, max_batch_bases=int(offsets[n] - offsets[0]))
    got = S.push(bases, offsets[:n + 1], want_shared=True)
    e = orc.stream(16, 0, S_, ref, np.full(N, S_, np.uint32), bases[:int(offsets[n])], offsets[:n + 1], top_k=2, want_shared=True)
    np.testing.assert_array_equal(got["shared"], e["shared"])
    np.testing.assert_array_equal(got["topk_sum"], e["topk_sum"])
    np.testing.assert_array_equal(got["topk_idx"], e["topk_idx"])
    # ... then larger pushes from that table (deferred passes of one batch each: gains and candidates from the patterns)
    cum = e["cum"]
    m = 8192
    Sb = api.SumOfSharedHashes(R, top=1, max_batch_reads=m, max_batch_bases=int(offsets[n + 3 * m] - offsets[n]))
    Sb.table_add(cum)
    for j in range(3):
        a = n + j * m
        cut = offsets[a:a + m + 1]
        g = Sb.push(bases, cut)
        e2 = orc.stream_fast(16, 0, S_, ref, None, bases, cut, top_k=1, cum=cum, rows=True)
        cum = e2["cum"]
        np.testing.assert_array_equal(g["topk_sum"], e2["topk_sum"], err_msg=f"push {j}")
        np.testing.assert_array_equal(g["topk_idx"], e2["topk_idx"], err_msg=f"push {j}")
    np.testing.assert_array_equal(Sb.table(), cum)
    S.close(); Sb.close()


def test_same_rows_without_the_patterns(snp):
    """the experiments build with SKX_PATTERNS=0 (bit rows only, round 5's path) gives the same rows: run as a child so the knob is read"""
    root = ROOT
    code = (
        "import sys, numpy as np, hashlib; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sketchy_amd import api, synth\n"
        "ref = synth.make_reference(3000, 1000, rng_seed=5, device='cuda', mode='snp', n_lineages=15)\n"
        "src = torch.from_numpy(ref['truth_genome']).to('cuda')\n"
        "R = api.ReferenceSketch(ref['ref'])\n"
        "S = api.SumOfSharedHashes(R, top=1, max_batch_reads=8192, max_batch_bases=8192 * 1500)\n"
        "h = hashlib.sha256(); keep = []\n"
        "for i in range(10):\n"
        "    b, o = synth.make_reads_torch(src, 8192, 1500, rng_seed=50 + i, device='cuda')\n"
        "    d_i, d_s = api.DeviceBuffer(8192 * 4), api.DeviceBuffer(8192 * 8); keep.append((b, o, d_i, d_s))\n"
        "    S.enqueue_device(b.data_ptr(), o.data_ptr(), 8192, int(o[-1].item()), d_i.ptr, d_s.ptr)\n"
        "S.sync()\n"
        "for b, o, d_i, d_s in keep:\n"
        "    h.update(d_i.to_numpy(np.uint32, (8192,)).tobytes()); h.update(d_s.to_numpy(np.uint64, (8192,)).tobytes())\n"
        "h.update(S.table().tobytes())\n"
        "print('RESULT', h.hexdigest(), R.patterns['pattern_lists'], S.stats()['batches_compact'])\n"
    ) % root
    exp = os.path.join(root, "sketchy_amd", "libsketchy_hip_exp.so")
    outs = []
    for knob in ("1", "0"):
        env = dict(os.environ, SKX_LIB_PATH=exp, SKX_PATTERNS=knob)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split())
    print(outs)
    assert int(outs[0][2]) > 0 and int(outs[1][2]) == 0      # patterns on / off
    assert outs[0][1] == outs[1][1]                           # same rows, same table
    assert int(outs[0][3]) > 0                                # compact rankings happened


def test_rare_rows_stay_exact_through_stride_changes_and_resets(snp):
    """The rare rows of a pass that ranks on everything are written into the group-major matrix as "only what changed" while a buffer
    set keeps its row stride, in full after a stride change (rare_to_mq_kernel).  Pushes of very different sizes, with
    table resets between them, alternate the two; every small push asks for the per-read x per-genome counts (src/sketchy.rs:425-438),
    which read EVERY bit of the rows its pairs touch -- a stale bit of an earlier pass would show."""
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = snp["R"], snp["ref"], snp["bases"], snp["offsets"]
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=int(np.max(offsets[B::B] - offsets[:-B:B])))
    col = np.full(N, S_, np.uint32)
    cum, pos = None, 0
    plan = [B, 300, 64, B, 257, 4096, 300, "reset", 300, B, 1, 2048, "reset", B, B, 129]
    for step in plan:
        if step == "reset":
            S.reset(); cum = None
            continue
        n = int(step)
        cut = np.ascontiguousarray(offsets[pos:pos + n + 1])
        small = n <= 300
        got = S.push(bases, cut, want_shared=small)
        if small:
            e = orc.stream(16, 0, S_, ref, col, bases, cut, top_k=1, want_shared=True, cum=cum)
            np.testing.assert_array_equal(got["shared"], e["shared"], err_msg=f"shared counts of the push of {n} at read {pos}")
        else:
            e = orc.stream_fast(16, 0, S_, ref, None, bases, cut, top_k=1, cum=cum, rows=True)
        cum = e["cum"]
        np.testing.assert_array_equal(got["topk_sum"], e["topk_sum"], err_msg=f"push of {n} at read {pos}")
        np.testing.assert_array_equal(got["topk_idx"], e["topk_idx"], err_msg=f"push of {n} at read {pos}")
        pos += n
    np.testing.assert_array_equal(S.table(), cum)
    S.close()
