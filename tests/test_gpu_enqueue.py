"""GPU parity of skx_stream_enqueue_device: the device-resident entry point whose halves of consecutive batches are
interleaved (sketch of batch i + 1 queued before the host waits for batch i's summary).  Rows and table must be those of
the oracle (oracle/oracle.c, the restatement of src/sketchy.rs:317-356) on the same reads in the same order -- through
the usual one-pass case, batches cut into several passes (the younger batch's speculative pair gather is undone), the
block-sketcher slow path, errors surfacing one call late, and the non-speculative dictionary placement."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import pack_reads, workload
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _enqueue_stream(S, bases, offsets, cuts, top, poke=None):
    """every batch [cuts[i], cuts[i+1]) through enqueue_device with its own device rows; poke(i) runs after batch i"""
    from sketchy_amd import api
    d_b = api.DeviceBuffer.from_numpy(bases if len(bases) else np.zeros(1, np.uint8))
    bufs, rows = [], []
    try:
        for i in range(len(cuts) - 1):
            a, b = cuts[i], cuts[i + 1]
            n = b - a
            d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:b + 1], np.uint64))
            d_i = api.DeviceBuffer(max(1, n * top) * 4)
            d_s = api.DeviceBuffer(max(1, n * top) * 8)
            bufs += [d_o, d_i, d_s]
            rows.append((n, d_i, d_s))
            S.enqueue_device(d_b.ptr, d_o.ptr, n, int(offsets[b] - offsets[a]), d_i.ptr if top else None, d_s.ptr if top else None)
            if poke:
                poke(i)
        S.sync()
        idx = np.concatenate([d_i.to_numpy(np.uint32, (n, top)) for n, d_i, _ in rows]) if top else None
        val = np.concatenate([d_s.to_numpy(np.uint64, (n, top)) for n, _, d_s in rows]) if top else None
        return idx, val
    finally:
        d_b.free()
        for d in bufs:
            d.free()


def _expect(ref, s, bases, offsets, top, k=16, seed=0):
    n = len(ref)
    return orc.stream(k, seed, s, ref, np.full(n, s, np.uint32), bases, offsets, top_k=top)


def test_enqueue_matches_oracle_uneven_batches(gpu):
    from sketchy_amd import api
    ref, bases, offsets = workload(700, 400, 1500, read_len=300, rng_seed=301)
    exp = _expect(ref["ref"], 400, bases, offsets, 2)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=2, max_batch_reads=600, max_batch_bases=len(bases))
    cuts = [0, 1, 400, 401, 1000, 1000 + 37, 1500]
    idx, val = _enqueue_stream(S, bases, offsets, cuts, 2)
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.reads == 1500
    # entry points that look at the stream flush the outstanding half first: table / rank / push_device in between
    S2 = api.SumOfSharedHashes(R, top=2, max_batch_reads=600, max_batch_bases=len(bases))
    seen = {}

    def poke(i):
        if i == 1:
            seen["table"] = S2.table()
        if i == 3:
            seen["rank"] = S2.rank(3)
    idx, val = _enqueue_stream(S2, bases, offsets, cuts, 2, poke)
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(S2.table(), exp["cum"])
    part = _expect(ref["ref"], 400, bases, offsets[:401], 1)
    np.testing.assert_array_equal(seen["table"], part["cum"])
    part = _expect(ref["ref"], 400, bases, offsets[:1001], 3)
    np.testing.assert_array_equal(seen["rank"][0].reshape(-1), part["topk_idx"][-1])
    np.testing.assert_array_equal(seen["rank"][1].reshape(-1), part["topk_sum"][-1])


def test_enqueue_batches_that_need_several_passes(gpu):
    """Dense batches (reads sharing hundreds of hashes with the reference: more pairs than one pass holds) between sparse
    ones: the dense batch finds out it needs several passes AFTER the next batch's pairs were gathered speculatively
    into the buffer set it now needs itself -- the gather is undone and redone on the scan stream."""
    from sketchy_amd import api, synth
    ref = synth.make_reference(40, 3000, genome_len=24000, rng_seed=311, device="numpy")
    dense_b, dense_o = synth.make_reads(ref["genome"], 120, 4000, err=0.01, rng_seed=312)
    sparse_b, sparse_o = synth.make_reads(ref["genome"], 300, 120, err=0.05, rng_seed=313)
    d = [dense_b[int(dense_o[i]):int(dense_o[i + 1])].tobytes() for i in range(120)]
    sp = [sparse_b[int(sparse_o[i]):int(sparse_o[i + 1])].tobytes() for i in range(300)]
    reads = sp[:100] + d[:40] + sp[100:200] + d[40:80] + d[80:120] + sp[200:300]
    cuts = [0, 100, 140, 240, 280, 320, 420]
    bases, offsets = pack_reads(reads)
    exp = _expect(ref["ref"], 3000, bases, offsets, 1)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=100, max_batch_bases=int(np.diff(offsets[cuts].astype(np.int64)).max()))
    passes = []
    idx, val = _enqueue_stream(S, bases, offsets, cuts, 1, poke=None)
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    # the dense batches really were cut: one of them alone, synchronously
    S1 = api.SumOfSharedHashes(R, top=1, max_batch_reads=100, max_batch_bases=int(np.diff(offsets[cuts].astype(np.int64)).max()))
    d_b, d_o = api.DeviceBuffer.from_numpy(bases), api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[100:141]))
    S1.push_device(d_b.ptr, d_o.ptr, 40, int(offsets[140] - offsets[100]), None, None)
    passes.append(S1.stats()["last_passes"])
    d_b.free(); d_o.free()
    assert passes[0] > 1, passes


def test_enqueue_block_sketcher_slow_path(gpu):
    """A production-mode batch with reads whose in-range hashes overflow a wave's slots (the `big` list) is finished by
    the block sketcher in its back half -- queued behind the NEXT batch's sketch, on that batch's other side."""
    from sketchy_amd import api, synth
    ref = synth.make_reference(12, 20000, genome_len=30000, rng_seed=181, device="numpy")
    bases, offsets = synth.make_reads(ref["genome"], 24, 9000, err=0.02, rng_seed=182, lognormal_sigma=0.8, min_len=300, max_len=29000)
    assert np.diff(offsets.astype(np.int64)).max() > 18000
    exp = _expect(ref["ref"], 20000, bases, offsets, 1)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=8, max_batch_bases=int(np.diff(offsets[::8].astype(np.int64)).max()))
    idx, val = _enqueue_stream(S, bases, offsets, [0, 8, 16, 24], 1)
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.stats()["reads_block_sketcher"] > 0


def test_enqueue_errors_surface_one_call_late(gpu):
    from sketchy_amd import api, _lib
    ref, bases, offsets = workload(40, 200, 60, rng_seed=75)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=20, max_batch_bases=len(bases))
    bad = offsets[:21].copy()
    bad[7], bad[8] = bad[8], bad[7]  # not monotonic
    d_b = api.DeviceBuffer.from_numpy(bases)
    d_bad = api.DeviceBuffer.from_numpy(np.ascontiguousarray(bad))
    d_o1 = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[20:41]))
    d_o0 = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[:21]))
    S.enqueue_device(d_b.ptr, d_bad.ptr, 20, int(bad[20]), None, None)   # accepted: nothing has looked at it yet
    with pytest.raises(_lib.SketchyHipError) as e:
        # (batches enqueued back to back share a pass: the next calls only queue their sketches; the call that finds the group
        # full -- or a flush -- runs the shared back half and finds the error)
        S.enqueue_device(d_b.ptr, d_o1.ptr, 20, int(offsets[40] - offsets[20]), None, None)
        S.enqueue_device(d_b.ptr, d_o1.ptr, 20, int(offsets[40] - offsets[20]), None, None)
        S.flush()
    assert e.value.code == _lib.ERR_INVALID
    assert S.reads == 0 and not S.table().any()     # all three batches dropped
    S.enqueue_device(d_b.ptr, d_bad.ptr, 20, int(bad[20]), None, None)
    with pytest.raises(_lib.SketchyHipError):
        S.flush()
    assert S.reads == 0 and not S.table().any()
    # the stream is still usable, and gives what it should
    S.enqueue_device(d_b.ptr, d_o0.ptr, 20, int(offsets[20]), None, None)
    S.enqueue_device(d_b.ptr, d_o1.ptr, 20, int(offsets[40] - offsets[20]), None, None)
    exp = _expect(ref["ref"], 200, bases, offsets[:41], 1)
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.reads == 40
    for d in (d_b, d_bad, d_o1, d_o0):
        d.free()


def test_a_refused_batch_leaves_nothing_behind(gpu):
    """A LARGE batch that fails validation (its speculative pair gather has already filled the pass's hash set) followed by
    SMALLER batches of DIFFERENT reads: the next passes on that buffer set must see only their own keys (|Q| <= their pair
    count sizes every matrix of the pass) -- rows and table against the oracle, through both entry points, both failure
    kinds (offsets not monotonic / a read outside n_bases), twice so both buffer sets are hit."""
    from sketchy_amd import api, _lib
    ref, bases, offsets = workload(700, 400, 1300, read_len=900, rng_seed=77)
    R = api.ReferenceSketch(ref["ref"])
    n_big, n_small = 1000, 24
    d_b = api.DeviceBuffer.from_numpy(bases)
    bad = offsets[:n_big + 1].copy()
    bad[500], bad[501] = bad[501], bad[500]
    d_bad = api.DeviceBuffer.from_numpy(np.ascontiguousarray(bad))
    d_big = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[:n_big + 1]))
    small = [(n_big + i * n_small, n_big + (i + 1) * n_small) for i in range(6)]
    d_small = [api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:b + 1])) for a, b in small]
    exp = _expect(ref["ref"], 400, bases[int(offsets[n_big]):], offsets[n_big:] - offsets[n_big], 1)
    for entry in ("push", "enqueue"):
        S = api.SumOfSharedHashes(R, top=1, max_batch_reads=n_big, max_batch_bases=int(offsets[n_big]))
        call = S.push_device if entry == "push" else S.enqueue_device
        rows = []
        for i, ((a, b), d_o) in enumerate(zip(small, d_small)):
            if i in (0, 1, 3):  # a failing large batch in front of this small one
                with pytest.raises(_lib.SketchyHipError) as e:
                    if i == 3:   # vouch for fewer bytes than the offsets reach
                        call(d_b.ptr, d_big.ptr, n_big, int(offsets[n_big // 2]), None, None)
                    else:
                        call(d_b.ptr, d_bad.ptr, n_big, int(offsets[n_big]), None, None)
                    S.flush()
                assert e.value.code == _lib.ERR_INVALID
            d_i, d_s = api.DeviceBuffer(n_small * 4), api.DeviceBuffer(n_small * 8)
            call(d_b.ptr, d_o.ptr, n_small, int(offsets[b] - offsets[a]), d_i.ptr, d_s.ptr)
            S.sync()
            rows.append((d_i.to_numpy(np.uint32, (n_small,)), d_s.to_numpy(np.uint64, (n_small,))))
            d_i.free(); d_s.free()
        n = n_small * len(small)
        np.testing.assert_array_equal(np.concatenate([r[0] for r in rows]), exp["topk_idx"][:n, 0], err_msg=entry)
        np.testing.assert_array_equal(np.concatenate([r[1] for r in rows]), exp["topk_sum"][:n, 0], err_msg=entry)
        assert S.reads == n
        full = _expect(ref["ref"], 400, bases[int(offsets[n_big]):int(offsets[n_big + n])], offsets[n_big:n_big + n + 1] - offsets[n_big], 1)
        np.testing.assert_array_equal(S.table(), full["cum"], err_msg=entry)
        S.close()
    for d in [d_b, d_bad, d_big] + d_small:
        d.free()


def test_row_pool_grows_when_a_batch_is_denser_than_it_is_sized_for(gpu):
    """Production sketch rows are exact-size reservations out of a per-side pool (16 pairs per read of the largest batch, at
    least 2^20 entries).  9 000 reads with ~160 pairs each ask for more: the summary reports the overflow and the request,
    the pool is re-allocated and the batch sketched again -- same rows as the oracle, through push_device and enqueue_device,
    with ordinary batches before and after, and the block-sketcher path on top (long dense reads)."""
    from sketchy_amd import api, synth
    ref = synth.make_reference(24, 3000, genome_len=24000, rng_seed=811, device="numpy")
    dense_b, dense_o = synth.make_reads(ref["genome"], 9000, 1500, err=0.01, rng_seed=812)
    long_b, long_o = synth.make_reads(ref["genome"], 6, 20000, err=0.01, rng_seed=813)
    reads = [dense_b[int(dense_o[i]):int(dense_o[i + 1])].tobytes() for i in range(9000)]
    longs = [long_b[int(long_o[i]):int(long_o[i + 1])].tobytes() for i in range(6)]
    stream = reads[:50] + reads[50:9000] + longs + reads[:200]
    cuts = [0, 50, 9000, 9006, 9206]
    bases, offsets = pack_reads(stream)
    exp = _expect(ref["ref"], 3000, bases, offsets, 1)
    R = api.ReferenceSketch(ref["ref"])
    for entry in ("enqueue", "push"):
        S = api.SumOfSharedHashes(R, top=1, max_batch_reads=9000, max_batch_bases=int(np.diff(offsets[cuts].astype(np.int64)).max()))
        if entry == "enqueue":
            idx, val = _enqueue_stream(S, bases, offsets, cuts, 1)
        else:
            d_b = api.DeviceBuffer.from_numpy(bases)
            parts = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:b + 1]))
                d_i, d_s = api.DeviceBuffer((b - a) * 4), api.DeviceBuffer((b - a) * 8)
                S.push_device(d_b.ptr, d_o.ptr, b - a, int(offsets[b] - offsets[a]), d_i.ptr, d_s.ptr)
                S.sync()
                parts.append((d_i.to_numpy(np.uint32, (b - a, 1)), d_s.to_numpy(np.uint64, (b - a, 1))))
                for d in (d_o, d_i, d_s):
                    d.free()
            d_b.free()
            idx, val = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        np.testing.assert_array_equal(idx, exp["topk_idx"], err_msg=entry)
        np.testing.assert_array_equal(val, exp["topk_sum"], err_msg=entry)
        np.testing.assert_array_equal(S.table(), exp["cum"], err_msg=entry)
        st = S.stats()
        assert st["row_pool_grown"] >= 1 and st["reads_block_sketcher"] >= 1, st
        S.close()


def test_more_distinct_query_hashes_than_matrix_rows(gpu):
    """The bit matrices of a pass have `stream_query_rows` rows (default 65 536; never fewer than one read can need: s).  A
    batch whose distinct query hashes exceed them -- known from the speculative gather's key count in the published summary --
    is cut into passes by its pair counts.  Forced with 128 rows and reads sketched with s = 192 (the collection's FIRST
    sketch is that short, src/sketchy.rs:520-527): ~1 000 distinct hashes per batch against 192 rows -- rows and table as the
    oracle's, many passes, the younger batch's speculation undone; the one-workgroup dictionary sort on every pass."""
    from sketchy_amd import api
    ref, bases, offsets = workload(300, 1000, 900, read_len=2000, rng_seed=821, err=0.01)
    col_len = np.full(300, 1000, np.uint32)
    col_len[0] = 192
    exp = orc.stream(16, 0, 192, ref["ref"], col_len, bases, offsets, top_k=2)
    R = api.ReferenceSketch(ref["ref"], col_len, s=192)
    try:
        api.set_option("stream_query_rows", 128)
        S = api.SumOfSharedHashes(R, top=2, max_batch_reads=400, max_batch_bases=400 * 2000)
    finally:
        api.set_option("stream_query_rows", 0)
    idx, val = _enqueue_stream(S, bases, offsets, [0, 5, 400, 401, 800, 900], 2)
    np.testing.assert_array_equal(idx, exp["topk_idx"])
    np.testing.assert_array_equal(val, exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.stats()["passes"] > 8 and S.stats()["dictionary_size"] <= 192
    S.reset()
    got = S.push(bases, offsets[:301], want_shared=True)
    full = orc.stream(16, 0, 192, ref["ref"], col_len, bases, offsets[:301], top_k=2, want_shared=True)
    np.testing.assert_array_equal(got["shared"], full["shared"])
    np.testing.assert_array_equal(got["topk_idx"], full["topk_idx"])


@pytest.mark.parametrize("env", [{"SKX_SPEC_INSERT": "0"}, {"SKX_PIPELINE": "2"}, {"SKX_PIPELINE": "1"}, {"SKX_PASS_READS": "128"},
                                 {"SKX_TWO_LEVEL": "1"}, {"SKX_TWO_LEVEL": "0"}, {"SKX_COALESCE": "3"}, {"SKX_SCAN_NT": "0"}])
def test_enqueue_under_other_placements(gpu, env):
    """the same stream with the pair gather on the scan stream (never speculative), with fewer pipeline streams, with
    every batch cut into passes of 128 reads (each enqueue undoes the younger batch's speculation), with the ranking's counts
    always / never in two levels (chunk sums first, per-segment increments only where a candidate can be), and with groups
    of three batches per pass -- separate processes: the knobs are read once"""
    here = os.path.dirname(os.path.abspath(__file__))
    from helpers import exp_env
    e = exp_env(**env)
    r = subprocess.run([sys.executable, os.path.join(here, "enqueue_check.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "enqueue_check ok" in r.stdout


def _groups(n_batches, n, flush_after=()):
    """(shared passes, passes) when up to n batches enqueued back to back share a pass and nothing else limits a group"""
    shared = passes = size = 0
    for i in range(n_batches):
        size += 1
        if size == n or i in flush_after or i == n_batches - 1:
            passes += 1
            shared += size > 1
            size = 0
    return shared, passes


@pytest.mark.parametrize("n", [2, 3, 8])
def test_enqueued_batches_share_a_pass(gpu, n):
    """Option "stream_coalesce" = n: up to n batches enqueued back to back take ONE dictionary / scan / transpose and a ranking
    each.  Same rows and table as the oracle -- with ragged batch sizes, a group left incomplete at the end, entry points that
    flush in between, and with the option off.  skx_stream_stats()[11] counts the shared passes."""
    from sketchy_amd import api
    ref, bases, offsets = workload(700, 400, 1400, read_len=400, rng_seed=901)
    exp = _expect(ref["ref"], 400, bases, offsets, 2)
    R = api.ReferenceSketch(ref["ref"])
    cuts = [0, 200, 400, 650, 700, 900, 1100, 1400]   # seven batches
    kw = dict(top=2, max_batch_reads=300, max_batch_bases=len(bases))

    def make(coalesce):
        before = api.get_option("stream_coalesce")
        try:
            api.set_option("stream_coalesce", coalesce)
            return api.SumOfSharedHashes(R, **kw)
        finally:
            api.set_option("stream_coalesce", before)

    def check(S, expected, poke=None):
        idx, val = _enqueue_stream(S, bases, offsets, cuts, 2, poke)
        np.testing.assert_array_equal(idx, exp["topk_idx"])
        np.testing.assert_array_equal(val, exp["topk_sum"])
        np.testing.assert_array_equal(S.table(), exp["cum"])
        st = S.stats()
        assert (st["passes_shared"], st["passes"]) == expected and st["groups_unshared"] == 0, st
        assert S.reads == 1400

    check(make(n), _groups(7, n))
    # flushes in between: the groups end there
    S = make(n)
    check(S, _groups(7, n, (0, 2, 3, 4)), poke=lambda i: S.flush() if i in (0, 2, 3, 4) else None)
    # the option off: one pass per batch
    check(make(1), (0, 7))


@pytest.mark.parametrize("lanes", [1, 3, 4])
def test_ranking_lanes_option(gpu, lanes):
    """Option "rank_lanes" (default 2; every other test runs with that): the rankings of the batches of a shared pass run as chains
    on 1 / 3 / 4 HIP streams, each with its own scratch, the tables rotating over lanes + 1 buffers.  Same rows and table as the
    oracle -- top-1 (the pruned two-level path) and top-2, with a reset in between (the rotation starts anywhere)."""
    from sketchy_amd import api
    ref, bases, offsets = workload(700, 400, 1400, read_len=400, rng_seed=917)
    R = api.ReferenceSketch(ref["ref"])
    cuts = [0, 150, 300, 500, 650, 800, 900, 1000, 1150, 1250, 1400]   # ten batches: one full group of eight and a rest
    before = api.get_option("rank_lanes")
    try:
        api.set_option("rank_lanes", lanes)
        assert api.get_option("rank_lanes") == lanes
        for top in (1, 2):
            exp = _expect(ref["ref"], 400, bases, offsets, top)
            S = api.SumOfSharedHashes(R, top=top, max_batch_reads=300, max_batch_bases=len(bases))
            for _ in range(2):
                idx, val = _enqueue_stream(S, bases, offsets, cuts, top)
                np.testing.assert_array_equal(idx, exp["topk_idx"])
                np.testing.assert_array_equal(val, exp["topk_sum"])
                np.testing.assert_array_equal(S.table(), exp["cum"])
                S.reset()
            S.close()
    finally:
        api.set_option("rank_lanes", before)
    from sketchy_amd import _lib
    with pytest.raises(_lib.SketchyHipError):
        api.set_option("rank_lanes", 5)


def test_a_group_that_does_not_fit_one_pass_is_unshared(gpu):
    """One batch fits the bit matrices, two together do not (matrix rows sized that way): the pair is un-shared in its back half --
    the joint hash set emptied, each batch gathered again and given its own pass -- and the stream stops pairing batches of that
    size.  Reads sketched with s = 192 (the collection's first sketch is that short) so that the matrices may be smaller than the
    other sketches' 1 000 hashes.  Then the same with groups of up to eight."""
    from sketchy_amd import api
    ref, bases, offsets = workload(300, 1000, 240, read_len=2000, rng_seed=905, err=0.01)
    col_len = np.full(300, 1000, np.uint32)
    col_len[0] = 192
    exp2 = orc.stream(16, 0, 192, ref["ref"], col_len, bases, offsets, top_k=2)
    R2 = api.ReferenceSketch(ref["ref"], col_len, s=192)
    cuts2 = [0, 30, 60, 75, 90, 120, 150, 180, 210, 240]   # nine batches
    kw2 = dict(top=2, max_batch_reads=60, max_batch_bases=len(bases))
    P = api.SumOfSharedHashes(R2, **kw2)

    def distinct(a, b):
        P.reset()
        P.push(bases, offsets[a:b + 1])
        assert P.stats()["last_passes"] == 1
        return P.stats()["dictionary_size"]
    single = [distinct(a, b) for a, b in zip(cuts2[:-1], cuts2[1:])]
    rows = (max(single) + 63) // 64 * 64
    assert rows >= 192 and distinct(0, 60) > rows, (single, rows)   # (else this workload does not test what it should)
    before = api.get_option("stream_coalesce")
    for n in (2, 8):
        try:
            api.set_option("stream_query_rows", rows)
            api.set_option("stream_coalesce", n)
            S2 = api.SumOfSharedHashes(R2, **kw2)
        finally:
            api.set_option("stream_query_rows", 0)
            api.set_option("stream_coalesce", before)
        idx, val = _enqueue_stream(S2, bases, offsets, cuts2, 2)
        np.testing.assert_array_equal(idx, exp2["topk_idx"], err_msg=str(n))
        np.testing.assert_array_equal(val, exp2["topk_sum"], err_msg=str(n))
        np.testing.assert_array_equal(S2.table(), exp2["cum"], err_msg=str(n))
        st = S2.stats()
        assert st["groups_unshared"] >= 1 and st["passes"] <= 9 - st["passes_shared"], (n, st, single)


def test_an_error_in_the_second_batch_of_a_pair(gpu):
    """the batch enqueued BEFORE a faulty one is processed (it shares nothing with it any more), the faulty one is dropped"""
    from sketchy_amd import api, _lib
    ref, bases, offsets = workload(40, 200, 60, rng_seed=75)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=20, max_batch_bases=len(bases))
    bad = offsets[20:41].copy()
    bad[7], bad[8] = bad[8], bad[7]
    d_b = api.DeviceBuffer.from_numpy(bases)
    d_bad = api.DeviceBuffer.from_numpy(np.ascontiguousarray(bad))
    d_o0 = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[:21]))
    d_o1 = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[20:41]))
    S.enqueue_device(d_b.ptr, d_o0.ptr, 20, int(offsets[20]), None, None)
    S.enqueue_device(d_b.ptr, d_bad.ptr, 20, int(offsets[40] - offsets[20]), None, None)
    with pytest.raises(_lib.SketchyHipError) as e:
        S.flush()
    assert e.value.code == _lib.ERR_INVALID
    exp = _expect(ref["ref"], 200, bases, offsets[:21], 1)
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.reads == 20
    S.enqueue_device(d_b.ptr, d_o1.ptr, 20, int(offsets[40] - offsets[20]), None, None)
    exp = _expect(ref["ref"], 200, bases, offsets[:41], 1)
    np.testing.assert_array_equal(S.table(), exp["cum"])
    assert S.reads == 40
    for d in (d_b, d_bad, d_o0, d_o1):
        d.free()
