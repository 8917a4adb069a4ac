"""Random walks over the stream's entry points -- enqueue_device / push_device / push / submit + wait / drain / table / rank /
flush / sync / reset, ASCII and 4-bit packed input -- on one stream, against the oracle fed the same reads in the same
order.  The batch halves, staging slots, buffer sets and pair slots of the pipeline are a state machine; this walks it
in orders no hand-written test does.  SKX_TEST_SEED selects another set of walks."""
import os

import numpy as np
import pytest

from helpers import workload, workload_snp
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _walk(seed, tree=False):
    from sketchy_amd import api
    rng = np.random.default_rng(seed)
    n, s, top = int(rng.choice([130, 600, 1100])), int(rng.choice([100, 300])), int(rng.choice([1, 1, 3]))
    if tree:
        # a SURVEY 8(d) clone tree with more than 1 024 genomes: the reference gets its rare-hash index, the lists' patterns and the
        # static dense dictionary -- the walk then also crosses the deferred passes' candidate paths
        n, s = int(rng.choice([1500, 2600])), 300
        ref, bases, offsets = workload_snp(n, s, 4000, read_len=int(rng.choice([200, 500])), rng_seed=3 + seed % 2, n_lineages=13)
    else:
        ref, bases, offsets = workload(n, s, 4000, read_len=int(rng.choice([200, 500])), rng_seed=int(rng.integers(1, 10 ** 6)))
    col_len = np.full(n, s, np.uint32)
    packed, poff = api.pack_reads(bases, offsets)
    R = api.ReferenceSketch(ref["ref"])
    # (how many enqueued batches may share a pass, and -- now and then -- bit matrices so small that groups have to be un-shared)
    # ... and on how many lanes the rankings of a shared pass run
    before, lanes_before = api.get_option("stream_coalesce"), api.get_option("rank_lanes")
    try:
        api.set_option("stream_coalesce", int(rng.choice([1, 2, 3, 5, 8, 8])))
        api.set_option("stream_query_rows", int(rng.choice([0, 0, 0, 64])))
        api.set_option("rank_lanes", int(rng.choice([1, 2, 2, 2, 3, 4])))
        api.set_option("reuse_membership", int(rng.choice([0, 0, 1])) if tree else 0)
        S = api.SumOfSharedHashes(R, top=top, max_batch_reads=400, max_batch_bases=400 * 700)
    finally:
        api.set_option("stream_coalesce", before)
        api.set_option("stream_query_rows", 0)
        api.set_option("rank_lanes", lanes_before)
        api.set_option("reuse_membership", 0)
    d_ascii, d_packed = api.DeviceBuffer.from_numpy(bases), api.DeviceBuffer.from_numpy(packed)
    h_ascii, h_packed = api.HostBuffer(len(bases)), api.HostBuffer(len(packed))
    h_ascii.view(np.uint8)[:] = bases
    h_packed.view(np.uint8)[:] = packed
    keep = [d_ascii, d_packed]
    cum = None            # oracle's table so far
    pos = 0               # next read
    checks = []           # (kind, device/host buffers, expected rows) resolved at the next synchronisation point
    is_packed = False
    outstanding = 0       # submitted, not yet waited

    def expect(a, b):
        nonlocal cum
        e = orc.stream(16, 0, s, ref["ref"], col_len, bases, offsets[a:b + 1], top_k=top, cum=cum)
        cum = e["cum"]
        return e

    def resolve():
        for kind, bi, bs, e, m in checks:
            gi = bi.to_numpy(np.uint32, (m, top)) if kind == "dev" else bi.view(np.uint32, m * top).reshape(m, top)
            gs = bs.to_numpy(np.uint64, (m, top)) if kind == "dev" else bs.view(np.uint64, m * top).reshape(m, top)
            np.testing.assert_array_equal(gi, e["topk_idx"])
            np.testing.assert_array_equal(gs, e["topk_sum"])
        checks.clear()

    for _ in range(45):
        m = int(rng.choice([1, 7, 64, 65, 200, 400]))
        if pos + m > 4000:
            break
        op = rng.choice(["enqueue", "enqueue", "enqueue", "push_device", "push", "submit", "table", "rank", "flush", "sync", "packed", "reset"])
        off = (poff if is_packed else offsets)[pos:pos + m + 1]
        if op in ("enqueue", "push_device"):
            if outstanding:
                S.drain(); outstanding = 0
            d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(off))
            d_i, d_s = api.DeviceBuffer(m * top * 4), api.DeviceBuffer(m * top * 8)
            keep += [d_o, d_i, d_s]
            fn = S.enqueue_device if op == "enqueue" else S.push_device
            fn((d_packed if is_packed else d_ascii).ptr, d_o.ptr, m, int(off[-1] - off[0]), d_i.ptr, d_s.ptr)
            checks.append(("dev", d_i, d_s, expect(pos, pos + m), m))
            pos += m
        elif op == "push":
            if outstanding:
                S.drain(); outstanding = 0
            got = S.push(packed if is_packed else bases, off)
            e = expect(pos, pos + m)
            np.testing.assert_array_equal(got["topk_idx"].reshape(m, top), e["topk_idx"])
            np.testing.assert_array_equal(got["topk_sum"].reshape(m, top), e["topk_sum"])
            pos += m
        elif op == "submit":
            S.flush()
            ho = api.HostBuffer((m + 1) * 8)
            ho.view(np.uint64)[:] = off
            hi, hs = api.HostBuffer(m * top * 4), api.HostBuffer(m * top * 8)
            keep += [ho, hi, hs]
            t = S.submit((h_packed if is_packed else h_ascii).ptr, ho.ptr, m, hi.ptr, hs.ptr)
            outstanding += 1
            checks.append(("host", hi, hs, expect(pos, pos + m), m))
            pos += m
            if rng.random() < 0.4:
                S.wait(t)
        elif op == "table":
            if outstanding:
                S.drain(); outstanding = 0
            t = S.table()
            np.testing.assert_array_equal(t, cum if cum is not None else np.zeros(n, np.uint64))
            resolve()
        elif op == "rank" and cum is not None:
            if outstanding:
                S.drain(); outstanding = 0
            gi, gs = S.rank(1)
            best = int(np.lexsort((np.arange(n), -cum.astype(np.int64)))[0])
            assert int(gi.reshape(-1)[0]) == best and int(gs.reshape(-1)[0]) == int(cum[best])
        elif op == "flush":
            S.flush()
        elif op == "sync":
            if outstanding:
                S.drain(); outstanding = 0
            S.sync()
            resolve()
        elif op == "packed":
            if outstanding:
                S.drain(); outstanding = 0
            is_packed = not is_packed
            S.set_packed_input(is_packed)
        elif op == "reset" and rng.random() < 0.3:
            if outstanding:
                S.drain(); outstanding = 0
            S.sync()
            resolve()
            S.reset()
            cum = None
    if outstanding:
        S.drain()
    S.sync()
    resolve()
    assert pos > 0
    np.testing.assert_array_equal(S.table(), cum if cum is not None else np.zeros(n, np.uint64))
    if tree:
        st = S.stats()
        print("walk", seed, "n", n, "top", top, "static", R.static_dense, "patterns", R.patterns["pattern_lists"], "passes", st["passes"],
              "shared", st["passes_shared"], "compact", st["batches_compact"], "full", st["batches_full"])
    for d in keep:
        d.free()
    h_ascii.free(); h_packed.free()
    return pos


def test_random_walks_over_the_stream_api(gpu):
    base = int(os.environ.get("SKX_TEST_SEED", "5"))
    total = sum(_walk(1000 * base + i) for i in range(6))
    assert total > 3000  # (the walks really pushed reads through)


def test_random_walks_on_a_clone_tree_reference(gpu):
    """the same walks where skx_ref_create has built everything it builds for a large collection (rare-hash index, patterns, static
    dense dictionary), with and without the `reuse_membership` policy"""
    base = int(os.environ.get("SKX_TEST_SEED", "5"))
    total = sum(_walk(2000 * base + i, tree=True) for i in range(5))
    assert total > 2500
