"""WHOLE-BATCH oracle parity at full size (BASELINE config C2: 40 000 genomes x s=10 000, batches of 98 304 reads -- what
bench.py times), at the head of a stream AND mid-stream, on the PRODUCT library.

The literal oracle loop (orc_stream) does ~8 reads/s at this size; orc_stream_fast (oracle/oracle.c: the same rows and
table through one membership matrix per block of reads, pinned against orc_stream in tests/test_oracle.py) checks every
row of a batch in seconds.  Twelve batches are enqueued back to back (skx_stream_enqueue_device): batches 0-7 share the
first pass over the reference, 8-11 the second; consecutive batches of a pass are ranked on alternating lanes; the
library itself decides where the counts go to two levels.  ALL rows of batches 0, 1 (stream head, both lanes) and 9, 10
(after >= 8 batches, both lanes of the second shared pass) and the final table are compared with the checker, which runs
the whole stream from ITS OWN table.  Two workloads: the bench's near-tie (reads from the common ancestor of a
random-hash clone tree) and SURVEY.md 8(d)'s (SNP clone tree, reads from one truth strain: a leader exists, the pruned
paths decide).  Reference file:lines: src/sketchy.rs:337-349 (table update + rank), :425-438 (intersection)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, S_, B = 40000, 10000, 98304
N_BATCHES = 12
ROW_BATCHES = (0, 1, 9, 10)


def _generate(mode, d):
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from sketchy_amd import synth\n"
        "dev = 'cuda' if torch.cuda.is_available() else 'cpu'\n"
        "mode = %r\n"
        "ref = synth.make_reference(%d, %d, rng_seed=1, device=dev if mode == 'snp' or dev == 'cuda' else 'numpy', mode=mode)\n"
        "src = ref['truth_genome'] if mode == 'snp' else ref['genome']\n"
        "parts, lens = [], []\n"
        "for i in range(%d):\n"
        "    if dev == 'cuda':\n"
        "        b, o = synth.make_reads_torch(src, %d, 1500, rng_seed=777 + i, device=dev)\n"
        "        b, o = b.cpu().numpy(), o.cpu().numpy().astype(np.uint64)\n"
        "    else:\n"
        "        b, o = synth.make_reads(src, %d, 1500, rng_seed=777 + i)\n"
        "    parts.append(b); lens.append(np.diff(o.astype(np.int64)))\n"
        "bases = np.concatenate(parts); offsets = np.zeros(1 + sum(len(x) for x in lens), np.uint64)\n"
        "offsets[1:] = np.cumsum(np.concatenate(lens)).astype(np.uint64)\n"
        "np.save(%r + '/ref.npy', ref['ref']); np.save(%r + '/bases.npy', bases); np.save(%r + '/offsets.npy', offsets)\n"
        "np.save(%r + '/truth.npy', np.array([ref.get('truth_index', -1)]))\n"
    ) % (ROOT, mode, N, S_, N_BATCHES, B, B, d, d, d, d)
    subprocess.check_call([sys.executable, "-c", code])


@pytest.fixture(scope="module", params=["pool", "snp"])
def c2(gpu, request):
    d = tempfile.mkdtemp(prefix="skx_c2o_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        _generate(request.param, d)
        out = dict(mode=request.param, ref=np.ascontiguousarray(np.load(d + "/ref.npy", mmap_mode="r")),
                   bases=np.load(d + "/bases.npy"), offsets=np.load(d + "/offsets.npy"), truth=int(np.load(d + "/truth.npy")[0]))
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    from sketchy_amd import api
    out["R"] = api.ReferenceSketch(out["ref"])
    yield out
    out["R"].close()


def test_every_row_of_full_batches_at_the_head_and_mid_stream(c2):
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = c2["R"], c2["ref"], c2["bases"], c2["offsets"]
    free0, _ = api.device_mem(0)
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=int(np.max(offsets[B::B] - offsets[:-B:B])))
    d_b = api.DeviceBuffer.from_numpy(bases)
    bufs = []
    stream_bytes = None
    try:
        for i in range(N_BATCHES):
            a = i * B
            d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:a + B + 1]))
            d_i, d_s = api.DeviceBuffer(B * 4), api.DeviceBuffer(B * 8)
            bufs.append((d_o, d_i, d_s))
            S.enqueue_device(d_b.ptr, d_o.ptr, B, int(offsets[a + B] - offsets[a]), d_i.ptr, d_s.ptr)
        S.sync()
        st = S.stats()
        assert st["passes_shared"] >= 2 and st["groups_unshared"] == 0, st     # the batches really shared their passes
        # footprint of the stream after its matrices have grown to what this workload needs (the buffers of this test are not the stream's)
        free1, _ = api.device_mem(0)
        own = len(bases) + sum(x.nbytes for t in bufs for x in t)
        stream_bytes = free0 - free1 - own
        got = {i: (bufs[i][1].to_numpy(np.uint32, (B, 1)), bufs[i][2].to_numpy(np.uint64, (B, 1))) for i in ROW_BATCHES}
        table = S.table()
    finally:
        d_b.free()
        for t in bufs:
            for x in t:
                x.free()
        S.close()
    # Footprint at C2 (SURVEY 8(d) reference: 893 k distinct hashes, 161 k distinct query hashes per batch -- the matrices' rows grow to ~1 M):
    # reference = the 3.2 GB matrix + the rare-hash index (lists, bit rows, their transpose, patterns); stream: round 5 held M (7.6 GB at
    # 1.2 M rows) + 2 x Mq + per-row arrays, ~25-30 GB with the index.  Since round 6 M holds the reference's static dense rows only
    # (2 x 52 MB): the stream's bound is the two group-major matrices.  Stated bounds: index + patterns <= 3.5 GB, one stream <= 16 GB
    # (SNP reference) / 8 GB (pool reference).
    ri, pt = R.rare_index, R.patterns
    print(f"{c2['mode']}: footprint: matrix {R.pass_bytes / 2**30:.2f} GiB, rare-hash index {ri['bytes'] / 2**30:.2f} GiB, patterns "
          f"{pt['bytes'] / 2**30:.3f} GiB ({pt['pattern_lists']} of {pt['long_lists']} long lists, {pt['patterns']} patterns), static dense "
          f"{R.static_dense}, stream {stream_bytes / 2**30:.2f} GiB, query rows {st['query_rows']}")
    assert ri["bytes"] + pt["bytes"] <= 3.5 * 2**30
    assert stream_bytes <= (16 if c2["mode"] == "snp" else 8) * 2**30, stream_bytes
    cum = None
    pairs = 0
    for i in range(N_BATCHES):
        a = i * B
        e = orc.stream_fast(16, 0, S_, ref, None, bases, offsets[a:a + B + 1], top_k=1, cum=cum, rows=i in ROW_BATCHES)
        cum = e["cum"]
        pairs += e["stats"]["pairs"]
        if i in ROW_BATCHES:
            np.testing.assert_array_equal(got[i][1], e["topk_sum"], err_msg=f"{c2['mode']}: sums of batch {i}")
            np.testing.assert_array_equal(got[i][0], e["topk_idx"], err_msg=f"{c2['mode']}: genomes of batch {i}")
    np.testing.assert_array_equal(table, cum, err_msg="final table")
    lead = int(np.argmax(cum))
    print(f"{c2['mode']}: {pairs / (N_BATCHES * B):.2f} in-range hashes per read, leader {lead} (truth {c2['truth']}), "
          f"sum {int(cum.max())}, second {int(np.sort(cum)[-2])}")
    if c2["mode"] == "snp":
        assert lead == c2["truth"]


def test_every_row_of_a_lone_full_batch_top5_push(c2):
    """the synchronous entry point (skx_stream_push: one batch, one pass, nothing shared) and the top-k rows at full size"""
    from oracle import oracle as orc
    from sketchy_amd import api
    R, ref, bases, offsets = c2["R"], c2["ref"], c2["bases"], c2["offsets"]
    a = 3 * B
    n = 32768
    S = api.SumOfSharedHashes(R, top=5, max_batch_reads=n, max_batch_bases=int(offsets[a + n] - offsets[a]))
    got = S.push(bases, offsets[a:a + n + 1])
    e = orc.stream_fast(16, 0, S_, ref, None, bases, offsets[a:a + n + 1], top_k=5)
    np.testing.assert_array_equal(got["topk_sum"], e["topk_sum"])
    np.testing.assert_array_equal(got["topk_idx"], e["topk_idx"])
    np.testing.assert_array_equal(S.table(), e["cum"])
    S.close()
