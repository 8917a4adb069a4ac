"""Run by tests/test_gpu_fullsize.py in a subprocess with library knobs forced through the environment (the
library reads them once per process): SKX_SCAN_SPLIT / SKX_SCAN_BIG (scan-kernel variant), SKX_PASS_READS (reads per
pass: several passes per push), SKX_PIPELINE (stream depth), SKX_NO_FILTER.  Parity against the oracle on small and
ragged inputs, top-3 with the debug outputs and top-1 (pruned path) through host and device pushes."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import workload  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from sketchy_amd import api  # noqa: E402

rng = np.random.default_rng(0)
for n, s, n_reads, seed in ((300, 500, 400, 1), (513, 96, 50, 2), (40, 3000, 64, 3)):
    ref, bases, offsets = workload(n, s, n_reads, read_len=900, rng_seed=seed, genome_len=max(60000, 280 * s))
    hashes = ref["ref"].copy()
    hashes[n // 2] = hashes[1]
    col_len = rng.integers(s // 2, s + 1, size=n).astype(np.uint32)
    col_len[0] = s
    exp = orc.stream(16, 0, s, hashes, col_len, bases, offsets, top_k=3, want_shared=True)
    R = api.ReferenceSketch(hashes, col_len)
    S = api.SumOfSharedHashes(R, top=3, max_batch_reads=n_reads, max_batch_bases=len(bases))
    got = S.push(bases, offsets, want_shared=True)
    assert np.array_equal(got["shared"], exp["shared"]), "shared"
    assert np.array_equal(got["topk_idx"], exp["topk_idx"]) and np.array_equal(got["topk_sum"], exp["topk_sum"]), "rows"
    assert np.array_equal(S.table(), exp["cum"]), "table"
    # production path (no debug outputs), top-1, two pushes: host buffers then device-resident buffers
    S1 = api.SumOfSharedHashes(R, top=1, max_batch_reads=n_reads, max_batch_bases=len(bases))
    half = n_reads // 2
    a = S1.push(bases, offsets[:half + 1])
    d_b, d_o = api.DeviceBuffer.from_numpy(bases), api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[half:]))
    d_i, d_s = api.DeviceBuffer((n_reads - half) * 4), api.DeviceBuffer((n_reads - half) * 8)
    S1.push_device(d_b.ptr, d_o.ptr, n_reads - half, int(offsets[-1] - offsets[half]), d_i.ptr, d_s.ptr)
    S1.sync()
    idx = np.concatenate([a["topk_idx"][:, 0], d_i.to_numpy(np.uint32, (n_reads - half,))])
    sm = np.concatenate([a["topk_sum"][:, 0], d_s.to_numpy(np.uint64, (n_reads - half,))])
    assert np.array_equal(idx, exp["topk_idx"][:, 0]) and np.array_equal(sm, exp["topk_sum"][:, 0]), "top-1 rows"
    assert np.array_equal(S1.table(), exp["cum"]), "top-1 table"
    for d in (d_b, d_o, d_i, d_s):
        d.free()
# several chunks of 1024 reads, top-1 and top-2 (the pruned rankings: chunk-level bounds, counts in one or two levels), in batches
# that share passes and through single pushes
ref, bases, offsets = workload(1300, 300, 5000, read_len=300, rng_seed=9)
for top in (1, 2):
    exp = orc.stream(16, 0, 300, ref["ref"], np.full(1300, 300, np.uint32), bases, offsets, top_k=top)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=top, max_batch_reads=2200, max_batch_bases=len(bases))
    d_b = api.DeviceBuffer.from_numpy(bases)
    keep, rows = [d_b], []
    for a, b in ((0, 2100), (2100, 2200), (2200, 4300), (4300, 5000)):
        d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:b + 1]))
        d_i, d_s = api.DeviceBuffer((b - a) * top * 4), api.DeviceBuffer((b - a) * top * 8)
        keep += [d_o, d_i, d_s]
        rows.append((b - a, d_i, d_s))
        S.enqueue_device(d_b.ptr, d_o.ptr, b - a, int(offsets[b] - offsets[a]), d_i.ptr, d_s.ptr)
    S.sync()
    idx = np.concatenate([d_i.to_numpy(np.uint32, (m, top)) for m, d_i, _ in rows])
    val = np.concatenate([d_s.to_numpy(np.uint64, (m, top)) for m, _, d_s in rows])
    assert np.array_equal(idx, exp["topk_idx"]) and np.array_equal(val, exp["topk_sum"]), "rows of the chunked stream, top %d" % top
    assert np.array_equal(S.table(), exp["cum"]), "table of the chunked stream"
    S.reset()
    got = S.push(bases, offsets[:2001])
    assert np.array_equal(got["topk_idx"], exp["topk_idx"][:2000]) and np.array_equal(got["topk_sum"], exp["topk_sum"][:2000]), "push"
    for d in keep:
        d.free()
print("variant ok", os.environ.get("SKX_SCAN_SPLIT"), os.environ.get("SKX_SCAN_BIG"))
