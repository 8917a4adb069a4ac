"""Run by tests/test_gpu_enqueue.py in a subprocess with library knobs forced through the environment (read once per
process): SKX_SPEC_INSERT=0 (pair gather on the scan stream), SKX_PIPELINE (stream depth), SKX_PASS_READS (several passes
per batch).  A stream of uneven batches through skx_stream_enqueue_device against the oracle, top-1 and top-3."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import workload  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from sketchy_amd import api  # noqa: E402

for n, s, n_reads, top, seed in ((600, 300, 1300, 1, 5), (130, 500, 700, 3, 6)):
    ref, bases, offsets = workload(n, s, n_reads, read_len=400, rng_seed=seed)
    exp = orc.stream(16, 0, s, ref["ref"], np.full(n, s, np.uint32), bases, offsets, top_k=top)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=top, max_batch_reads=500, max_batch_bases=len(bases))
    cuts = [0, 300, 301, 800, n_reads] if n_reads > 800 else [0, 10, 500, n_reads]
    d_b = api.DeviceBuffer.from_numpy(bases)
    keep, rows = [d_b], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(offsets[a:b + 1]))
        d_i, d_s = api.DeviceBuffer((b - a) * top * 4), api.DeviceBuffer((b - a) * top * 8)
        keep += [d_o, d_i, d_s]
        rows.append((b - a, d_i, d_s))
        S.enqueue_device(d_b.ptr, d_o.ptr, b - a, int(offsets[b] - offsets[a]), d_i.ptr, d_s.ptr)
    S.sync()
    idx = np.concatenate([d_i.to_numpy(np.uint32, (m, top)) for m, d_i, _ in rows])
    val = np.concatenate([d_s.to_numpy(np.uint64, (m, top)) for m, _, d_s in rows])
    assert np.array_equal(idx, exp["topk_idx"]), "rows idx"
    assert np.array_equal(val, exp["topk_sum"]), "rows sum"
    assert np.array_equal(S.table(), exp["cum"]), "table"
    if os.environ.get("SKX_PASS_READS"):
        assert S.stats()["last_passes"] > 1
    for d in keep:
        d.free()
print("enqueue_check ok")
