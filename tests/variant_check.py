"""Run by tests/test_gpu_fullsize.py in a subprocess with SKX_SCAN_SPLIT / SKX_SCAN_BIG forced (the library reads
them once per process): parity of the forced scan-kernel variant against the oracle on small and ragged inputs."""
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
print("variant ok", os.environ.get("SKX_SCAN_SPLIT"), os.environ.get("SKX_SCAN_BIG"))
