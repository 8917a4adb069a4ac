#!/usr/bin/env python3
"""tools/diag_dead.py (GPU box): how far behind the leader are the rank groups of the C2 workload?  For a group to be skipped for a
whole batch / pass its best genome must stay below the leader even after the batch's / pass's whole gain: rate_g / rate_leader <
n / (n + N) after n reads with N reads in the batch / pass.  Prints the distribution of the groups' best rates after 24 batches."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchy_amd import api, synth
B, L = 98304, 1500
ref = synth.make_reference(40000, 10000, k=16, hash_seed=0, rng_seed=1, device="cuda:0", shuffle=True, n_lineages=0)
g = torch.from_numpy(ref["genome"]).to("cuda:0")
R = api.ReferenceSketch([ref["ref"]], [ref["col_len"]], k=16, seed=0, device=0)
S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=B * L)
ti = torch.zeros((B, 1), dtype=torch.int32, device="cuda:0"); ts = torch.zeros((B, 1), dtype=torch.int64, device="cuda:0")
prev = None
for i in range(24):
    b, o = synth.make_reads_torch(g, B, L, err=0.05, rng_seed=1000 + i, lognormal_sigma=0.0, device="cuda:0")
    S.enqueue_device(b.data_ptr(), o.data_ptr(), B, int(o[-1].item()), ti.data_ptr(), ts.data_ptr())
    if i in (7, 15, 23):
        S.sync()
        t = S.table().astype(np.float64)
        n = (i + 1) * B
        lead = t.max()
        pad = (-len(t)) % 512
        grp = np.concatenate([t, np.zeros(pad)]).reshape(-1, 512).max(axis=1) / lead
        gain = None if prev is None else (t - prev)
        print(f"after {n} reads: leader {lead:.0f}; groups {len(grp)}; best rate of a group / leader's: min {grp.min():.3f} median {np.median(grp):.3f}; "
              f"groups below 0.99: {(grp < 0.99).sum()}, below 0.95: {(grp < 0.95).sum()}, below 0.9: {(grp < 0.9).sum()}, below 0.72: {(grp < 0.72).sum()}")
        for N, what in ((B, "batch"), (8 * B, "pass of 8")):
            thr = n / (n + N)
            print(f"   dead for a whole {what} (ratio < {thr:.3f}): {(grp < thr).sum()} of {len(grp)} groups")
        prev = t
