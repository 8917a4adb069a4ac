#!/usr/bin/env python3
"""tools/diag_snp.py (GPU box): the C2 stream on SURVEY 8(d)'s generator (SNP clone tree, reads from one truth strain) next to the
bench's ancestor workload: stats of the stream (pairs per read, |Q|, shared / un-shared groups), candidates a batch-level test
would leave, reads/s from a fresh table (K batches enqueued back to back, median of a few repetitions)."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketchy_amd import api, synth
B, L, K = 98304, 1500, int(os.environ.get("K", "20"))
N = int(os.environ.get("N", "40000"))
modes = sys.argv[1:] or ["snp", "pool"]
for mode in modes:
    t0 = time.time()
    ref = synth.make_reference(N, 10000, k=16, hash_seed=0, rng_seed=1, device="cuda:0", mode=mode)
    src = ref["truth_genome"] if mode == "snp" else ref["genome"]
    g = torch.from_numpy(src).to("cuda:0")
    print(f"[{mode}] reference in {time.time() - t0:.1f} s", flush=True)
    R = api.ReferenceSketch([ref["ref"]], [ref["col_len"]], k=16, seed=0, device=0)
    print(f"[{mode}] rare-hash index: {R.rare_index}", flush=True)
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=B * L)
    ti = torch.zeros((B, 1), dtype=torch.int32, device="cuda:0"); ts = torch.zeros((B, 1), dtype=torch.int64, device="cuda:0")
    batches = [synth.make_reads_torch(g, B, L, err=0.05, rng_seed=1000 + i, device="cuda:0") for i in range(K)]
    torch.cuda.synchronize()
    prev = None
    for rep in range(4):
        S.reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        call_ms = []
        for i, (b, o) in enumerate(batches):
            tq = time.perf_counter()
            S.enqueue_device(b.data_ptr(), o.data_ptr(), B, B * L, ti.data_ptr(), ts.data_ptr())
            call_ms.append(round(1e3 * (time.perf_counter() - tq), 2))
            if rep == 0 and i in (0, 1, 2, 3, 7, 15):
                S.sync()
                t = S.table().astype(np.int64)
                st = S.stats()
                line = f"  after batch {i}: pairs/read {st['last_pairs'] / B / max(1, 1 if st['passes_shared'] == 0 else 1):.2f} |Q| {st['dictionary_size']} stats {st}"
                if prev is not None:
                    lead0 = prev.max()
                    cand = int((t >= lead0).sum())
                    line += f"\n     candidates of this batch (end value >= leader's start {lead0}): {cand}"
                print(line, flush=True)
                prev = t
        S.sync()
        dt = time.perf_counter() - t1
        if rep == 2:
            print(f"[{mode}] host ms per enqueue call:", call_ms, flush=True)
        if rep:
            print(f"[{mode}] rep {rep}: {K * B / dt / 1e6:.1f} M reads/s ({1e3 * dt / K:.3f} ms per batch)", flush=True)
    S.set_profiling(1)
    S.profile()
    n_prof = 8
    for i in range(n_prof):
        b, o = batches[i % K]
        S.enqueue_device(b.data_ptr(), o.data_ptr(), B, B * L, ti.data_ptr(), ts.data_ptr())
    S.sync()
    print(f"[{mode}] stage ms per batch:", {k: round(v["ms"] / n_prof, 3) for k, v in S.profile().items()}, flush=True)
    S.set_profiling(False)
    t = S.table().astype(np.int64)
    order = np.argsort(-t)
    print(f"[{mode}] leader {order[0]} (truth {ref.get('truth_index')}) sums {t[order[:4]].tolist()} stats {S.stats()}", flush=True)
    S.close(); R.close()
    del batches, g
