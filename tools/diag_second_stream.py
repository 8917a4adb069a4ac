#!/usr/bin/env python3
"""tools/diag_second_stream.py (GPU box): does a SECOND stream on the same reference run as fast as the first one?
C2-shaped workload; streams created one after the other, timed alone, with the other alive / closed, with stage profiling."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sketchy_amd import api, synth  # noqa: E402

B, K = 98304, 20
tdev = "cuda:0"
ref = synth.make_reference(40000, 10000, k=16, hash_seed=0, rng_seed=1, device=tdev)
genome_t = torch.from_numpy(ref["genome"]).to(tdev)
batches = [synth.make_reads_torch(genome_t, B, 1500, err=0.05, rng_seed=1000 + i, lognormal_sigma=0.0, device=tdev) for i in range(8)]
nb = [int(o[-1].item()) for _, o in batches]
R = api.ReferenceSketch([ref["ref"]], [ref["col_len"]], k=16, seed=0, device=0)
d_ti = torch.zeros((B, 1), dtype=torch.int32, device=tdev)
d_ts = torch.zeros((B, 1), dtype=torch.int64, device=tdev)
co = int(sys.argv[1]) if len(sys.argv) > 1 else 1
api.set_option("stream_coalesce", co)


def make():
    return api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=max(nb))


def run(S, label, prof=False):
    for i in range(3):
        b, o = batches[i % 8]
        S.enqueue_device(b.data_ptr(), o.data_ptr(), B, nb[i % 8], d_ti.data_ptr(), d_ts.data_ptr())
    S.sync()
    ts = []
    if prof:
        S.profile(); S.set_profiling(1)
    for rep in range(4):
        S.reset()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(K):
            b, o = batches[i % 8]
            S.enqueue_device(b.data_ptr(), o.data_ptr(), B, nb[i % 8], d_ti.data_ptr(), d_ts.data_ptr())
        S.sync()
        ts.append(time.perf_counter() - t)
    extra = ""
    if prof:
        p = S.profile(); S.set_profiling(0)
        extra = " ".join(f"{n}={v['ms'] / max(1, v['launches']):.3f}" for n, v in p.items() if v["launches"])
    print(f"{label}: {K * B / float(np.median(ts)) / 1e6:.1f} M reads/s  ({', '.join('%.1f' % (K * B / t / 1e6) for t in ts)}) {extra}", flush=True)


A = make()
run(A, "A (first stream)")
Bs = make()
run(Bs, "B (second, A alive)")
run(A, "A again (B alive)")
run(Bs, "B again")
run(Bs, "B profiled", prof=True)
run(A, "A profiled", prof=True)
A.close()
run(Bs, "B (A closed)")
C = make()
run(C, "C (created after A closed)")
print("free", api.device_mem(0))
