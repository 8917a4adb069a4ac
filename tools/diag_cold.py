#!/usr/bin/env python3
"""tools/diag_cold.py [n_cold] [ancestor|truth] (GPU box): the cold leg alone -- table reset, ONE enqueue of a C2 batch, sync -- for a kernel
trace of a lone first batch (rocprofv3 --kernel-trace --stats -- python3 tools/diag_cold.py)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sketchy_amd import api, synth  # noqa: E402

B = 98304
n_cold = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workload = sys.argv[2] if len(sys.argv) > 2 else "ancestor"
tdev = "cuda:0"
ref = synth.make_reference(40000, 10000, k=16, hash_seed=0, rng_seed=1, device=tdev, mode="snp" if workload == "truth" else "pool")
genome_t = torch.from_numpy(ref["truth_genome"] if workload == "truth" else ref["genome"]).to(tdev)
batches = [synth.make_reads_torch(genome_t, B, 1500, err=0.05, rng_seed=1000 + i, lognormal_sigma=0.0, device=tdev) for i in range(2)]
nb = [int(o[-1].item()) for _, o in batches]
R = api.ReferenceSketch([ref["ref"]], [ref["col_len"]], k=16, seed=0, device=0)
d_ti = torch.zeros((B, 1), dtype=torch.int32, device=tdev)
d_ts = torch.zeros((B, 1), dtype=torch.int64, device=tdev)
S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=max(nb))
ts = []
for i in range(n_cold + 2):
    b, o = batches[0]
    S.reset()
    torch.cuda.synchronize()
    t = time.perf_counter()
    S.enqueue_device(b.data_ptr(), o.data_ptr(), B, nb[0], d_ti.data_ptr(), d_ts.data_ptr())
    S.sync()
    ts.append(time.perf_counter() - t)
print("cold ms:", " ".join("%.3f" % (1e3 * t) for t in ts), " median %.3f" % (1e3 * float(np.median(ts[2:]))), flush=True)
S.set_profiling(1)
S.profile()
for i in range(4):
    S.reset()
    S.enqueue_device(batches[0][0].data_ptr(), batches[0][1].data_ptr(), B, nb[0], d_ti.data_ptr(), d_ts.data_ptr())
    S.sync()
p = S.profile()
print(" ".join(f"{n}={v['ms'] / max(1, v['launches']):.3f}" for n, v in p.items() if v["launches"]))

# experiments build only: where the ranking's waves went in a lone first batch, and in a batch far from the start
import ctypes as C
import os
if "exp" in os.environ.get("SKX_LIB_PATH", ""):
    from sketchy_amd import _lib
    L = _lib.load()
    L.skx_debug_rank_counters.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    L.skx_debug_rank_counters.restype = None
    buf = (C.c_ulonglong * 128)()

    def show(label):
        L.skx_debug_rank_counters(buf, 1)
        v = list(buf)
        print(label, "waves: chunk-dead %d, no live word %d, no candidate %d, few %d, replay %d; replay: cands/wave %.1f, pairs/wave %.1f; by cands (1|2-4|5-16|17-64|65+): %s"
              % (v[0], v[1], v[2], v[3], v[4], v[5] / max(1, v[4]), v[6] / max(1, v[4]), v[8:13]))
        print("   replaying waves per chunk:", v[16:16 + 96])
    S.set_profiling(0)
    S.reset(); S.sync(); L.skx_debug_rank_counters(buf, 1)
    S.enqueue_device(batches[0][0].data_ptr(), batches[0][1].data_ptr(), B, nb[0], d_ti.data_ptr(), d_ts.data_ptr())
    S.sync()
    show("first batch:")
    for i in range(6):
        S.enqueue_device(batches[(i + 1) % 2][0].data_ptr(), batches[(i + 1) % 2][1].data_ptr(), B, nb[(i + 1) % 2], d_ti.data_ptr(), d_ts.data_ptr())
        S.sync()
        show("batch %d:" % (i + 2))
