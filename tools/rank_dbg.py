import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sketchy_amd import api, synth, _lib
tdev = torch.device("cuda:0")
ref = synth.make_reference(40000, 10000, k=16, hash_seed=0, rng_seed=1, device=tdev, shuffle=True)
g = torch.from_numpy(ref["genome"]).to(tdev)
B = 98304
R = api.ReferenceSketch([ref["ref"]], [ref["col_len"]], k=16, seed=0, device=0)
S = api.SumOfSharedHashes(R, top=1, max_batch_reads=B, max_batch_bases=B * 1600)
lib = _lib.load()
lib.skx_debug_rank_counters.argtypes = [C.c_void_p, C.c_int]
out = (C.c_uint64 * 8)()
d_ti = torch.zeros((B, 1), dtype=torch.int32, device=tdev); d_ts = torch.zeros((B, 1), dtype=torch.int64, device=tdev)
print("step waves chunk_pruned no_cand replayed words/replayed cand/replayed pairs/replayed")
for i in range(12):
    bases, offs = synth.make_reads_torch(g, B, 1500, err=0.05, rng_seed=1000 + i, lognormal_sigma=0.0, device=tdev)
    S.push_device(bases.data_ptr(), offs.data_ptr(), B, int(offs[-1].item()), d_ti.data_ptr(), d_ts.data_ptr())
    S.sync()
    lib.skx_debug_rank_counters(out, 1)
    v = [int(x) for x in out]
    rp = max(v[3], 1)
    print(i, v[0], v[1], v[2], v[3], round(v[4] / rp, 2), round(v[5] / rp, 1), round(v[6] / rp, 1))
