#!/usr/bin/env python3
"""Diagnostic (GPU box): which species' rank groups receive bits when the sample belongs to species 0?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sketchy_amd import api, synth
species = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2048,2048,1536,1024,512").split(",")]
s = 10000
tdev = "cuda:0"
refs = [synth.make_reference(n, s, rng_seed=1 + i, device=tdev) for i, n in enumerate(species)]
g = torch.from_numpy(refs[0]["genome"]).to(tdev)
bases, offs = synth.make_reads_torch(g, 8192, 1500, err=0.05, rng_seed=1000, lognormal_sigma=1.0, device=tdev)
nb = int(offs[-1].item())
R = api.ReferenceSketch([r["ref"] for r in refs], [r["col_len"] for r in refs])
S = api.SumOfSharedHashes(R, top=1, max_batch_reads=8192, max_batch_bases=nb)
S.push_device(bases.data_ptr(), offs.data_ptr(), 8192, nb, None, None)
S.sync()
t = S.table()
print("stats", S.stats())
a = 0
for i, n in enumerate(species):
    part = t[a:a + n]
    print(f"species {i}: genomes {n} nonzero {int((part > 0).sum())} max {int(part.max())} sum {int(part.sum())}")
    a += n
# which hashes do species 0 and 1 share?
u0, u1 = np.unique(refs[0]["ref"]), np.unique(refs[1]["ref"])
print("distinct", len(u0), len(u1), "shared", len(np.intersect1d(u0, u1)))
