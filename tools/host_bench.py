#!/usr/bin/env python3
"""End-to-end rate of the C++ host (`sketchy-hip predict --stream`) on the GPU box: FASTQ file -> parser -> batches
-> C ABI -> rows on stdout.  usage: python tools/host_bench.py [n_genomes s n_reads batch]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sketchy_amd import synth  # noqa: E402
from mshio import write_msh    # noqa: E402

n_genomes, s, n_reads, batch = (int(x) for x in (sys.argv[1:5] + ["5000", "1000", "1000000", "16384"][len(sys.argv) - 1:]))
d = tempfile.mkdtemp(prefix="skx_host_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t0 = time.time()
ref = synth.make_reference(n_genomes, s, rng_seed=1, device="numpy")
bases, offsets = synth.make_reads(ref["genome"], n_reads, 1500, err=0.05, rng_seed=7)
names = [f"genome{i:05d}.fa" for i in range(n_genomes)]
write_msh(d + "/ref.msh", names, ref["ref"])
with open(d + "/geno.tsv", "w") as f:
    f.write("id\tmlst\tmeca\n" + "".join(f"{n}\tST{i % 97}\t{'R' if i % 3 else 'S'}\n" for i, n in enumerate(names)))
raw = bases.tobytes()
with open(d + "/reads.fq", "wb") as f:
    for i in range(n_reads):
        a, b = int(offsets[i]), int(offsets[i + 1])
        f.write(b"@r%d\n" % i + raw[a:b] + b"\n+\n" + b"I" * (b - a) + b"\n")
print(f"setup {time.time() - t0:.1f} s; fastq {os.path.getsize(d + '/reads.fq') / 1e6:.0f} MB", flush=True)
exe = os.path.join(ROOT, "sketchy_amd", "sketchy-hip")


def run(limit, b_):
    t1 = time.time()
    with open(d + "/out.tsv", "wb") as out:
        rc = subprocess.run([exe, "predict", "-r", d + "/ref.msh", "-g", d + "/geno.tsv", "-i", d + "/reads.fq", "-s", "-b", str(b_)]
                            + (["-l", str(limit)] if limit else []), stdout=out).returncode
    dt = time.time() - t1
    rows = sum(1 for _ in open(d + "/out.tsv", "rb"))
    assert rc == 0 and rows == (limit or n_reads), (rc, rows)
    return dt


for b_ in (batch, 4096):
    part = n_reads // 5
    t_part = min(run(part, b_) for _ in range(2))
    t_full = min(run(0, b_) for _ in range(2))
    print(f"batch={b_}: {part} reads {t_part:.2f} s, {n_reads} reads {t_full:.2f} s  ->  steady {(n_reads - part) / (t_full - t_part):,.0f} reads/s "
          f"(fixed start-up {t_part - part * (t_full - t_part) / (n_reads - part):.2f} s: HIP init, .msh load, reference upload)", flush=True)
for f_ in os.listdir(d):
    os.remove(os.path.join(d, f_))
os.rmdir(d)
