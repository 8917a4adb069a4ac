#!/usr/bin/env python3
"""tools/pin_inputs.py <dir> -- the golden INPUT files of tools/pin_from_reference.sh (seeded; no reference code):
genomes/*.fa (a small two-level clone tree of SNP variants of one random ancestor), genomes.txt (their paths, in
order), reads.fq (reads of one strain, 3 % substitutions, both strands), reads_edge.fq (the edge cases the parity
tests hold: lower case, N, IUPAC codes, U, '-', '.', shorter than k, exactly k, a read with no valid k-mer, a long
read), genotypes.tsv (id + three feature columns, one row per genome, in sketch order)."""
import os
import sys

import numpy as np

out = sys.argv[1]
os.makedirs(os.path.join(out, "genomes"), exist_ok=True)
rng = np.random.default_rng(20261002)
ALPHA = np.frombuffer(b"ACGT", np.uint8)
G = 60000
ancestor = ALPHA[rng.integers(0, 4, G)]


def mutate(seq, rate):
    seq = seq.copy()
    hit = rng.random(len(seq)) < rate
    code = np.searchsorted(ALPHA, seq[hit])
    seq[hit] = ALPHA[(code + rng.integers(1, 4, int(hit.sum()))) % 4]
    return seq


def wrap(b, w=70):
    return b"\n".join(b[i:i + w] for i in range(0, len(b), w))


names, genomes = [], []
for lin in range(6):
    base = mutate(ancestor, 0.01)
    for st in range(4):
        g = mutate(base, 0.0005)
        nm = f"lin{lin}_strain{st}.fa"
        with open(os.path.join(out, "genomes", nm), "wb") as f:
            f.write(b">" + nm.encode() + b" synthetic\n" + wrap(g.tobytes()) + b"\n")
        names.append(nm)
        genomes.append(g)
with open(os.path.join(out, "genomes.txt"), "w") as f:
    f.write(" ".join(os.path.join(out, "genomes", n) for n in names) + "\n")
with open(os.path.join(out, "genotypes.tsv"), "w") as f:
    f.write("id\tmlst\tmeca\tpvl\n")
    for i, nm in enumerate(names):
        f.write(f"{nm}\tST{i // 4}\t{'R' if i % 3 else 'S'}\t{'+' if i % 5 == 0 else '-'}\n")

comp = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    comp[a] = b
truth = genomes[9]
with open(os.path.join(out, "reads.fq"), "wb") as f:
    for i in range(300):
        L = int(np.clip(np.exp(rng.normal(np.log(1500), 0.6)), 200, 20000))
        a = int(rng.integers(0, G - L))
        r = mutate(truth[a:a + L], 0.03)
        if rng.integers(0, 2):
            r = comp[r[::-1]]
        f.write(b"@read%d\n" % i + r.tobytes() + b"\n+\n" + b"I" * L + b"\n")

g = truth.tobytes()
edge = [g[100:104], g[100:115], g[100:116], g[1000:1400].lower(), g[2000:2200] + b"N" + g[2201:2500],
        g[4000:4300].replace(b"T", b"U"), b"RYKMSWBDHV" * 20, b"A" * 500, b"ACGTACGTACGTACGTACGT" * 10,
        g[7000:7300] + b"-" + g[7301:7600] + b"." + g[7601:7700], g[10000:45000], g[5000:7063]]
with open(os.path.join(out, "reads_edge.fq"), "wb") as f:
    for i, r in enumerate(edge):
        f.write(b"@edge%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
print(f"{len(names)} genomes, 300 + {len(edge)} reads -> {out}")
