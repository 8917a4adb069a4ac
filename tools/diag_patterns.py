"""How well do the LONG genome lists of the rare-hash index compress as (pattern, exceptions)?  (CPU, numpy)

A scaled-down SURVEY 8(d) reference (same lineage size as C2: 200 strains per lineage, s = 10 000): every distinct hash's genome
list; the lists of 9 .. rare_max genomes ("long") are grouped by a one-permutation MinHash signature of the list (two lists of
Jaccard similarity J share it with probability J), the most frequent exact list of a group is its PATTERN, and every list is written as its
group's pattern plus the symmetric difference.  Prints the distribution of the exceptions per list.

    python tools/diag_patterns.py [n_genomes=4000] [n_lineages=20] [s=10000]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sketchy_amd import synth  # noqa: E402


def mix32(g):
    x = g.astype(np.uint64)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    return (x ^ (x >> np.uint64(16))).astype(np.uint32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    s = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    rare_max = 1024
    t0 = time.time()
    ref = synth.make_reference_snp(n, s, n_lineages=nl, device="cpu")
    print(f"reference {n} x {s} in {time.time() - t0:.1f} s")
    h = ref["ref"].reshape(-1)
    g = np.repeat(np.arange(n, dtype=np.uint32), s)
    order = np.argsort(h, kind="stable")
    h, g = h[order], g[order]
    start = np.flatnonzero(np.r_[True, h[1:] != h[:-1]])
    cnt = np.diff(np.r_[start, len(h)])
    print(f"distinct hashes {len(start)}; held by 1: {(cnt == 1).sum()}, 2..8: {((cnt > 1) & (cnt <= 8)).sum()}, "
          f"9..{rare_max}: {((cnt > 8) & (cnt <= rare_max)).sum()}, more: {(cnt > rare_max).sum()}")
    long_ix = np.flatnonzero((cnt > 8) & (cnt <= rare_max))
    print("long lists:", len(long_ix), "entries", int(cnt[long_ix].sum()), "length quantiles", np.quantile(cnt[long_ix], [0, .1, .5, .9, 1]))
    mg = mix32(g)
    sig = np.minimum.reduceat(mg, start)[long_ix]
    # groups by signature; pattern = the MOST FREQUENT exact list of the group (content hash = sum of mixed genome ids)
    m64 = (mg.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (g.astype(np.uint64) << np.uint64(32))
    content = np.add.reduceat(m64, start)[long_ix]
    us, inv = np.unique(sig, return_inverse=True)
    best = np.zeros(len(us), np.int64) - 1
    from collections import Counter
    per_group = [Counter() for _ in us]
    first_of = {}
    for i, li in enumerate(long_ix):
        per_group[inv[i]][int(content[i])] += 1
        first_of.setdefault((int(inv[i]), int(content[i])), li)
    for gi, c in enumerate(per_group):
        best[gi] = first_of[(gi, c.most_common(1)[0][0])]
    print("signature groups:", len(us), "group sizes quantiles", np.quantile(np.bincount(inv), [0, .5, .9, 1]))
    exc = np.zeros(len(long_ix), np.int64)
    sets = {}
    for i, li in enumerate(long_ix):
        p = best[inv[i]]
        if p not in sets:
            sets[p] = set(g[start[p]:start[p] + cnt[p]].tolist())
        mine = set(g[start[li]:start[li] + cnt[li]].tolist())
        exc[i] = len(mine ^ sets[p])
    print("exceptions per long list: mean %.2f" % exc.mean(), "quantiles", np.quantile(exc, [0, .5, .9, .99, .999, 1]))
    for cap in (4, 6, 8, 14, 30):
        print(f"  lists with more than {cap} exceptions: {(exc > cap).sum()} ({100.0 * (exc > cap).mean():.2f} %)")
    print("distinct patterns:", len(sets))
    # ... and as a UNION of up to four disjoint patterns + exceptions: every genome votes for the smallest pattern that holds it
    pats = sorted(sets.items(), key=lambda kv: -len(kv[1]))
    gpat = np.full(n, -1, np.int64)
    for pi, (_, members) in enumerate(pats):
        gpat[list(members)] = pi          # (smaller patterns later: they overwrite)
    psize = np.array([len(m) for _, m in pats])
    pset = [m for _, m in pats]
    exc2 = np.zeros(len(long_ix), np.int64)
    npat = np.zeros(len(long_ix), np.int64)
    for i, li in enumerate(long_ix):
        mine = g[start[li]:start[li] + cnt[li]]
        votes = np.bincount(gpat[mine][gpat[mine] >= 0], minlength=len(pats))
        sel = [pi for pi in np.flatnonzero(votes * 2 >= psize) if votes[pi] > 0][:4]
        u = set()
        for pi in sel:
            u |= pset[pi]
        exc2[i] = len(set(mine.tolist()) ^ u)
        npat[i] = len(sel)
    print("as unions of <= 4 patterns: exceptions mean %.2f" % exc2.mean(), "quantiles", np.quantile(exc2, [0, .5, .9, .99, .999, 1]),
          "patterns per list", np.bincount(npat))
    for cap in (8, 11, 14):
        print(f"  lists with more than {cap} exceptions: {(exc2 > cap).sum()} ({100.0 * (exc2 > cap).mean():.2f} %)")


if __name__ == "__main__":
    main()
