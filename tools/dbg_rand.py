import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np
from helpers import pack_reads
from oracle import oracle as orc
from sketchy_amd import synth, api
rng = np.random.default_rng(2024)
alphabet = np.frombuffer(b"ACGTACGTACGTACGTacgtNRYKM-\n ", np.uint8)
for case in range(40):
    n = int(rng.integers(1, 700)); s = int(rng.choice([1, 7, 64, 200, 513])); k = int(rng.choice([16, 16, 16, 11, 21, 32])); seed = int(rng.choice([0, 0, 42, 7]))
    ref = synth.make_reference(n, s, k=k, hash_seed=seed, genome_len=max(3000, 40 * s), rng_seed=1000 + case, device="numpy")
    hashes = ref["ref"].copy(); col_len = np.full(n, s, np.uint32)
    if n > 3 and rng.random() < 0.5: hashes[rng.integers(0, n)] = hashes[rng.integers(0, n)]
    if rng.random() < 0.5: col_len = rng.integers(0, s + 1, size=n).astype(np.uint32)
    n_reads = int(rng.integers(1, 200))
    bases, offsets = synth.make_reads(ref["genome"], n_reads, int(rng.choice([30, 150, 600])), err=0.03, rng_seed=5000 + case, lognormal_sigma=0.7, min_len=0, max_len=4000)
    reads = [bytearray(bases[int(offsets[i]):int(offsets[i + 1])].tobytes()) for i in range(n_reads)]
    for r in reads:
        if len(r) and rng.random() < 0.3:
            for _ in range(int(rng.integers(1, 6))): r[int(rng.integers(0, len(r)))] = int(alphabet[rng.integers(0, len(alphabet))])
    if rng.random() < 0.3: reads[int(rng.integers(0, n_reads))] = bytearray(b"")
    bases, offsets = pack_reads([bytes(r) for r in reads])
    top = int(rng.integers(0, min(n, 20) + 1)); batches = int(rng.integers(1, 4)); ws = bool(rng.random() < 0.3); wk = bool(rng.random() < 0.3)
    exp = orc.stream(k, seed, s, hashes, col_len, bases, offsets, top_k=max(top, 1), want_shared=True, want_sketches=True)
    R = api.ReferenceSketch(hashes, col_len, k=k, seed=seed)
    S = api.SumOfSharedHashes(R, top=0, max_batch_reads=n_reads, max_batch_bases=max(1, len(bases)))
    got = S.push(bases, offsets, want_shared=True)   # production sketch path + debug shared
    bad = np.argwhere(got["shared"] != exp["shared"])
    if len(bad):
        r = bad[0][0]
        print("case", case, dict(n=n, s=s, k=k, seed=seed, n_reads=n_reads, top=top, batches=batches), "first bad read", r, "len", int(offsets[r+1]-offsets[r]), "n bad reads", len(set(bad[:,0])))
        print("read:", bytes(reads[r])[:200])
        print("got", got["shared"][r][:12], "exp", exp["shared"][r][:12])
        # with sketches (non-inrange path)
        S2 = api.SumOfSharedHashes(R, top=0, max_batch_reads=n_reads, max_batch_bases=max(1, len(bases)))
        g2 = S2.push(bases, offsets, want_shared=True, want_sketches=True)
        print("full-sketch path equal:", np.array_equal(g2["shared"], exp["shared"]), "sketches equal:", np.array_equal(g2["sketches"], exp["sketches"]))
        break
else:
    print("no mismatch in shared; production path fine")
