"""Seeded synthetic workloads for the read-vs-reference MinHash path.

Uniform-random reference hashes would make every shared count 0 and every rank a tie, so
the generator creates real sharing:

  1. one random ancestor genome G (uniform ACGT); its canonical k-mer MurmurHash3 values
     (same hashing as the path under test, restated here in vectorised numpy) give the
     pool of smallest genome hashes;
  2. a two-level clone tree: each lineage keeps a fraction ``p_lineage`` of the pool and
     replaces the rest by fresh uniform hashes (standing in for mutated k-mers, whose
     murmur values are uniform anyway); each strain does the same to its lineage with
     ``p_strain``; a strain's sketch is the bottom-s of its set;
  3. reads are sampled from G itself (random strand, substitution errors), so their
     k-mers genuinely hit the reference hashes.

``make_reference(..., mode="snp")`` is SURVEY.md 8(d)'s generator proper (items 1-3): the variants are SNP copies of
the ancestor (lineage divergence 1 %, strain divergence 0.05 %), a variant's sketch is the bottom-s of the REAL canonical
k-mer hashes of its genome (ancestor hashes - k-mers killed by its SNPs + k-mers created, computed incrementally: k
windows per SNP), and the reads are sampled from ONE truth strain's genome (``ref["truth_genome"]``), so that its
lineage, and within it the strain, accumulates fastest -- the regime of a real sample (/root/reference/README.md:38).

Nothing here imports oracle/: bench.py and the tests hand the same arrays to the HIP path
and to the oracle.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_C1 = np.uint64(0x87C37B91114253D5)
_C2 = np.uint64(0x4CF5AD432745937F)


def _rotl(x, r):
    return (x << np.uint64(r)) | (x >> np.uint64(64 - r))


def _fmix(k):
    k = k ^ (k >> np.uint64(33))
    k = k * np.uint64(0xFF51AFD7ED558CCD)
    k = k ^ (k >> np.uint64(33))
    k = k * np.uint64(0xC4CEB9FE1A85EC53)
    k = k ^ (k >> np.uint64(33))
    return k


def murmur3_h1_rows(rows: np.ndarray, seed: int) -> np.ndarray:
    """MurmurHash3_x64_128 h1 of every row of a [m, k] uint8 matrix (vectorised)."""
    rows = np.ascontiguousarray(rows, np.uint8)
    m, k = rows.shape
    with np.errstate(over="ignore"):
        h1 = np.full(m, seed, np.uint64)
        h2 = np.full(m, seed, np.uint64)

        def le64(cols):
            v = np.zeros(m, np.uint64)
            for i in range(cols.shape[1]):
                v |= cols[:, i].astype(np.uint64) << np.uint64(8 * i)
            return v

        nb = k // 16
        for b in range(nb):
            k1 = le64(rows[:, 16 * b:16 * b + 8])
            k2 = le64(rows[:, 16 * b + 8:16 * b + 16])
            k1 = k1 * _C1; k1 = _rotl(k1, 31); k1 = k1 * _C2; h1 = h1 ^ k1
            h1 = _rotl(h1, 27); h1 = h1 + h2; h1 = h1 * np.uint64(5) + np.uint64(0x52DCE729)
            k2 = k2 * _C2; k2 = _rotl(k2, 33); k2 = k2 * _C1; h2 = h2 ^ k2
            h2 = _rotl(h2, 31); h2 = h2 + h1; h2 = h2 * np.uint64(5) + np.uint64(0x38495AB5)
        t = k - 16 * nb
        if t > 8:
            k2 = le64(rows[:, 16 * nb + 8:])
            k2 = k2 * _C2; k2 = _rotl(k2, 33); k2 = k2 * _C1; h2 = h2 ^ k2
        if t > 0:
            k1 = le64(rows[:, 16 * nb:16 * nb + min(t, 8)])
            k1 = k1 * _C1; k1 = _rotl(k1, 31); k1 = k1 * _C2; h1 = h1 ^ k1
        h1 = h1 ^ np.uint64(k); h2 = h2 ^ np.uint64(k)
        h1 = h1 + h2; h2 = h2 + h1
        h1 = _fmix(h1); h2 = _fmix(h2)
        h1 = h1 + h2
    return h1


_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


def canonical_kmer_hashes(seq: np.ndarray, k: int, seed: int, chunk: int = 1 << 20) -> np.ndarray:
    """Hashes of the canonical k-mers of an upper-case ACGT sequence (position order)."""
    seq = np.ascontiguousarray(seq, np.uint8)
    m = len(seq) - k + 1
    if m <= 0:
        return np.zeros(0, np.uint64)
    out = np.empty(m, np.uint64)
    for lo in range(0, m, chunk):
        hi = min(m, lo + chunk)
        win = np.lib.stride_tricks.sliding_window_view(seq[lo:hi + k - 1], k)
        rc = _COMP[win[:, ::-1]]
        # bytewise lexicographic fwd < rc ?
        diff = win != rc
        first = np.argmax(diff, axis=1)
        rows = np.arange(hi - lo)
        fwd_less = diff[rows, first] & (win[rows, first] < rc[rows, first])
        canon = np.where(fwd_less[:, None], win, rc)
        out[lo:hi] = murmur3_h1_rows(canon, seed)
    return out


def random_genome(length: int, rng: np.random.Generator) -> np.ndarray:
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=length)]


def make_reference(n_genomes: int, s: int, k: int = 16, hash_seed: int = 0, genome_len: int = 0,
                   n_lineages: int = 0, p_lineage: float = 0.85, p_strain: float = 0.99,
                   rng_seed: int = 1, shuffle: bool = True, device: str = "auto", mode: str = "pool", **snp_kw):
    """Returns dict(genome [uint8], ref [n_genomes, s] uint64 ascending rows, col_len).  mode="snp": see
    make_reference_snp (adds truth_index / truth_genome / lineage)."""
    if mode == "snp":
        return make_reference_snp(n_genomes, s, k=k, hash_seed=hash_seed, genome_len=genome_len, n_lineages=n_lineages,
                                  rng_seed=rng_seed, shuffle=shuffle, device=device, **snp_kw)
    assert mode == "pool" and not snp_kw
    rng = np.random.default_rng(rng_seed)
    if genome_len <= 0:
        genome_len = max(20000, 280 * s)  # keeps (bottom-s range)/(hash space) near real data
    if n_lineages <= 0:
        n_lineages = max(1, int(round(n_genomes ** 0.5)))
    genome = random_genome(genome_len, rng)
    hg = np.unique(canonical_kmer_hashes(genome, k, hash_seed))
    pool_n = int(np.ceil(1.35 * s)) + 8
    if len(hg) < pool_n:
        raise ValueError("genome too short for requested sketch size")
    pool = hg[:pool_n]
    pool_max = int(pool[-1])
    per_lin = -(-n_genomes // n_lineages)

    use_torch = False
    if device != "numpy":
        try:
            import torch
            if device == "auto":
                device = "cuda" if torch.cuda.is_available() else "numpy"
            use_torch = device != "numpy"
        except ImportError:
            device = "numpy"

    ref = np.empty((n_genomes, s), np.uint64)
    done = 0
    if use_torch:
        import torch
        dev = torch.device(device)
        gen = torch.Generator(device=dev)
        gen.manual_seed(rng_seed * 7919 + 13)
        assert pool_max < (1 << 62)
        pool_t = torch.from_numpy(pool.astype(np.int64)).to(dev)
        for lin in range(n_lineages):
            n_here = min(per_lin, n_genomes - done)
            if n_here <= 0:
                break
            keep = torch.rand(pool_n, generator=gen, device=dev) < p_lineage
            fresh = (torch.rand(pool_n, generator=gen, device=dev, dtype=torch.float64) * pool_max).to(torch.int64)
            base = torch.where(keep, pool_t, fresh)
            keep2 = torch.rand((n_here, pool_n), generator=gen, device=dev) < p_strain
            fresh2 = (torch.rand((n_here, pool_n), generator=gen, device=dev, dtype=torch.float64) * pool_max).to(torch.int64)
            rows = torch.where(keep2, base[None, :], fresh2)
            rows, _ = torch.sort(rows, dim=1)
            ref[done:done + n_here] = _dedup_take(rows.cpu().numpy().view(np.uint64), s)
            done += n_here
    else:
        for lin in range(n_lineages):
            n_here = min(per_lin, n_genomes - done)
            if n_here <= 0:
                break
            keep = rng.random(pool_n) < p_lineage
            fresh = rng.integers(0, pool_max, size=pool_n, dtype=np.uint64)
            base = np.where(keep, pool, fresh)
            keep2 = rng.random((n_here, pool_n)) < p_strain
            fresh2 = rng.integers(0, pool_max, size=(n_here, pool_n), dtype=np.uint64)
            rows = np.sort(np.where(keep2, base[None, :], fresh2), axis=1)
            ref[done:done + n_here] = _dedup_take(rows, s)
            done += n_here
    if shuffle:
        ref = ref[rng.permutation(n_genomes)]
    col_len = np.full(n_genomes, s, np.uint32)
    return dict(genome=genome, ref=np.ascontiguousarray(ref), col_len=col_len, k=k, seed=hash_seed, s=s)


def _dedup_take(rows: np.ndarray, s: int) -> np.ndarray:
    """rows ascending per row; make each row strictly ascending (columns must be distinct
    hashes) by bumping the rare duplicate, then keep the first s.  (<=, not ==: a row that came back from the device
    sort out of order -- seen once, under the profiler -- is sorted again here instead of being refused by skx_ref_create)"""
    rows = rows.copy()
    dup = rows[:, 1:] <= rows[:, :-1]
    if dup.any():
        for r in np.nonzero(dup.any(axis=1))[0]:
            u = np.unique(rows[r])
            extra = rows[r].max() + np.arange(1, len(rows[r]) - len(u) + 1, dtype=np.uint64)
            rows[r] = np.concatenate([u, extra])
    return rows[:, :s]


def make_reads(genome: np.ndarray, n_reads: int, read_len=1500, err: float = 0.05, rng_seed: int = 2,
               lognormal_sigma: float = 0.0, min_len: int = 200, max_len: int = 50000):
    """Reads sampled from ``genome``: random start and strand, substitution errors.
    read_len fixed, or log-normal around read_len when lognormal_sigma > 0 (clipped).
    Returns (bases uint8 concatenated, offsets uint64 [n_reads+1])."""
    rng = np.random.default_rng(rng_seed)
    G = len(genome)
    if lognormal_sigma > 0:
        lens = np.exp(rng.normal(np.log(read_len), lognormal_sigma, size=n_reads))
        lens = np.clip(lens, min_len, min(max_len, G)).astype(np.int64)
    else:
        lens = np.full(n_reads, min(read_len, G), np.int64)
    offsets = np.zeros(n_reads + 1, np.uint64)
    offsets[1:] = np.cumsum(lens).astype(np.uint64)
    total = int(offsets[-1])
    starts = (rng.random(n_reads) * (G - lens + 1)).astype(np.int64)
    # gather indices for all reads at once
    read_id = np.repeat(np.arange(n_reads), lens)
    within = np.arange(total, dtype=np.int64) - np.repeat(offsets[:-1].astype(np.int64), lens)
    strand = rng.integers(0, 2, size=n_reads).astype(bool)
    rev = strand[read_id]
    pos = np.where(rev, starts[read_id] + lens[read_id] - 1 - within, starts[read_id] + within)
    bases = genome[pos]
    bases = np.where(rev, _COMP[bases], bases)
    if err > 0:
        hit = rng.random(total) < err
        nh = int(hit.sum())
        alphabet = np.frombuffer(b"ACGT", np.uint8)
        code = np.searchsorted(alphabet, bases[hit])  # A,C,G,T -> 0..3
        bases = bases.copy()
        bases[hit] = alphabet[(code + rng.integers(1, 4, size=nh)) % 4]
    return np.ascontiguousarray(bases, np.uint8), offsets


def make_reads_torch(genome, n_reads: int, read_len=1500, err: float = 0.05, rng_seed: int = 2,
                     lognormal_sigma: float = 0.0, min_len: int = 200, max_len: int = 50000, device="cuda"):
    """make_reads with torch on ``device`` (bench.py: batches are generated straight into HBM, one call per batch, so
    a run can hold many distinct batches without the host ever materialising them).  ``genome``: uint8 numpy array or
    torch tensor.  Returns (bases uint8 tensor, offsets int64 tensor [n_reads + 1]) on ``device``.  Same construction
    as make_reads (random start and strand, substitution errors) but a different random stream."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(rng_seed))
    g = genome if isinstance(genome, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(genome, np.uint8))
    g = g.to(dev)
    G = int(g.numel())
    if lognormal_sigma > 0:
        z = torch.randn(n_reads, generator=gen, device=dev, dtype=torch.float64)
        lens = torch.exp(z * lognormal_sigma + float(np.log(read_len)))
        lens = lens.clamp(min_len, min(max_len, G)).to(torch.int64)
    else:
        lens = torch.full((n_reads,), min(int(read_len), G), dtype=torch.int64, device=dev)
    offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(lens, 0)
    total = int(offsets[-1].item())
    starts = (torch.rand(n_reads, generator=gen, device=dev, dtype=torch.float64) * (G - lens + 1).to(torch.float64)).to(torch.int64)
    rev_read = torch.randint(0, 2, (n_reads,), generator=gen, device=dev, dtype=torch.int64).bool()
    read_id = torch.repeat_interleave(torch.arange(n_reads, device=dev), lens, output_size=total)
    within = torch.arange(total, device=dev, dtype=torch.int64) - offsets[:-1][read_id]
    rev = rev_read[read_id]
    pos = torch.where(rev, (starts + lens - 1)[read_id] - within, starts[read_id] + within)
    del within, read_id
    bases = g[pos]
    del pos
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    bases = torch.where(rev, comp[bases.long()], bases)
    del rev
    if err > 0:
        hit = torch.rand(total, generator=gen, device=dev) < err
        idx = hit.nonzero(as_tuple=True)[0]
        del hit
        code = torch.zeros(256, dtype=torch.int64, device=dev)
        for i, a in enumerate(b"ACGT"):
            code[a] = i
        alphabet = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        shift = torch.randint(1, 4, (idx.numel(),), generator=gen, device=dev, dtype=torch.int64)
        bases[idx] = alphabet[(code[bases[idx].long()] + shift) % 4]
    return bases.contiguous(), offsets


def mix_reads_torch(parts, rng_seed: int = 0):
    """One shuffled read stream out of several (bases, offsets) batches on the same device (make_reads_torch outputs): the
    reads of all parts in a random order -- a mixed-species sample.  Returns (bases, offsets)."""
    import torch
    dev = parts[0][0].device
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(rng_seed) * 2654435761 % (1 << 62) + 7)
    lens = torch.cat([o[1:] - o[:-1] for _, o in parts])
    base0, starts = 0, []
    for b, o in parts:
        starts.append(o[:-1] + base0)
        base0 += int(b.numel())
    starts = torch.cat(starts)
    allb = torch.cat([b for b, _ in parts])
    n = int(lens.numel())
    perm = torch.randperm(n, generator=gen, device=dev)
    nl = lens[perm]
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(nl, 0)
    total = int(offsets[-1].item())
    shift = torch.repeat_interleave(starts[perm] - offsets[:-1], nl, output_size=total)
    bases = allb[torch.arange(total, device=dev, dtype=torch.int64) + shift]
    return bases.contiguous(), offsets


# ----------------------------------------------------------------------------- SNP clone tree
_I64_MAX = (1 << 63) - 1


def _s64(c):
    """a 64-bit constant as the signed value torch's int64 holds"""
    return c - (1 << 64) if c >= (1 << 63) else c


def _t_shr(x, n):
    """logical shift right of int64 lanes"""
    return (x >> n) & ((1 << (64 - n)) - 1)


def _t_rotl(x, r):
    return (x << r) | _t_shr(x, 64 - r)


def _t_fmix(h):
    h = h ^ _t_shr(h, 33)
    h = h * _s64(0xFF51AFD7ED558CCD)
    h = h ^ _t_shr(h, 33)
    h = h * _s64(0xC4CEB9FE1A85EC53)
    return h ^ _t_shr(h, 33)


def torch_murmur3_h1(rows, seed: int):
    """MurmurHash3_x64_128 h1 of every row of a [m, k] uint8 tensor, as int64 lanes holding the unsigned value's bits
    (int64 multiplication wraps like u64; shifts are made logical).  Same function as murmur3_h1_rows."""
    import torch
    m, k = rows.shape
    c1, c2 = _s64(0x87C37B91114253D5), _s64(0x4CF5AD432745937F)
    h1 = torch.full((m,), _s64(seed & 0xFFFFFFFFFFFFFFFF), dtype=torch.int64, device=rows.device)
    h2 = h1.clone()

    def le64(cols):
        v = torch.zeros(m, dtype=torch.int64, device=rows.device)
        for i in range(cols.shape[1]):
            v |= cols[:, i].to(torch.int64) << (8 * i)
        return v

    nb = k // 16
    for b in range(nb):
        k1 = le64(rows[:, 16 * b:16 * b + 8])
        k2 = le64(rows[:, 16 * b + 8:16 * b + 16])
        k1 = k1 * c1; k1 = _t_rotl(k1, 31); k1 = k1 * c2; h1 = h1 ^ k1
        h1 = _t_rotl(h1, 27); h1 = h1 + h2; h1 = h1 * 5 + 0x52DCE729
        k2 = k2 * c2; k2 = _t_rotl(k2, 33); k2 = k2 * c1; h2 = h2 ^ k2
        h2 = _t_rotl(h2, 31); h2 = h2 + h1; h2 = h2 * 5 + 0x38495AB5
    t = k - 16 * nb
    if t > 8:
        k2 = le64(rows[:, 16 * nb + 8:])
        k2 = k2 * c2; k2 = _t_rotl(k2, 33); k2 = k2 * c1; h2 = h2 ^ k2
    if t > 0:
        k1 = le64(rows[:, 16 * nb:16 * nb + min(t, 8)])
        k1 = k1 * c1; k1 = _t_rotl(k1, 31); k1 = k1 * c2; h1 = h1 ^ k1
    h1 = h1 ^ k; h2 = h2 ^ k
    h1 = h1 + h2; h2 = h2 + h1
    h1 = _t_fmix(h1); h2 = _t_fmix(h2)
    return h1 + h2


def _t_window_hashes(codes, starts, k: int, seed: int, chunk: int = 1 << 21):
    """Canonical k-mer hashes (int64 bit patterns) of the windows codes[..., start : start + k].  codes: [L] or [J, L] uint8
    2-bit codes (0..3 = ACGT: the order of the ASCII letters, so comparing codes is the bytewise comparison needletail
    makes); starts: [n] or [J, n] int64.  canonical = forward if forward < reverse complement else reverse complement."""
    import torch
    dev = codes.device
    flat = codes.reshape(-1)
    if codes.dim() == 2:
        J, L = codes.shape
        base = (torch.arange(J, device=dev, dtype=torch.int64) * L)[:, None]
        st = (starts + base).reshape(-1)
    else:
        st = starts.reshape(-1)
    out = torch.empty(st.numel(), dtype=torch.int64, device=dev)
    ar = torch.arange(k, device=dev, dtype=torch.int64)
    ascii_ = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    kh = min(k, 16)
    wh = (4 ** torch.arange(kh - 1, -1, -1, device=dev, dtype=torch.int64))
    wl = (4 ** torch.arange(k - kh - 1, -1, -1, device=dev, dtype=torch.int64)) if k > kh else None
    for lo in range(0, st.numel(), chunk):
        a = st[lo:lo + chunk]
        win = flat[(a[:, None] + ar[None, :])]                     # [m, k] codes
        rc = (3 - win).flip(1)
        wf, wr = win.to(torch.int64), rc.to(torch.int64)
        fh, rh = (wf[:, :kh] * wh).sum(1), (wr[:, :kh] * wh).sum(1)
        less = fh < rh
        if wl is not None:
            fl, rl = (wf[:, kh:] * wl).sum(1), (wr[:, kh:] * wl).sum(1)
            less = less | ((fh == rh) & (fl < rl))
        canon = torch.where(less[:, None], win, rc)
        out[lo:lo + chunk] = torch_murmur3_h1(ascii_[canon.long()], seed)
    return out.reshape(starts.shape)


def _snp_alt(codes_at, pos, salt: int):
    """the base a SNP puts at `pos`: a function of (position, salt) only, so that a position drawn twice gets one value"""
    shift = ((pos * 2654435761 + salt * 40503 + 12345) >> 7) % 3 + 1
    return ((codes_at.to(pos.dtype) + shift) % 4).to(codes_at.dtype)


def make_reference_snp(n_genomes: int, s: int, k: int = 16, hash_seed: int = 0, genome_len: int = 0, n_lineages: int = 0,
                       div_lineage: float = 0.01, div_strain: float = 0.0005, rng_seed: int = 1, shuffle: bool = True,
                       device: str = "auto", truth: int = -1, strain_chunk_bytes: int = 1 << 30):
    """a two-level clone tree of SNP variants of one random ancestor.

    Every lineage is the ancestor with SNPs at ``div_lineage`` of its positions, every strain its lineage with further
    SNPs at ``div_strain``; row g of ``ref`` is the bottom-s of the canonical k-mer hashes of strain g's genome (exactly:
    positions whose window holds no SNP keep the parent's hash, the k windows over every SNP are hashed again; only hashes
    below a threshold a little above the ancestor's s-th smallest are tracked, and every row is checked to hold s of them).
    Returns dict(genome = the ancestor, ref, col_len, k, seed, s, lineage [n_genomes] = lineage of every row,
    truth_index = row of the truth strain, truth_genome = its genome (ASCII): sample the reads from THAT
    (``make_reads(ref["truth_genome"], ...)``, item 3).  truth = ordinal of the truth strain before shuffling
    (default: a strain in the middle of the tree).  torch on ``device`` ("auto": the GPU when there is one)."""
    import torch
    if device == "auto":
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if device == "numpy":
        device = "cpu"
    dev = torch.device(device)
    rng = np.random.default_rng(rng_seed)
    if genome_len <= 0:
        genome_len = max(20000, 280 * s)
    if n_lineages <= 0:
        n_lineages = max(1, int(round(n_genomes ** 0.5)))
    per_lin = -(-n_genomes // n_lineages)
    if truth < 0:
        truth = (n_lineages // 2) * per_lin + per_lin // 2
    truth = min(truth, n_genomes - 1)
    L = genome_len
    gen = torch.Generator(device=dev)
    gen.manual_seed(rng_seed * 7919 + 29)
    anc = torch.from_numpy(rng.integers(0, 4, size=L).astype(np.uint8)).to(dev)          # 2-bit codes
    ascii_np = np.frombuffer(b"ACGT", np.uint8)
    m = L - k + 1
    hg = _t_window_hashes(anc, torch.arange(m, device=dev, dtype=torch.int64), k, hash_seed)
    pos_h = hg[hg >= 0]
    pool_n = int(np.ceil(1.35 * s)) + 8
    if pos_h.numel() < pool_n:
        raise ValueError("genome too short for requested sketch size")
    T = int(torch.sort(pos_h).values[pool_n - 1].item()) + 1           # track hashes in [0, T)
    n_l = max(1, int(round(div_lineage * L)))
    n_s = max(1, int(round(div_strain * L)))
    ar_k = torch.arange(k, device=dev, dtype=torch.int64)
    ref = np.empty((n_genomes, s), np.uint64)
    lineage = np.empty(n_genomes, np.int32)
    truth_genome = None
    done = 0
    for lin in range(n_lineages):
        n_here = min(per_lin, n_genomes - done)
        if n_here <= 0:
            break
        # the lineage's genome and position-indexed hashes
        P = torch.randint(0, L, (n_l,), generator=gen, device=dev, dtype=torch.int64)
        gl = anc.clone()
        gl[P] = _snp_alt(anc[P], P, 1 + lin)
        A = torch.unique((P[:, None] - ar_k[None, :]).clamp(0, m - 1).reshape(-1))
        hl = hg.clone()
        hl[A] = _t_window_hashes(gl, A, k, hash_seed)
        pp = ((hl >= 0) & (hl < T)).nonzero(as_tuple=True)[0]
        hp, order = torch.sort(hl[pp])
        pp = pp[order]
        Np = pp.numel()
        # strains, a few at a time (dense copies of the lineage genome: J x L bytes, windows J x n_s*k x k x 8 bytes)
        per_strain = L + n_s * k * (k * 9 + 32) + Np * 24
        J_max = max(1, int(strain_chunk_bytes // per_strain))
        for j0 in range(0, n_here, J_max):
            J = min(J_max, n_here - j0)
            salt = 1000003 * (1 + lin) + j0
            Ps = torch.randint(0, L, (J, n_s), generator=gen, device=dev, dtype=torch.int64)
            sg = gl[None, :].expand(J, L).clone()
            salts = (salt + torch.arange(J, device=dev, dtype=torch.int64))[:, None]
            sg.scatter_(1, Ps, _snp_alt(torch.gather(gl[None, :].expand(J, L), 1, Ps), Ps + salts * 7, 0))
            As = (Ps[:, :, None] - ar_k[None, None, :]).clamp(0, m - 1).reshape(J, n_s * k)
            hn = _t_window_hashes(sg, As, k, hash_seed)
            hn = torch.where((hn >= 0) & (hn < T), hn, torch.full_like(hn, _I64_MAX))
            Psort = torch.sort(Ps, dim=1).values
            ppe = pp[None, :].expand(J, Np).contiguous()
            ix = torch.searchsorted(Psort, ppe)                          # first SNP at or behind the window's start
            nxt = torch.gather(Psort, 1, ix.clamp(max=n_s - 1))
            killed = (ix < n_s) & (nxt <= ppe + (k - 1))
            rows = torch.cat([torch.where(killed, torch.full_like(ppe, _I64_MAX), hp[None, :].expand(J, Np)), hn], dim=1)
            rows = torch.sort(rows, dim=1).values
            dup = rows[:, 1:] == rows[:, :-1]
            if bool(dup.any()):
                rows[:, 1:][dup] = _I64_MAX
                rows = torch.sort(rows, dim=1).values
            rows = rows[:, :s]
            if bool((rows[:, -1] == _I64_MAX).any()):
                raise ValueError("a strain holds fewer than s hashes below the tracked threshold (genome too short?)")
            block = rows.cpu().numpy().view(np.uint64)
            bad = np.nonzero((block[:, 1:] <= block[:, :-1]).any(axis=1))[0]
            if len(bad):  # (a row that came back from the device sort out of order -- seen once with the pool generator, under the
                block = block.copy()  # profiler: sorted again here rather than refused by skx_ref_create; the values are distinct)
                block[bad] = np.sort(block[bad], axis=1)
            ref[done + j0:done + j0 + J] = block
            t_local = truth - (done + j0)
            if 0 <= t_local < J:
                truth_genome = ascii_np[sg[t_local].cpu().numpy()]
            del sg, hn, rows, killed, ix, nxt, ppe, As, Ps
        lineage[done:done + n_here] = lin
        done += n_here
    truth_index = truth
    if shuffle:
        perm = rng.permutation(n_genomes)
        ref, lineage = ref[perm], lineage[perm]
        truth_index = int(np.nonzero(perm == truth)[0][0])
    col_len = np.full(n_genomes, s, np.uint32)
    return dict(genome=ascii_np[anc.cpu().numpy()], ref=np.ascontiguousarray(ref), col_len=col_len, k=k, seed=hash_seed, s=s,
                lineage=lineage, truth_index=truth_index, truth_genome=np.ascontiguousarray(truth_genome))


def write_bgzf(path, data, block=65280, level=1):
    """BGZF (bgzip / htslib): a series of gzip members of at most 64 KB, each with a 'BC' extra subfield holding its size - 1, closed
    by an empty member -- the form of .fastq.gz whose members `sketchy-hip` inflates in parallel.  data: bytes or a uint8 array."""
    import struct
    import zlib
    mv = memoryview(data).cast("B")
    with open(path, "wb") as f:
        for a in list(range(0, len(mv), block)) + [None]:
            chunk = b"" if a is None else mv[a:a + block]
            z = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = z.compress(chunk) + z.flush()
            bsize = 12 + 6 + len(comp) + 8
            f.write(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            f.write(comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
