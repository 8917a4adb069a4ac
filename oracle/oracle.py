"""ctypes binding of oracle/liboracle.so plus an independent pure-Python restatement.

TEST INFRASTRUCTURE ONLY (see oracle.c header): imported by tests/, by
``__graft_entry__.smoke()`` and by bench.py's ``cpu_baseline`` leg; never by sketchy_amd/.

PARITY UNPINNED against upstream: the reference has no tests/fixtures for this path and its
Rust toolchain and crates are unavailable; the oracle is pinned by public MurmurHash3
known answers and by the two restatements (C, Python) agreeing with each other.

The ``py_*`` functions are a second, deliberately naive restatement (slow; small cases only)
of the same reference semantics, written independently of the C code:
  py_murmur3_x64_128  murmurhash3 0.0.5 (canonical MurmurHash3_x64_128; finch keeps h1)
  py_normalize        needletail 0.4.1 sequence.rs normalize(iupac=false)
  py_sketch           finch 0.4.1 MashSketcher net semantics (src/sketchy.rs:331-335)
  py_common           src/sketchy.rs:419-438
  py_stream           src/sketchy.rs:317-356 (+ stable sort :348, rows :391)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
M64 = (1 << 64) - 1


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.orc_murmur3_x64_128.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, u64p]
        L.orc_murmur3_x64_128.restype = None
        L.orc_normalize.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_normalize.restype = C.c_uint64
        L.orc_kmer_hashes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_kmer_hashes.restype = C.c_uint64
        for f in (L.orc_sketch_sort, L.orc_sketch_heap):
            f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
            f.restype = C.c_uint64
        L.orc_common_hashes.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_common_hashes.restype = C.c_uint64
        L.orc_stable_rank.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_stable_rank.restype = None
        L.orc_stream.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_stream.restype = C.c_int
        L.orc_stream_strided.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_stream_strided.restype = C.c_int
        L.orc_stream_mt.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int]
        L.orc_stream_mt.restype = C.c_int
        L.orc_stream_fast.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_stream_fast.restype = C.c_int
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- C oracle
def murmur3_x64_128(data: bytes, seed: int = 0):
    out = (C.c_uint64 * 2)()
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    lib().orc_murmur3_x64_128(_ptr(buf), len(data), seed, out)
    return int(out[0]), int(out[1])


def normalize(seq: bytes) -> bytes:
    src = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(1, np.uint8)
    dst = np.zeros(len(seq) + 1, np.uint8)
    n = lib().orc_normalize(_ptr(src), len(seq), _ptr(dst))
    return dst[:n].tobytes()


def kmer_hashes(seq: bytes, k: int, seed: int):
    src = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(1, np.uint8)
    out = np.zeros(len(seq) + 1, np.uint64)
    rc = np.zeros(len(seq) + 1, np.uint8)
    m = lib().orc_kmer_hashes(_ptr(src), len(seq), k, seed, _ptr(out), _ptr(rc))
    return out[:m].copy(), rc[:m].copy()


def sketch(seq: bytes, k: int, seed: int, s: int, impl: str = "heap") -> np.ndarray:
    src = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(1, np.uint8)
    out = np.zeros(s + 1, np.uint64)
    f = lib().orc_sketch_heap if impl == "heap" else lib().orc_sketch_sort
    n = f(_ptr(src), len(seq), k, seed, s, _ptr(out))
    return out[:n].copy()


def common_hashes(ref: np.ndarray, query: np.ndarray) -> int:
    ref = np.ascontiguousarray(ref, np.uint64)
    query = np.ascontiguousarray(query, np.uint64)
    return int(lib().orc_common_hashes(_ptr(ref), len(ref), _ptr(query), len(query)))


def stable_rank(sums: np.ndarray) -> np.ndarray:
    sums = np.ascontiguousarray(sums, np.uint64)
    idx = np.zeros(len(sums) + 1, np.uint32)
    lib().orc_stable_rank(_ptr(sums), len(sums), _ptr(idx))
    return idx[: len(sums)].copy()


def stream(k, seed, s, ref_hashes, col_len, bases, offsets, top_k=1, cum=None,
           want_shared=False, want_sketches=False, rank_every_read=True):
    """Run the streaming driver over a packed batch.  ref_hashes: [n_genomes, stride] uint64
    (row g = genome g's ascending hashes, first col_len[g] valid); s = the size reads are sketched with -- the row
    stride unless the collection's first sketch is shorter / longer than others (src/sketchy.rs:82, :520-527)."""
    ref_hashes = np.ascontiguousarray(ref_hashes, np.uint64)
    n_genomes, stride = ref_hashes.shape
    col_len = np.ascontiguousarray(col_len, np.uint32)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n_reads = len(offsets) - 1
    cum = np.zeros(n_genomes, np.uint64) if cum is None else np.ascontiguousarray(cum, np.uint64).copy()
    tk_i = np.zeros((n_reads, top_k), np.uint32)
    tk_s = np.zeros((n_reads, top_k), np.uint64)
    shared = np.zeros((n_reads, n_genomes), np.uint32) if want_shared else None
    sk = np.zeros((n_reads, s), np.uint64) if want_sketches else None
    sl = np.zeros(n_reads, np.uint32) if want_sketches else None
    bases_p = bases if len(bases) else np.zeros(1, np.uint8)
    rc = lib().orc_stream_strided(k, seed, s, stride, n_genomes, _ptr(ref_hashes), _ptr(col_len), _ptr(bases_p), _ptr(offsets),
                                  n_reads, top_k, _ptr(cum), _ptr(tk_i), _ptr(tk_s), _ptr(shared), _ptr(sk), _ptr(sl),
                                  1 if rank_every_read else 0)
    if rc != 0:
        raise ValueError("orc_stream failed (top_k > n_genomes?)")
    return dict(cum=cum, topk_idx=tk_i, topk_sum=tk_s, shared=shared, sketches=sk, sketch_len=sl)


def stream_mt(k, seed, s, ref_hashes, col_len, bases, offsets, top_k=1, cum=None, n_threads=0):
    """orc_stream with the per-read intersections spread over host threads (OpenMP over genomes): the
    all-host-cores upper bound of the CPU baseline.  n_threads 0 = os.cpu_count()."""
    ref_hashes = np.ascontiguousarray(ref_hashes, np.uint64)
    n_genomes = ref_hashes.shape[0]
    assert ref_hashes.shape[1] == s
    col_len = np.ascontiguousarray(col_len, np.uint32)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n_reads = len(offsets) - 1
    cum = np.zeros(n_genomes, np.uint64) if cum is None else np.ascontiguousarray(cum, np.uint64).copy()
    tk_i = np.zeros((n_reads, top_k), np.uint32)
    tk_s = np.zeros((n_reads, top_k), np.uint64)
    bases_p = bases if len(bases) else np.zeros(1, np.uint8)
    rc = lib().orc_stream_mt(k, seed, s, n_genomes, _ptr(ref_hashes), _ptr(col_len), _ptr(bases_p), _ptr(offsets),
                             n_reads, top_k, _ptr(cum), _ptr(tk_i), _ptr(tk_s), n_threads or (os.cpu_count() or 1))
    if rc != 0:
        raise ValueError("orc_stream_mt failed (top_k > n_genomes?)")
    return dict(cum=cum, topk_idx=tk_i, topk_sum=tk_s)


def usable_threads():
    """host threads this process may really use (affinity mask and cgroup CPU quota; a 16-CPU quota on a 256-thread box
    makes 256 threads slower than one)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def stream_fast(k, seed, s, ref_hashes, col_len, bases, offsets, top_k=1, cum=None, n_threads=0, block_reads=0,
                rows=True, index_min=0):
    """orc_stream_fast: the rows and the table of stream() for a FULL-SIZE batch in seconds (blocks of reads scored
    through one distinct-hash membership matrix, genomes split over host threads).  Exact; pinned against stream() in
    tests/test_oracle.py.  rows=False: table only.  Returns dict(cum, topk_idx, topk_sum, stats)."""
    ref_hashes = np.ascontiguousarray(ref_hashes, np.uint64)
    n_genomes, stride = ref_hashes.shape
    col_len = np.full(n_genomes, stride, np.uint32) if col_len is None else np.ascontiguousarray(col_len, np.uint32)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n_reads = len(offsets) - 1
    cum = np.zeros(n_genomes, np.uint64) if cum is None else np.ascontiguousarray(cum, np.uint64).copy()
    tk_i = np.zeros((n_reads, top_k), np.uint32) if rows and top_k else None
    tk_s = np.zeros((n_reads, top_k), np.uint64) if rows and top_k else None
    stats = np.zeros(8, np.uint64)
    bases_p = bases if len(bases) else np.zeros(1, np.uint8)
    rc = lib().orc_stream_fast(k, seed, s, stride, n_genomes, _ptr(ref_hashes), _ptr(col_len), _ptr(bases_p), _ptr(offsets),
                               n_reads, top_k, _ptr(cum), _ptr(tk_i), _ptr(tk_s), n_threads or usable_threads(),
                               block_reads, index_min, _ptr(stats))
    if rc != 0:
        raise ValueError("orc_stream_fast failed (top_k > n_genomes?)")
    names = ("reads_without_pairs", "pairs", "distinct_sum", "blocks", "member_bits")
    return dict(cum=cum, topk_idx=tk_i, topk_sum=tk_s, stats=dict(zip(names, [int(x) for x in stats[:5]])))


def stream_fast_species(k, seed, s, refs, bases, offsets, top_k=1, cums=None, n_threads=0, rows=True):
    """One stream_fast per species over the same reads (one `sketchy predict` run per species, src/sketchy.rs:81-82):
    rows [n_reads, n_species, top_k] with indices local to the species, cum = the species' tables one after the other."""
    outs = [stream_fast(k, seed, s, r, None, bases, offsets, top_k, None if cums is None else cums[i], n_threads, rows=rows)
            for i, r in enumerate(refs)]
    res = dict(cum=np.concatenate([o["cum"] for o in outs]), cums=[o["cum"] for o in outs], stats=[o["stats"] for o in outs])
    if rows and top_k:
        res["topk_idx"] = np.stack([o["topk_idx"] for o in outs], axis=1)
        res["topk_sum"] = np.stack([o["topk_sum"] for o in outs], axis=1)
    return res


# ----------------------------------------------------------------------------- pure Python
def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(k):
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & M64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & M64
    k ^= k >> 33
    return k


def py_murmur3_x64_128(data: bytes, seed: int = 0):
    c1, c2 = 0x87C37B91114253D5, 0x4CF5AD432745937F
    h1 = h2 = seed & M64
    n = len(data)
    nb = n // 16
    for b in range(nb):
        k1 = int.from_bytes(data[16 * b:16 * b + 8], "little")
        k2 = int.from_bytes(data[16 * b + 8:16 * b + 16], "little")
        k1 = (k1 * c1) & M64; k1 = _rotl(k1, 31); k1 = (k1 * c2) & M64; h1 ^= k1
        h1 = _rotl(h1, 27); h1 = (h1 + h2) & M64; h1 = (h1 * 5 + 0x52DCE729) & M64
        k2 = (k2 * c2) & M64; k2 = _rotl(k2, 33); k2 = (k2 * c1) & M64; h2 ^= k2
        h2 = _rotl(h2, 31); h2 = (h2 + h1) & M64; h2 = (h2 * 5 + 0x38495AB5) & M64
    tail = data[16 * nb:]
    t = len(tail)
    if t > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = (k2 * c2) & M64; k2 = _rotl(k2, 33); k2 = (k2 * c1) & M64; h2 ^= k2
    if t > 0:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = (k1 * c1) & M64; k1 = _rotl(k1, 31); k1 = (k1 * c2) & M64; h1 ^= k1
    h1 ^= n; h2 ^= n
    h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    h1 = _fmix(h1); h2 = _fmix(h2)
    h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    return h1, h2


_NORM = {}
for _c in b"ACGTN-":
    _NORM[_c] = _c
for _a, _b in zip(b"acgt", b"ACGT"):
    _NORM[_a] = _b
_NORM[ord("u")] = _NORM[ord("U")] = ord("T")
_NORM[ord(".")] = _NORM[ord("~")] = ord("-")
_WS = set(b" \t\r\n")
_COMP = {ord("A"): ord("T"), ord("T"): ord("A"), ord("C"): ord("G"), ord("G"): ord("C")}


def py_normalize(seq: bytes) -> bytes:
    return bytes(_NORM.get(c, ord("N")) for c in seq if c not in _WS)


def py_canonical_kmers(seq: bytes, k: int):
    """[(pos, kmer_bytes, is_rc)] over the normalised sequence."""
    norm = py_normalize(seq)
    out = []
    for p in range(0, len(norm) - k + 1):
        fwd = norm[p:p + k]
        if any(c not in _COMP for c in fwd):
            continue
        rev = bytes(_COMP[c] for c in reversed(fwd))
        out.append((p, fwd, False) if fwd < rev else (p, rev, True))
    return out


def py_sketch(seq: bytes, k: int, seed: int, s: int):
    hs = {py_murmur3_x64_128(km, seed)[0] for _, km, _ in py_canonical_kmers(seq, k)}
    return sorted(hs)[:s]


def py_common(ref, query) -> int:
    return len(set(int(x) for x in ref) & set(int(x) for x in query))


def py_stream(k, seed, s, ref_cols, reads, top_k=1):
    """ref_cols: list of ascending hash lists; reads: list of bytes. Returns per-read rows
    [(idx, sum) * top_k], the per-read shared matrix and the final table."""
    cum = [0] * len(ref_cols)
    rows, shared = [], []
    for rd in reads:
        sk = py_sketch(rd, k, seed, s)
        sh = [py_common(col, sk) for col in ref_cols]
        shared.append(sh)
        cum = [a + b for a, b in zip(cum, sh)]
        order = sorted(range(len(cum)), key=lambda i: -cum[i])  # Python's sort is stable
        rows.append([(i, cum[i]) for i in order[:top_k]])
    return rows, shared, cum
