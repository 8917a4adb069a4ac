"""Mash `.msh` writer for large collections (un-packed Cap'n Proto, one segment): what bench.py / tools hand the C++ host when
they need a species-scale reference on disk -- 40 000 sketches x 10 000 hashes are 3.2 GB, and a writer that goes through
Python integers (tests/mshio.py, the tests' independent one) takes minutes for that.  Same layout as
sketchy_amd/host/formats.hpp::write_mash_file and tests/mshio.py ([UPSTREAM-RECALL] Mash's MinHash.capnp): root struct 3 data
words + 4 pointers, hashSeed at byte 20 stored XOR 42, referenceList = pointer 3, Reference = 3 data words + 7 pointers
(name 2, comment 3, hashes64 5).  Words are laid out as: header, reference table, then per reference its name, its (empty) comment
and its hashes; every position is computed up front, the hashes go out with ndarray.tofile."""
import struct

import numpy as np


def _sptr(off, dw, pw):
    return ((off << 2) & 0xFFFFFFFF) | (dw << 32) | (pw << 48)


def _lptr(off, code, count):
    return (((off << 2) & 0xFFFFFFFF) | 1) | (code << 32) | (count << 35)


def write_msh(path, names, hashes, col_len=None, kmer=16, seed=0, lengths=None):
    """names: list[str]; hashes: [n, s] uint64 array (row g = sketch g, ascending; the first col_len[g] entries count)."""
    hashes = np.ascontiguousarray(hashes, np.uint64)
    n, s = hashes.shape
    if len(names) != n:
        raise ValueError("one name per sketch")
    lens = np.full(n, s, np.int64) if col_len is None else np.asarray(col_len, np.int64)
    enc = [nm.encode() for nm in names]
    name_words = np.array([(len(b) + 1 + 7) // 8 for b in enc], np.int64)
    esz, head = 10, 1 + 3 + 4 + 1            # root pointer + root struct + ReferenceList struct
    tag = head
    body0 = tag + 1 + n * esz
    per = name_words + 1 + lens              # name, one word of empty comment text, hashes
    start = body0 + np.concatenate(([0], np.cumsum(per)[:-1]))
    total = int(body0 + per.sum())
    if total - 1 >= (1 << 29):
        raise ValueError("collection too large for one Cap'n Proto segment")
    w = np.zeros(body0, np.uint64)
    w[0] = _sptr(0, 3, 4)
    w[1] = kmer
    w[3] = ((seed ^ 42) & 0xFFFFFFFF) << 32
    rl = 1 + 3 + 4
    w[1 + 3 + 3] = _sptr(rl - (1 + 3 + 3) - 1, 0, 1)
    w[rl] = _lptr(tag - rl - 1, 7, n * esz)
    w[tag] = (n << 2) | (3 << 32) | (7 << 48)
    e = tag + 1 + np.arange(n, dtype=np.int64) * esz
    ln = np.zeros(n, np.int64) if lengths is None else np.asarray(lengths, np.int64)
    w[e + 0] = np.minimum(ln, 0xFFFFFFFF).astype(np.uint64)
    w[e + 1] = ln.astype(np.uint64)

    def lptr_vec(at, slot, code, count):
        off = (at - slot - 1).astype(np.int64)
        return ((off << 2) & 0xFFFFFFFF).astype(np.uint64) | np.uint64(1) | (np.uint64(code) << np.uint64(32)) | (count.astype(np.uint64) << np.uint64(35))

    name_cnt = np.array([len(b) + 1 for b in enc], np.int64)
    w[e + 3 + 2] = lptr_vec(start, e + 3 + 2, 2, name_cnt)
    w[e + 3 + 3] = lptr_vec(start + name_words, e + 3 + 3, 2, np.ones(n, np.int64))
    w[e + 3 + 5] = lptr_vec(start + name_words + 1, e + 3 + 5, 5, lens)
    zero = np.zeros(1, np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0, total))
        w.tofile(f)
        for g in range(n):
            b = enc[g]
            f.write(b + b"\0" * (int(name_words[g]) * 8 - len(b)))
            zero.tofile(f)
            hashes[g, :int(lens[g])].tofile(f)
