"""Independent (Python) writer/reader of Mash `.msh` files (un-packed Cap'n Proto, single segment), used to
hand sketches to the C++ host in tests.  Field layout restated from Mash's public MinHash.capnp
([UPSTREAM-RECALL], see sketchy_amd/host/formats.hpp): root struct 3 data words + 4 pointers; hashSeed at
byte 20 stored XOR 42; referenceList = pointer 3; Reference = 3 data words + 7 pointers (name 2, comment 3,
hashes64 5)."""
import struct

import numpy as np


def _sptr(off, dw, pw):
    return ((off << 2) & 0xFFFFFFFF) | (dw << 32) | (pw << 48)


def _lptr(off, code, count):
    return (((off << 2) & 0xFFFFFFFF) | 1) | (code << 32) | (count << 35)


def write_msh(path, names, hashes_list, kmer=16, seed=0, lengths=None, hashes32=False):
    """hashes32: store the hashes as Mash does for k <= 16 -- 32-bit values in the hashes32 list (pointer 4), hashes64 empty"""
    w = [0] * (1 + 3 + 4)
    w[0] = _sptr(0, 3, 4)
    root = 1
    w[root + 0] = kmer
    w[root + 2] = ((seed ^ 42) & 0xFFFFFFFF) << 32
    rl = len(w); w.append(0)
    w[root + 3 + 3] = _sptr(rl - (root + 3 + 3) - 1, 0, 1)
    n, esz = len(names), 10
    tag = len(w); w.extend([0] * (1 + n * esz))
    w[rl] = _lptr(tag - rl - 1, 7, n * esz)
    w[tag] = (n << 2) | (3 << 32) | (7 << 48)
    blobs = []
    for i, (name, hs) in enumerate(zip(names, hashes_list)):
        e = tag + 1 + i * esz
        ln = int(lengths[i]) if lengths is not None else 0
        w[e + 0] = min(ln, 0xFFFFFFFF); w[e + 1] = ln; w[e + 2] = 0
        for pidx, txt in ((2, name.encode()), (3, b"")):
            at = len(w); cnt = len(txt) + 1
            buf = txt + b"\0" * ((-cnt) % 8 + 1)
            w.extend(struct.unpack("<%dQ" % (len(buf) // 8), buf))
            w[e + 3 + pidx] = _lptr(at - (e + 3 + pidx) - 1, 2, cnt)
        at = len(w)
        if hashes32:
            h32 = [int(x) & 0xFFFFFFFF for x in hs] + [0] * (len(hs) & 1)
            w.extend(h32[j] | (h32[j + 1] << 32) for j in range(0, len(h32), 2))
            w[e + 3 + 4] = _lptr(at - (e + 3 + 4) - 1, 4, len(hs))
        else:
            w.extend(int(x) for x in hs)
            w[e + 3 + 5] = _lptr(at - (e + 3 + 5) - 1, 5, len(hs))
    arr = np.array(w, dtype=np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0, len(arr)))
        f.write(arr.tobytes())


def read_msh(path):
    """Reads back a single-segment `.msh` (as written above or by the C++ host's write_mash_file):
    returns (kmer, seed, [dict(name, length, num_valid_kmers, hashes)])."""
    raw = open(path, "rb").read()
    nseg, seg_len = struct.unpack("<II", raw[:8])
    assert nseg == 0, "single segment expected"
    w = np.frombuffer(raw[8:8 + 8 * seg_len], dtype=np.uint64)

    def sptr(at):  # -> (first content word, data words, pointer words)
        p = int(w[at])
        assert p & 3 == 0
        off = ((p & 0xFFFFFFFF) >> 2)
        off = off - (1 << 30) if off >= (1 << 29) else off
        return at + 1 + off, (p >> 32) & 0xFFFF, (p >> 48) & 0xFFFF

    def lptr(at):  # -> (first content word, element code, count)
        p = int(w[at])
        if p == 0:
            return None
        assert p & 3 == 1
        off = ((p & 0xFFFFFFFF) >> 2)
        off = off - (1 << 30) if off >= (1 << 29) else off
        return at + 1 + off, (p >> 32) & 7, p >> 35

    root, dw, pw = sptr(0)
    kmer = int(w[root]) & 0xFFFFFFFF
    seed = ((int(w[root + 2]) >> 32) & 0xFFFFFFFF) ^ 42
    rl, _, _ = sptr(root + dw + 3)
    tag, code, _ = lptr(rl)
    assert code == 7
    n = (int(w[tag]) & 0xFFFFFFFF) >> 2
    edw, epw = (int(w[tag]) >> 32) & 0xFFFF, (int(w[tag]) >> 48) & 0xFFFF
    out = []
    for i in range(n):
        e = tag + 1 + i * (edw + epw)
        name_at, _, cnt = lptr(e + edw + 2)
        name = w[name_at:name_at + (cnt + 7) // 8].tobytes()[:cnt - 1].decode()
        hl = lptr(e + edw + 5)
        hashes = np.array(w[hl[0]:hl[0] + hl[2]], dtype=np.uint64) if hl else np.zeros(0, np.uint64)
        out.append(dict(name=name, length=int(w[e + 1]), num_valid_kmers=int(w[e + 2]), hashes=hashes))
    return kmer, seed, out
