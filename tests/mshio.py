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


def write_msh(path, names, hashes_list, kmer=16, seed=0, lengths=None):
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
        w.extend(int(x) for x in hs)
        w[e + 3 + 5] = _lptr(at - (e + 3 + 5) - 1, 5, len(hs))
    arr = np.array(w, dtype=np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0, len(arr)))
        f.write(arr.tobytes())
