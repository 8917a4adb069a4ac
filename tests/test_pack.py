"""skx_pack_bases / skx_pack_line (host helpers of the C ABI, no device): the vectorised packers against the definition in
include/sketchy_hip.h -- classify every byte as the sketchers do (A C G T/U in either case = 0..3, whitespace dropped, anything
else 4), an even nibble starts its byte afresh, an odd one keeps the low nibble already there."""
import ctypes as C

import numpy as np

from sketchy_amd import _lib

CODE = np.full(256, 4, np.uint8)
for _ch, _v in zip(b"ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3]):
    CODE[_ch] = _v
for _ch in b" \t\r\n":
    CODE[_ch] = 5


def _ref_pack(a, pos, buf):
    for c in a:
        k = CODE[c]
        if k == 5:
            continue
        if pos & 1:
            buf[pos >> 1] = (buf[pos >> 1] & 0x0F) | (k << 4)
        else:
            buf[pos >> 1] = k
        pos += 1
    return pos


def _inputs(rng, trial):
    n = int(rng.integers(0, 300))
    mode = trial % 4
    if mode == 0:
        return rng.integers(0, 256, n).astype(np.uint8)           # every byte value
    alphabet = np.frombuffer(b"ACGTacgtNnUu-RYK \t\n\r@!~\x00\xff", np.uint8)
    p = [0.5, 0.97, 1.0][mode - 1]
    return np.where(rng.random(n) < p, rng.choice(np.frombuffer(b"ACGT", np.uint8), n), rng.choice(alphabet, n)).astype(np.uint8)


def test_pack_bases_matches_the_definition():
    L = _lib.load()
    rng = np.random.default_rng(1)
    for trial in range(800):
        a = _inputs(rng, trial)
        pos0 = int(rng.integers(0, 7))
        b1 = np.full(len(a) // 2 + 8, 0xAB, np.uint8)
        b2 = b1.copy()
        e1 = _ref_pack(a, pos0, b1)
        e2 = L.skx_pack_bases(a.ctypes.data_as(C.c_void_p), len(a), b2.ctypes.data_as(C.c_void_p), pos0)
        assert e1 == e2
        np.testing.assert_array_equal(b1[pos0 // 2:(e1 + 1) // 2], b2[pos0 // 2:(e1 + 1) // 2])
        np.testing.assert_array_equal(b2[(e1 + 1) // 2 + 16:], b1[(e1 + 1) // 2 + 16:])   # (vector stores stay within 16 bytes of the end)


def test_pack_line_stops_at_the_first_line_feed():
    L = _lib.load()
    rng = np.random.default_rng(2)
    for trial in range(800):
        a = _inputs(rng, trial)
        if trial % 3 == 0 and len(a):
            a[int(rng.integers(0, len(a)))] = 10                  # a line feed somewhere (also in front of the vector blocks)
        nl = np.nonzero(a == 10)[0]
        m = int(nl[0]) if len(nl) else len(a)
        pos0 = int(rng.integers(0, 7))
        b1 = np.full(len(a) // 2 + 8, 0xCD, np.uint8)
        b2 = b1.copy()
        e1 = _ref_pack(a[:m], pos0, b1)
        used = C.c_uint64(0)
        e2 = L.skx_pack_line(a.ctypes.data_as(C.c_void_p), len(a), b2.ctypes.data_as(C.c_void_p), pos0, C.byref(used))
        assert e1 == e2 and used.value == (m + 1 if len(nl) else len(a))
        np.testing.assert_array_equal(b1[pos0 // 2:(e1 + 1) // 2], b2[pos0 // 2:(e1 + 1) // 2])
