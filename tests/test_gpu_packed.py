"""4-bit packed input (skx_stream_set_packed_input / skx_pack_bases): the same reads as two bases per byte must give
the rows, sketches, per-read counts and table of the ASCII path (and of the oracle) -- through the wave sketchers, the
block sketcher, the device-resident entry points and the host-fed pipeline; reads starting on odd nibbles, N / IUPAC /
lower case / U / whitespace (dropped by the packer), reads shorter than k, lengths around the 256-base groups and the
2048-base chunks."""
import numpy as np
import pytest

from helpers import pack_reads, workload
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _reads():
    ref, _, _ = workload(30, 300, 1, read_len=300, genome_len=120000, rng_seed=501)
    g = ref["genome"].tobytes()
    rng = np.random.default_rng(7)
    reads = [b"", b"ACGT", g[100:115], g[100:116], g[1000:1400].lower(), g[2000:2200] + b"N" + g[2201:2500],
             g[3000:3100] + b"\n" + g[3100:3200] + b"\r\n" + g[3200:3300] + b" \t", g[4000:4300].replace(b"T", b"U"),
             b"RYKMSWBDHV" * 20, b"A" * 500, g[5000:5000 + 2063], g[7000:7300] + b"-" + g[7301:7600]]
    for n in list(range(1, 40)) + [255, 256, 257, 258, 259, 1023, 1025, 2047, 2048, 2049, 2062, 2063, 2064, 4097, 9000]:
        a = int(rng.integers(0, len(g) - n - 1))
        reads.append(g[a:a + n])
    reads.append(g[60000:60000 + 30000])  # the long-read path
    return ref, reads


def test_packed_input_matches_ascii_and_oracle(gpu):
    from sketchy_amd import api
    ref, reads = _reads()
    bases, offsets = pack_reads(reads)
    exp = orc.stream(16, 0, 300, ref["ref"], np.full(30, 300, np.uint32), bases, offsets, top_k=2, want_shared=True, want_sketches=True)
    R = api.ReferenceSketch(ref["ref"])
    for first in (0, 1):
        packed, poff = api.pack_reads(bases, offsets, first_nibble=first)
        assert int(poff[-1]) - first == sum(len(bytes(r).translate(None, b" \t\r\n")) for r in reads)
        S = api.SumOfSharedHashes(R, top=2, max_batch_reads=len(reads), max_batch_bases=max(int(poff[-1]), len(bases)) + 2)
        S.set_packed_input(True)
        got = S.push(packed, poff, want_shared=True, want_sketches=True)   # debug outputs: full sketches
        np.testing.assert_array_equal(got["sketch_len"], exp["sketch_len"])
        np.testing.assert_array_equal(got["sketches"], exp["sketches"])
        np.testing.assert_array_equal(got["shared"], exp["shared"])
        np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"])
        np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"])
        np.testing.assert_array_equal(S.table(), exp["cum"])
        # production path, in two batches (the second one starts mid-stream, possibly on an odd nibble), then ASCII again
        S.reset()
        h = len(reads) // 2 + 1
        g1 = S.push(packed, poff[:h + 1])
        g2 = S.push(packed, poff[h:])
        np.testing.assert_array_equal(np.concatenate([g1["topk_idx"], g2["topk_idx"]]), exp["topk_idx"])
        np.testing.assert_array_equal(np.concatenate([g1["topk_sum"], g2["topk_sum"]]), exp["topk_sum"])
        S.set_packed_input(False)
        S.reset()
        g3 = S.push(bases, offsets)
        np.testing.assert_array_equal(g3["topk_idx"], exp["topk_idx"])


def test_packed_input_device_resident_and_host_fed(gpu):
    import ctypes as C
    from sketchy_amd import api
    ref, bases, offsets = workload(200, 400, 900, read_len=600, rng_seed=511)
    exp = orc.stream(16, 0, 400, ref["ref"], np.full(200, 400, np.uint32), bases, offsets, top_k=1)
    packed, poff = api.pack_reads(bases, offsets, first_nibble=1)
    R = api.ReferenceSketch(ref["ref"])
    S = api.SumOfSharedHashes(R, top=1, max_batch_reads=600, max_batch_bases=int(poff[-1]) + 2)
    S.set_packed_input(True)
    d_b = api.DeviceBuffer.from_numpy(packed)
    keep, rows = [d_b], []
    for a, b in ((0, 300), (300, 301), (301, 700), (700, 900)):
        d_o = api.DeviceBuffer.from_numpy(np.ascontiguousarray(poff[a:b + 1]))
        d_i, d_s = api.DeviceBuffer((b - a) * 4), api.DeviceBuffer((b - a) * 8)
        keep += [d_o, d_i, d_s]
        rows.append((b - a, d_i, d_s))
        S.enqueue_device(d_b.ptr, d_o.ptr, b - a, int(poff[b] - poff[a]), d_i.ptr, d_s.ptr)
    S.sync()
    np.testing.assert_array_equal(np.concatenate([d.to_numpy(np.uint32, (m, 1)) for m, d, _ in rows]), exp["topk_idx"])
    np.testing.assert_array_equal(np.concatenate([d.to_numpy(np.uint64, (m, 1)) for m, _, d in rows]), exp["topk_sum"])
    np.testing.assert_array_equal(S.table(), exp["cum"])
    for d in keep:
        d.free()
    # host-fed: page-locked packed batches through submit / drain
    S.reset()
    hb = api.HostBuffer(len(packed))
    hb.view(np.uint8)[:] = packed
    outs = []
    for a, b in ((0, 350), (350, 351), (351, 900)):
        ho = api.HostBuffer((b - a + 1) * 8)
        ho.view(np.uint64)[:] = poff[a:b + 1]
        hi, hs = api.HostBuffer((b - a) * 4), api.HostBuffer((b - a) * 8)
        S.submit(hb.ptr, ho.ptr, b - a, hi.ptr, hs.ptr)
        outs.append((b - a, ho, hi, hs))
    S.drain()
    np.testing.assert_array_equal(np.concatenate([hi.view(np.uint32)[:m] for m, _, hi, _ in outs]), exp["topk_idx"][:, 0])
    np.testing.assert_array_equal(np.concatenate([hs.view(np.uint64)[:m] for m, _, _, hs in outs]), exp["topk_sum"][:, 0])
    np.testing.assert_array_equal(S.table(), exp["cum"])
