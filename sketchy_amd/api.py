"""Thin Python handles over the C ABI (numpy in, numpy out; no torch, no CPU fallback).

Names follow the reference's own vocabulary for this path:
  ReferenceSketch              the `Vec<Sketch>` returned by Sketchy::_read_sketch (src/sketchy.rs:497-536)
  SumOfSharedHashes            the state and loop body of Sketchy::_sum_of_shared_hashes (src/sketchy.rs:317-356)
  common_hashes / sketch_reads Sketchy::_common_hashes (:419-459) and finch's process/to_vec (:331-335)
"""
import ctypes as C

import numpy as np

from . import _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_reads(bases, offsets, first_nibble=0):
    """ASCII reads (bases, offsets as for push()) -> (packed uint8 array, uint64 offsets in bases): two bases per byte, low
    nibble first, whitespace dropped, everything not ACGTU -> 4 (skx_pack_bases).  first_nibble: start the stream at that
    nibble (tests: reads on odd nibbles)."""
    L = _lib.load()
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n = len(offsets) - 1
    packed = np.zeros((first_nibble + len(bases) + 1) // 2 + 1, np.uint8)
    out = np.zeros(n + 1, np.uint64)
    pos = int(first_nibble)
    out[0] = pos
    for r in range(n):
        a, b = int(offsets[r]), int(offsets[r + 1])
        if b > a:
            pos = int(L.skx_pack_bases(bases[a:b].ctypes.data_as(C.c_void_p), b - a, _p(packed), pos))
        out[r + 1] = pos
    return packed[: (pos + 1) // 2 + 1], out


def set_option(name: str, value: int):
    """Process-wide policy for references / streams created from now on (skx_set_option), e.g. "kmer_prefilter"."""
    _lib.check(_lib.load().skx_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_uint64(0)
    _lib.check(_lib.load().skx_get_option(name.encode(), C.byref(v)))
    return v.value


def device_count() -> int:
    return _lib.load().skx_device_count()


def device_info(device=0):
    L = _lib.load()
    name = C.create_string_buffer(256)
    cu, mem = C.c_int(0), C.c_uint64(0)
    _lib.check(L.skx_device_info(device, name, 256, C.byref(cu), C.byref(mem)))
    return dict(name=name.value.decode(), compute_units=cu.value, total_mem=mem.value)


def device_mem(device=0):
    """(free, total) bytes of device memory"""
    f, t = C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.load().skx_dev_mem_info(device, C.byref(f), C.byref(t)))
    return f.value, t.value


class ReferenceSketch:
    """Reference sketch collection(s) resident in HBM.  hashes: [n_genomes, s] uint64, row g = genome g's ascending
    distinct hashes, first col_len[g] valid -- or a LIST of such matrices, one per species (same s, k, seed): they are
    scanned together in one pass per batch, every species keeps its own table and ranking (one `sketchy predict` run
    per species over the same reads, src/sketchy.rs:81-82).  s: the size reads are sketched with; default = the matrix
    width.  The reference uses the length of the collection's FIRST sketch (src/sketchy.rs:82, :520-527) -- pass
    s=col_len[0] to mirror a collection whose sketches differ in length."""

    def __init__(self, hashes, col_len=None, k=16, seed=0, device=0, s=None):
        L = _lib.load()
        many = isinstance(hashes, (list, tuple))
        mats = [np.ascontiguousarray(h, np.uint64) for h in (hashes if many else [hashes])]
        if any(m.ndim != 2 for m in mats) or len({m.shape[1] for m in mats}) != 1:
            raise ValueError("hashes must be [n_genomes, s] (one matrix per species, all with the same s)")
        lens = col_len if many else [col_len]
        lens = [None] * len(mats) if lens is None else list(lens)
        lens = [np.full(m.shape[0], m.shape[1], np.uint32) if c is None else np.ascontiguousarray(c, np.uint32)
                for m, c in zip(mats, lens)]
        self.species = [int(m.shape[0]) for m in mats]
        self.n_species = len(mats)
        self.n_genomes, self.stride = sum(self.species), int(mats[0].shape[1])
        self.s = self.stride if s is None else int(s)
        self.k, self.seed, self.device = int(k), int(seed), int(device)
        h = C.c_void_p()
        if self.n_species == 1:
            _lib.check(L.skx_ref_create(C.byref(h), device, self.k, self.seed, self.s, self.stride, self.n_genomes, _p(mats[0]), _p(lens[0])))
        else:
            ng = (C.c_uint32 * self.n_species)(*self.species)
            hp = (C.c_void_p * self.n_species)(*[m.ctypes.data for m in mats])
            cp = (C.c_void_p * self.n_species)(*[c.ctypes.data for c in lens])
            _lib.check(L.skx_ref_create_multi(C.byref(h), device, self.k, self.seed, self.s, self.stride, self.n_species, ng, hp, cp))
        self._h = h

    @property
    def kmer_filter(self):
        """(keys, table bytes) of the reference's k-mer prefilter; (0, 0) when none was built"""
        n, b = C.c_uint64(0), C.c_uint64(0)
        _lib.check(_lib.load().skx_ref_kmer_filter(self._h, C.byref(n), C.byref(b)))
        return n.value, b.value

    @property
    def rare_index(self):
        """dict(keys, rare_keys, postings, bytes) of the reference's rare-hash index (all 0: none was built)"""
        v = [C.c_uint64(0) for _ in range(4)]
        _lib.check(_lib.load().skx_ref_rare_index(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("keys", "rare_keys", "postings", "bytes"), [x.value for x in v]))

    @property
    def patterns(self):
        """dict(long_lists, patterns, pattern_lists, bytes): the long genome lists of the rare-hash index and how many of them are stored
        as a shared pattern + exceptions (all 0: none)"""
        v = [C.c_uint64(0) for _ in range(4)]
        _lib.check(_lib.load().skx_ref_patterns(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("long_lists", "patterns", "pattern_lists", "bytes"), [x.value for x in v]))

    @property
    def static_dense(self):
        """(is_static, n_hashes): does the reference have a static dense dictionary, and how many hashes the scan can be asked for"""
        f, n = C.c_int(0), C.c_uint64(0)
        _lib.check(_lib.load().skx_ref_static_dense(self._h, C.byref(f), C.byref(n)))
        return bool(f.value), n.value

    @property
    def pass_bytes(self) -> int:
        b = C.c_uint64(0)
        _lib.check(_lib.load().skx_ref_pass_bytes(self._h, C.byref(b)))
        return b.value

    def common_hashes(self, query, query_len=None) -> np.ndarray:
        """[n_query, n_genomes] uint32 intersection sizes (Sketchy::_common_hashes for every pair)."""
        query = np.ascontiguousarray(query, np.uint64)
        nq, stride = query.shape
        query_len = np.full(nq, stride, np.uint32) if query_len is None else np.ascontiguousarray(query_len, np.uint32)
        out = np.zeros((nq, self.n_genomes), np.uint32)
        _lib.check(_lib.load().skx_common_hashes(self._h, _p(query), _p(query_len), nq, stride, _p(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().skx_ref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SumOfSharedHashes:
    """Streaming predictor: running sum-of-shared-hashes table + per-read top rows."""

    def __init__(self, ref: ReferenceSketch, top=1, max_batch_reads=4096, max_batch_bases=None):
        L = _lib.load()
        self.ref, self.top = ref, int(top)
        self.max_batch_reads = int(max_batch_reads)
        self.max_batch_bases = int(max_batch_bases if max_batch_bases is not None else max_batch_reads * 2063)
        h = C.c_void_p()
        _lib.check(L.skx_stream_create(C.byref(h), ref._h, self.top, self.max_batch_reads, self.max_batch_bases))
        self._h = h

    def push(self, bases, offsets, want_shared=False, want_sketches=False):
        """Consume a packed batch; returns dict(topk_idx, topk_sum[, shared, sketches, sketch_len]).  Rows are
        [n_reads, top] for a single reference and [n_reads, n_species, top] (genome indices local to the species)
        for a multi-species one; shared / table() hold the species one after the other."""
        L = _lib.load()
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = len(offsets) - 1
        out = {}
        shape = (n, self.top) if self.ref.n_species == 1 else (n, self.ref.n_species, self.top)  # rows per species
        ti = np.zeros(shape, np.uint32) if self.top else None
        ts = np.zeros(shape, np.uint64) if self.top else None
        sh = np.zeros((n, self.ref.n_genomes), np.uint32) if want_shared else None
        sk = np.zeros((n, self.ref.s), np.uint64) if want_sketches else None
        sl = np.zeros(n, np.uint32) if want_sketches else None
        b = bases if len(bases) else np.zeros(1, np.uint8)
        _lib.check(L.skx_stream_push(self._h, _p(b), _p(offsets), n, _p(ti), _p(ts), _p(sh), _p(sk), _p(sl)))
        out.update(topk_idx=ti, topk_sum=ts, shared=sh, sketches=sk, sketch_len=sl)
        return out

    def push_device(self, d_bases, d_offsets, n_reads, n_bases, d_topk_idx=None, d_topk_sum=None):
        _lib.check(_lib.load().skx_stream_push_device(self._h, d_bases, d_offsets, n_reads, n_bases, d_topk_idx, d_topk_sum))

    def enqueue_device(self, d_bases, d_offsets, n_reads, n_bases, d_topk_idx=None, d_topk_sum=None):
        """push_device with the host wait of batch i overlapped by the sketch of batch i + 1: the passes (and any
        error) of a batch are queued by the NEXT enqueue / flush / sync.  Same rows, same table."""
        _lib.check(_lib.load().skx_stream_enqueue_device(self._h, d_bases, d_offsets, n_reads, n_bases, d_topk_idx, d_topk_sum))

    def flush(self):
        _lib.check(_lib.load().skx_stream_flush(self._h))

    def set_packed_input(self, on=True):
        """From now on `bases` of every entry point are 4-bit packed (pack_reads()), offsets count bases."""
        _lib.check(_lib.load().skx_stream_set_packed_input(self._h, int(bool(on))))

    def sync(self):
        _lib.check(_lib.load().skx_stream_sync(self._h))

    def submit(self, h_bases, h_offsets, n_reads, h_topk_idx=None, h_topk_sum=None) -> int:
        """Queue a batch held in page-locked host memory (HostBuffer pointers / addresses); returns its ticket.  The
        copy overlaps the previous batch's kernels; rows are valid after wait(ticket) or drain()."""
        t = C.c_uint64(0)
        _lib.check(_lib.load().skx_stream_submit(self._h, h_bases, h_offsets, n_reads, h_topk_idx, h_topk_sum, C.byref(t)))
        return t.value

    def wait(self, ticket):
        _lib.check(_lib.load().skx_stream_wait(self._h, ticket))

    def drain(self):
        _lib.check(_lib.load().skx_stream_drain(self._h))

    def table(self) -> np.ndarray:
        cum = np.zeros(self.ref.n_genomes, np.uint64)
        _lib.check(_lib.load().skx_stream_table(self._h, _p(cum)))
        return cum

    def table_add(self, add):
        add = np.ascontiguousarray(add, np.uint64)
        assert len(add) == self.ref.n_genomes
        _lib.check(_lib.load().skx_stream_table_add(self._h, _p(add)))

    def reset(self):
        _lib.check(_lib.load().skx_stream_reset(self._h))

    @property
    def reads(self) -> int:
        n = C.c_uint64(0)
        _lib.check(_lib.load().skx_stream_reads(self._h, C.byref(n)))
        return n.value

    def rank(self, top=None):
        top = self.top if top is None else int(top)
        shape = (top,) if self.ref.n_species == 1 else (self.ref.n_species, top)
        idx, sm = np.zeros(shape, np.uint32), np.zeros(shape, np.uint64)
        _lib.check(_lib.load().skx_stream_rank(self._h, top, _p(idx), _p(sm)))
        return idx, sm

    def stats(self):
        """Counters of the stream (skx_stream_stats): pairs / passes of the last push, dictionary size, ..."""
        v = (C.c_uint64 * 18)()
        _lib.check(_lib.load().skx_stream_stats(self._h, v, 18))
        names = ("last_pairs", "last_passes", "dictionary_size", "reads_block_sketcher", "passes", "passes_lean_scan",
                 "pair_capacity", "live_rank_groups", "reads_split_over_waves", "read_segments", "row_pool_grown",
                 "passes_shared", "groups_unshared", "query_rows", "query_rows_grown", "dictionary_dense", "batches_compact", "batches_full")
        return dict(zip(names, [int(x) for x in v]))

    def set_profiling(self, on=True):
        """False/0: off; True/1: every stage; 2: only the reference scan (cheapest way to time the roofline kernel)."""
        _lib.check(_lib.load().skx_stream_set_profiling(self._h, int(on)))

    def profile(self):
        ms = (C.c_double * _lib.N_STAGES)()
        n = (C.c_uint64 * _lib.N_STAGES)()
        _lib.check(_lib.load().skx_stream_profile(self._h, ms, n))
        return {name: dict(ms=ms[i], launches=int(n[i])) for i, name in enumerate(_lib.STAGE_NAMES)}

    def scan_alone(self, reps=3) -> float:
        """average milliseconds of `reps` reference scans with nothing beside them (streams on a reference with a static dense dictionary)"""
        ms = C.c_double(0)
        _lib.check(_lib.load().skx_stream_scan_alone(self._h, int(reps), C.byref(ms)))
        return ms.value

    def allreduce(self, comm):
        _lib.check(_lib.load().skx_stream_allreduce(self._h, comm._h))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().skx_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator (one process per GPU)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.load().skx_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, device, rank, n_ranks, uid: bytes):
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        _lib.check(_lib.load().skx_comm_create(C.byref(h), device, rank, n_ranks, buf))
        self._h = h

    @property
    def n_ranks(self) -> int:
        n = C.c_int(0)
        _lib.check(_lib.load().skx_comm_n_ranks(self._h, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().skx_comm_destroy(self._h)
            self._h = None


def sketch_reads(bases, offsets, k=16, seed=0, s=1000, device=0):
    """finch MashSketcher process/to_vec per read: ([n_reads, s] uint64 ascending, lengths)."""
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n = len(offsets) - 1
    sk = np.zeros((n, s), np.uint64)
    sl = np.zeros(n, np.uint32)
    b = bases if len(bases) else np.zeros(1, np.uint8)
    _lib.check(_lib.load().skx_sketch_reads(device, k, seed, s, _p(b), _p(offsets), n, _p(sk), _p(sl)))
    return sk, sl


class DeviceBuffer:
    """A raw HBM allocation (bench path: inputs resident on the device before timing starts)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        _lib.check(_lib.load().skx_dev_malloc(device, C.byref(p), self.nbytes))
        self.ptr = p

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        buf = cls(max(a.nbytes, 1), device)
        if a.nbytes:
            _lib.check(_lib.load().skx_dev_upload(device, buf.ptr, _p(a), a.nbytes))
        return buf

    def to_numpy(self, dtype, shape):
        out = np.zeros(shape, dtype)
        _lib.check(_lib.load().skx_dev_download(self.device, _p(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _lib.load().skx_dev_free(self.device, self.ptr)
            self.ptr = None


class HostBuffer:
    """Page-locked host memory (skx_host_alloc) with a numpy view: batch buffers for SumOfSharedHashes.submit."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        _lib.check(_lib.load().skx_host_alloc(device, C.byref(p), max(self.nbytes, 1)))
        self.ptr = p

    def view(self, dtype, count=None, offset=0):
        dt = np.dtype(dtype)
        count = (self.nbytes - offset) // dt.itemsize if count is None else count
        buf = (C.c_uint8 * (count * dt.itemsize)).from_address(self.ptr.value + offset)
        return np.frombuffer(buf, dtype=dt, count=count)

    def free(self):
        if self.ptr:
            _lib.load().skx_host_free(self.device, self.ptr)
            self.ptr = None
