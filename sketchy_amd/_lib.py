"""ctypes binding of libsketchy_hip.so (include/sketchy_hip.h).

Loading fails loudly when the library has not been built: there is no CPU or PyTorch
fallback for the hot path.  Build it with ``python -m sketchy_amd.build``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SKX_LIB_PATH: load another build of the same ABI instead -- A/B measurements of kernel changes on one GPU box)
LIB_PATH = os.environ.get("SKX_LIB_PATH") or os.path.join(_HERE, "libsketchy_hip.so")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSORTED, ERR_CAPACITY, ERR_COMM, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
N_STAGES = 5
STAGE_NAMES = ("sketch", "dictionary", "scan", "transpose", "rank")
COMM_ID_BYTES = 128
MAX_K = 32
MAX_TOP = 64
MAX_SPECIES = 64

# every symbol include/sketchy_hip.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("skx_last_error", C.c_char_p, []),
    ("skx_version", C.c_char_p, []),
    ("skx_device_count", _i, []),
    ("skx_device_info", _i, [_i, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(_u64)]),
    ("skx_device_pci_bus_id", _i, [_i, C.c_char_p, _sz]),
    ("skx_set_option", _i, [C.c_char_p, _u64]),
    ("skx_get_option", _i, [C.c_char_p, C.POINTER(_u64)]),
    ("skx_ref_kmer_filter", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    ("skx_ref_rare_index", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    ("skx_ref_patterns", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    ("skx_ref_static_dense", _i, [_vp, C.POINTER(C.c_int), C.POINTER(_u64)]),
    ("skx_ref_create", _i, [_pp, _i, _u32, _u64, _u32, _u32, _u32, _vp, _vp]),
    ("skx_ref_create_multi", _i, [_pp, _i, _u32, _u64, _u32, _u32, _u32, _vp, _vp, _vp]),
    ("skx_ref_sketch_size", _i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    ("skx_ref_n_genomes", _i, [_vp, C.POINTER(_u32)]),
    ("skx_ref_n_species", _i, [_vp, C.POINTER(_u32)]),
    ("skx_ref_species_genomes", _i, [_vp, _u32, C.POINTER(_u32)]),
    ("skx_ref_pass_bytes", _i, [_vp, C.POINTER(_u64)]),
    ("skx_ref_destroy", None, [_vp]),
    ("skx_stream_create", _i, [_pp, _vp, _u32, _u32, _u64]),
    ("skx_stream_push", _i, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    ("skx_stream_push_device", _i, [_vp, _vp, _vp, _u32, _u64, _vp, _vp]),
    ("skx_stream_enqueue_device", _i, [_vp, _vp, _vp, _u32, _u64, _vp, _vp]),
    ("skx_stream_flush", _i, [_vp]),
    ("skx_stream_set_packed_input", _i, [_vp, _i]),
    ("skx_pack_bases", _u64, [_vp, _u64, _vp, _u64]),
    ("skx_pack_line", _u64, [_vp, _u64, _vp, _u64, C.POINTER(_u64)]),
    ("skx_stream_sync", _i, [_vp]),
    ("skx_stream_submit", _i, [_vp, _vp, _vp, _u32, _vp, _vp, C.POINTER(_u64)]),
    ("skx_stream_wait", _i, [_vp, _u64]),
    ("skx_stream_drain", _i, [_vp]),
    ("skx_stream_table", _i, [_vp, _vp]),
    ("skx_stream_table_add", _i, [_vp, _vp]),
    ("skx_stream_reset", _i, [_vp]),
    ("skx_stream_reads", _i, [_vp, C.POINTER(_u64)]),
    ("skx_stream_stats", _i, [_vp, C.POINTER(_u64), _u32]),
    ("skx_stream_rank", _i, [_vp, _u32, _vp, _vp]),
    ("skx_stream_destroy", None, [_vp]),
    ("skx_stream_set_profiling", _i, [_vp, _i]),
    ("skx_stream_profile", _i, [_vp, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("skx_stream_scan_alone", _i, [_vp, C.c_uint32, C.POINTER(C.c_double)]),
    ("skx_sketch_reads", _i, [_i, _u32, _u64, _u32, _vp, _vp, _u32, _vp, _vp]),
    ("skx_common_hashes", _i, [_vp, _vp, _vp, _u32, _u32, _vp]),
    ("skx_comm_unique_id", _i, [_vp]),
    ("skx_comm_create", _i, [_pp, _i, _i, _i, _vp]),
    ("skx_comm_n_ranks", _i, [_vp, C.POINTER(_i)]),
    ("skx_stream_allreduce", _i, [_vp, _vp]),
    ("skx_comm_destroy", None, [_vp]),
    ("skx_dev_malloc", _i, [_i, _pp, _sz]),
    ("skx_dev_free", _i, [_i, _vp]),
    ("skx_dev_upload", _i, [_i, _vp, _vp, _sz]),
    ("skx_dev_download", _i, [_i, _vp, _vp, _sz]),
    ("skx_dev_synchronize", _i, [_i]),
    ("skx_dev_mem_info", _i, [_i, C.POINTER(_u64), C.POINTER(_u64)]),
    ("skx_host_alloc", _i, [_i, _pp, _sz]),
    ("skx_host_free", _i, [_i, _vp]),
]

_LIB = None


class SketchyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def load():
    """dlopen the in-tree library and type every entry point; raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built (python -m sketchy_amd.build). "
                "sketchy_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def check(rc):
    if rc != OK:
        raise SketchyHipError(rc, load().skx_last_error().decode())
