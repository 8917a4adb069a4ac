"""The C-ABI library builds, loads and exports every symbol include/sketchy_hip.h declares.
No compute calls here (no GPU in the CPU container): only entry points that need no device."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sketchy_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(skx_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_loads():
    from sketchy_amd import _lib, build
    build.build()
    lib = _lib.load()
    assert b"gfx950" in lib.skx_version()


def test_every_declared_symbol_is_exported_and_bound():
    from sketchy_amd import _lib
    lib = _lib.load()
    decl = declared_symbols()
    assert len(decl) >= 30
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(decl) == bound, (set(decl) ^ bound)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (skx_[a-z0-9_]+)", out))
    assert set(decl) <= exported, set(decl) - exported
    for n in decl:
        assert hasattr(lib, n)


def test_product_library_has_no_environment_knobs():
    """The product .so neither imports getenv nor carries the name of any SKX_* knob; the experiments twin
    (-DSKX_EXPERIMENTS, loaded only through SKX_LIB_PATH by tests/ and tools/) does both and exports the same ABI."""
    from sketchy_amd import _lib, build
    assert os.path.realpath(_lib.LIB_PATH) == os.path.realpath(build.LIB) or os.environ.get("SKX_LIB_PATH")
    und = subprocess.check_output(["nm", "-D", "--undefined-only", build.LIB], text=True)
    assert "getenv" not in und
    blob = open(build.LIB, "rb").read()
    names = set(re.findall(rb"SKX_[A-Z][A-Z0-9_]{3,}", blob)) - {b"SKX_MAX_TOP"}  # (an error message quotes the header's macro)
    assert not names, sorted(names)[:5]
    und_exp = subprocess.check_output(["nm", "-D", "--undefined-only", build.LIB_EXP], text=True)
    assert "getenv" in und_exp
    blob_exp = open(build.LIB_EXP, "rb").read()
    assert b"SKX_SCAN_ABLATE" in blob_exp and b"SKX_NO_FILTER" in blob_exp
    exp = set(re.findall(r"\bT (skx_[a-z0-9_]+)", subprocess.check_output(["nm", "-D", "--defined-only", build.LIB_EXP], text=True)))
    assert set(declared_symbols()) <= exp


def test_library_contains_gfx950_code_object():
    from sketchy_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"scan_kernel" in blob


def test_no_cpu_fallback_without_device():
    """Without a device the product path fails loudly instead of computing on the host."""
    import ctypes as C
    from sketchy_amd import _lib
    lib = _lib.load()
    if lib.skx_device_count() > 0:
        pytest.skip("a device is present")
    p = C.c_void_p()
    assert lib.skx_dev_malloc(0, C.byref(p), 16) == _lib.ERR_NO_DEVICE
    assert b"no HIP device" in lib.skx_last_error()
    import numpy as np
    from sketchy_amd import api
    with pytest.raises(_lib.SketchyHipError) as e:
        api.ReferenceSketch(np.arange(8, dtype=np.uint64).reshape(1, 8))
    assert e.value.code == _lib.ERR_NO_DEVICE


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under sketchy_amd/ or include/ may reference it."""
    bad = []
    for base in ("sketchy_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in fn:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|orc_", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_pack_bases_helper_matches_the_normalisation_rules():
    """skx_pack_bases (host-side, no device): ACGTU in either case -> 0..3, any other retained byte -> 4, whitespace dropped;
    two bases per byte, low nibble first; appending at an odd nibble keeps the nibble already there."""
    import ctypes as C
    import numpy as np
    from sketchy_amd import _lib
    L = _lib.load()
    text = b"ACGTUacgtuNnRy-.* \t\r\nXA"
    src = np.frombuffer(text, np.uint8)
    out = np.full(32, 0xEE, np.uint8)
    out[0] = 0x07  # an earlier base in the low nibble of byte 0
    pos = int(L.skx_pack_bases(src.ctypes.data_as(C.c_void_p), len(src), out.ctypes.data_as(C.c_void_p), 1))
    kept = [c for c in text if c not in b" \t\r\n"]
    assert pos == 1 + len(kept)
    codes = [(int(out[i >> 1]) >> (4 * (i & 1))) & 0xF for i in range(pos)]
    assert codes[0] == 7
    want = [{65: 0, 67: 1, 71: 2, 84: 3, 85: 3}.get(c & 0xDF if chr(c).isalpha() else c, 4) for c in kept]
    assert codes[1:] == want
