"""Builds libsketchy_hip.so (hand-written gfx950 kernels + the C ABI) in-tree with hipcc.

The library is the product: there is no CPU or PyTorch fallback behind it.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsketchy_hip.so")
# the same sources with -DSKX_EXPERIMENTS: the environment knobs of tests/ and tools/ (kernel variants, pass sizes,
# pipeline depths, ablations) exist only here; selected with SKX_LIB_PATH, never loaded by default
LIB_EXP = os.path.join(HERE, "libsketchy_hip_exp.so")
SOURCES = ["skx_kernels.hip", "skx_capi.hip"]
ARCH = "gfx950"


def source_sha():
    """sha256 (first 16 hex digits) over the kernel / C-ABI sources the library is built from: what ties a committed profile
    (profiles/scan_traffic.json, profiles/valu_insts.json) to the code that produced it -- bench.py marks a quoted figure
    `stale` when the tree has moved on since."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [os.path.join("..", "..", "include", "sketchy_hip.h")]
    for f in names:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _build_lib(lib, objdir, extra, force, verbose):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "sketchy_hip.h"))
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-result", *extra]
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + headers):
            cmd = [_hipcc(), *flags, "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or _stale(lib, objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib, "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library (and its experiments twin)."""
    _build_lib(LIB, os.path.join(HERE, "build"), [], force, verbose)
    _build_lib(LIB_EXP, os.path.join(HERE, "build", "exp"), ["-DSKX_EXPERIMENTS"], force, verbose)
    build_host(force=force, verbose=verbose)
    return LIB


HOST_BIN = os.path.join(HERE, "sketchy-hip")


def build_host(force=False, verbose=False):
    """The C++ host (sketchy_amd/host): `sketchy-hip predict|shared|info` above the C ABI."""
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, "sketchy_host.cpp")]
    deps = srcs + [os.path.join(hdir, "formats.hpp"), os.path.join(ROOT, "include", "sketchy_hip.h"), LIB]
    if force or _stale(HOST_BIN, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", hdir, *srcs, "-o", HOST_BIN,
               "-L", HERE, "-lsketchy_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
