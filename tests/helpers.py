"""Shared builders for the parity tests: seeded workloads handed to BOTH the oracle and the HIP path."""
import os

import numpy as np

from sketchy_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "sketchy_amd", "libsketchy_hip_exp.so")


def exp_env(**knobs):
    """Environment of a subprocess that loads the EXPERIMENTS build of the library (-DSKX_EXPERIMENTS: the only build that
    reads SKX_* knobs) with the given knobs set; the product library ignores every one of them."""
    assert os.path.exists(EXP_LIB), f"{EXP_LIB} not built (python -m sketchy_amd.build)"
    return dict(os.environ, SKX_LIB_PATH=EXP_LIB, **{k: str(v) for k, v in knobs.items()})


def pack_reads(reads):
    """list[bytes] -> (bases uint8, offsets uint64)"""
    offsets = np.zeros(len(reads) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in reads]).astype(np.uint64)
    bases = np.frombuffer(b"".join(reads), np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return bases, offsets


def unpack_reads(bases, offsets):
    b = bases.tobytes()
    return [b[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]


def workload(n_genomes, s, n_reads, read_len=1500, k=16, seed=0, genome_len=0, rng_seed=1, err=0.05, **kw):
    ref = synth.make_reference(n_genomes, s, k=k, hash_seed=seed, genome_len=genome_len, rng_seed=rng_seed,
                               device="numpy", **kw)
    bases, offsets = synth.make_reads(ref["genome"], n_reads, read_len, err=err, rng_seed=rng_seed + 1)
    return ref, bases, offsets


def assert_stream_equal(got, exp, top, what=""):
    if top:
        np.testing.assert_array_equal(got["topk_sum"], exp["topk_sum"], err_msg=f"{what} topk_sum")
        np.testing.assert_array_equal(got["topk_idx"], exp["topk_idx"], err_msg=f"{what} topk_idx")
    if got.get("shared") is not None and exp.get("shared") is not None:
        np.testing.assert_array_equal(got["shared"], exp["shared"], err_msg=f"{what} per-read shared")
    if got.get("sketches") is not None and exp.get("sketches") is not None:
        np.testing.assert_array_equal(got["sketch_len"], exp["sketch_len"], err_msg=f"{what} sketch_len")
        np.testing.assert_array_equal(got["sketches"], exp["sketches"], err_msg=f"{what} sketches")


def workload_species(sizes, s, n_reads, read_len=1500, k=16, seed=0, genome_len=0, rng_seed=1, err=0.05, lognormal_sigma=0.0,
                     max_len=50000):
    """Several species (reference collections with unrelated ancestors) and ONE read stream sampled from all of their
    ancestors, shuffled: (refs [list of make_reference dicts], bases, offsets)."""
    refs = [synth.make_reference(n, s, k=k, hash_seed=seed, genome_len=genome_len, rng_seed=rng_seed + 17 * i, device="numpy")
            for i, n in enumerate(sizes)]
    per = [n_reads // len(sizes) + (1 if i < n_reads % len(sizes) else 0) for i in range(len(sizes))]
    reads = []
    for i, (r, m) in enumerate(zip(refs, per)):
        b, o = synth.make_reads(r["genome"], m, read_len, err=err, rng_seed=rng_seed + 1000 + i, lognormal_sigma=lognormal_sigma,
                                max_len=max_len)
        reads += unpack_reads(b, o)
    order = np.random.default_rng(rng_seed).permutation(len(reads))
    bases, offsets = pack_reads([reads[i] for i in order])
    return refs, bases, offsets
