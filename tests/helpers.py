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


_SNP_CACHE = {}


def workload_snp(n_genomes, s, n_reads, read_len=1500, k=16, seed=0, genome_len=0, rng_seed=1, err=0.05, device="cpu",
                 lognormal_sigma=0.0, max_len=50000, source="truth", **kw):
    """SURVEY.md 8(d)'s generator (synth.make_reference(mode="snp")): SNP clone tree with real k-mer hashes, reads drawn
    from ONE truth strain (source="truth") or from the ancestor.  It needs torch, and a process that has loaded the HIP
    library must not import torch afterwards, so the arrays are made by a child process and cached for the session.
    Returns (ref dict incl. truth_index / lineage, bases, offsets)."""
    import subprocess
    import sys
    import tempfile
    key = (n_genomes, s, n_reads, read_len, k, seed, genome_len, rng_seed, err, device, lognormal_sigma, max_len, source,
           tuple(sorted(kw.items())))
    if key in _SNP_CACHE:
        return _SNP_CACHE[key]
    with tempfile.TemporaryDirectory(prefix="skx_snp_") as d:
        out = os.path.join(d, "w.npz")
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from sketchy_amd import synth\n"
            "dev = %r\n"
            "dev = ('cuda' if torch.cuda.is_available() else 'cpu') if dev == 'auto' else dev\n"
            "ref = synth.make_reference(%d, %d, k=%d, hash_seed=%d, genome_len=%d, rng_seed=%d, mode='snp', device=dev, **%r)\n"
            "src = ref['truth_genome'] if %r == 'truth' else ref['genome']\n"
            "b, o = synth.make_reads(src, %d, %d, err=%r, rng_seed=%d, lognormal_sigma=%r, max_len=%d)\n"
            "np.savez(%r, ref=ref['ref'], col_len=ref['col_len'], genome=ref['genome'], truth_genome=ref['truth_genome'],\n"
            "         lineage=ref['lineage'], truth_index=ref['truth_index'], bases=b, offsets=o)\n"
        ) % (ROOT, device, n_genomes, s, k, seed, genome_len, rng_seed, kw, source, n_reads, read_len, err, rng_seed + 1,
             lognormal_sigma, max_len, out)
        subprocess.check_call([sys.executable, "-c", code])
        z = np.load(out)
        ref = dict(ref=z["ref"], col_len=z["col_len"], genome=z["genome"], truth_genome=z["truth_genome"], lineage=z["lineage"],
                   truth_index=int(z["truth_index"]), k=k, seed=seed, s=s)
        res = (ref, z["bases"], z["offsets"])
    _SNP_CACHE[key] = res
    return res


from sketchy_amd.synth import write_bgzf  # noqa: E402,F401  (BGZF writer: bench.py's gz leg uses the same)
