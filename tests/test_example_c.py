"""examples/predict_stream.c: the reference's streaming loop against the C ABI from plain C (gcc, no HIP headers, no torch).
Without a GPU: it compiles, links against libsketchy_hip.so and reports that there is no device (exit code 2: no CPU path).
On the GPU: its rows -- read number, best genome, running sum -- equal the oracle's for the same data (the example's genomes and
reads come from a small LCG that is rebuilt here)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from sketchy_amd import build
    build.build()
    exe = str(tmp_path / "predict_stream")
    libdir = os.path.join(ROOT, "sketchy_amd")
    out = subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=c99", "-I", os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "examples", "predict_stream.c"), "-L", libdir, "-lsketchy_hip", f"-Wl,-rpath,{libdir}",
                          "-o", exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_the_c_example_builds_and_has_no_cpu_path(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "no HIP device" in out.stderr and out.stdout == ""


class _Lcg:
    def __init__(self):
        self.s = 0x9E3779B97F4A7C15

    def __call__(self):
        self.s = (self.s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self.s >> 33


@pytest.mark.gpu
def test_the_c_example_prints_the_oracles_rows(gpu, tmp_path):
    from oracle import oracle as orc
    n_g, g_len, s, k, n_reads, r_len = 6, 30000, 200, 16, 240, 600
    lcg = _Lcg()
    acgt = b"ACGT"
    genomes = [bytearray(acgt[lcg() & 3] for _ in range(g_len))]
    for _ in range(1, n_g):
        g = bytearray(genomes[0])
        for i in range(g_len):
            if lcg() % 10 == 0:
                g[i] = acgt[lcg() & 3]
        genomes.append(g)
    cols = [orc.sketch(bytes(g), k, 0, s) for g in genomes]
    col_len = np.array([len(c) for c in cols], np.uint32)
    ref = np.full((n_g, s), np.iinfo(np.uint64).max, np.uint64)
    for i, c in enumerate(cols):
        ref[i, :len(c)] = c
    reads = bytearray()
    for r in range(n_reads):
        at = lcg() % (g_len - r_len)
        reads += genomes[r % n_g][at:at + r_len]
    bases = np.frombuffer(bytes(reads), np.uint8)
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * r_len
    exp = orc.stream(k, 0, int(col_len[0]), ref, col_len, bases, offsets, top_k=1)
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = np.array([[int(x) for x in ln.split("\t")] for ln in out.stdout.strip().splitlines()], np.uint64)
    assert rows.shape == (n_reads, 3)
    np.testing.assert_array_equal(rows[:, 0], np.arange(1, n_reads + 1))
    np.testing.assert_array_equal(rows[:, 1], exp["topk_idx"].reshape(-1))
    np.testing.assert_array_equal(rows[:, 2], exp["topk_sum"].reshape(-1))
    table = np.array([int(x) for x in out.stderr.strip().splitlines()[-1].split(":")[1].split()], np.uint64)
    np.testing.assert_array_equal(table, exp["cum"])
