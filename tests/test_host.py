"""The C++ host (sketchy_amd/host): parsers on CPU, `predict` / `shared` text output on the GPU against rows built
from the oracle.  Output grammar: src/sketchy.rs:99-101 (header), :391-398 (rows)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from helpers import unpack_reads, workload
from mshio import write_msh
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "sketchy_amd", "sketchy-hip")


def _run(*args, stdin=None):
    p = subprocess.run([BIN, *args], input=stdin, capture_output=True)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _fixture(tmp_path, n=40, s=128, n_reads=30, k=16, seed=0):
    ref, bases, offsets = workload(n, s, n_reads, read_len=500, k=k, seed=seed, genome_len=40000, rng_seed=321)
    names = [f"genome{i:03d}.fa" for i in range(n)]
    msh = str(tmp_path / "ref.msh")
    write_msh(msh, names, ref["ref"], kmer=k, seed=seed, lengths=[40000] * n)
    tsv = str(tmp_path / "geno.tsv")
    with open(tsv, "w") as f:
        f.write("id\tmlst\tmeca\tpvl\n")
        for i, nm in enumerate(names):
            f.write(f"{nm}\tST{i % 7}\t{'R' if i % 3 else 'S'}\t{'+' if i % 5 == 0 else '-'}\n")
    reads = unpack_reads(bases, offsets)
    return ref, names, msh, tsv, reads, bases, offsets


def test_binary_is_built():
    from sketchy_amd import build
    build.build()
    assert os.path.exists(BIN)


def test_info_reads_msh_written_by_python(tmp_path):
    ref, names, msh, tsv, reads, _, _ = _fixture(tmp_path)
    rc, out, err = _run("info", "-i", msh)
    assert rc == 0, err
    lines = out.strip().split("\n")
    assert lines == [f"{nm} 40000 128" for nm in names]
    rc, out, err = _run("info", "-i", msh, "-p")
    assert rc == 0 and "sketch_size=128" in out and "kmer_size=16" in out and "seed=0" in out
    write_msh(str(tmp_path / "s42.msh"), names[:2], ref["ref"][:2], kmer=21, seed=42)
    rc, out, err = _run("info", "-i", str(tmp_path / "s42.msh"), "-p")
    assert "kmer_size=21" in out and "seed=42" in out


def test_errors_without_gpu_work(tmp_path):
    rc, out, err = _run("info", "-i", str(tmp_path / "nope.msh"))
    assert rc == 1 and "failed to open" in err
    (tmp_path / "x.txt").write_text("x")
    rc, out, err = _run("info", "-i", str(tmp_path / "x.txt"))
    assert rc == 1 and "must have Mash (.msh) or Finch (.fsh) extension" in err
    ref, names, msh, tsv, reads, _, _ = _fixture(tmp_path)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-c", "-t", "2", "-i", msh)
    assert rc == 1 and "--top must be an odd number" in err  # src/sketchy.rs:74-79


def test_check_subcommand(tmp_path):
    """src/sketchy.rs:212-236: only the length comparison has an effect."""
    ref, names, msh, tsv, reads, _, _ = _fixture(tmp_path)
    rc, out, err = _run("check", "-r", msh, "-g", tsv)
    assert rc == 0 and out == "ok\n"
    with open(tsv, "a") as f:
        f.write("extra.fa\tST1\tR\t-\n")
    rc, out, err = _run("check", "-r", msh, "-g", tsv)
    assert rc == 1 and "must have the same length" in err


def _expected_stream(ref, names, tsv, bases, offsets, top, limit=0, header=False):
    n = len(offsets) - 1 if not limit else min(limit, len(offsets) - 1)
    exp = orc.stream(16, 0, ref["ref"].shape[1], ref["ref"], ref["col_len"], bases, offsets[:n + 1], top_k=top)
    geno = {l.split("\t")[0]: l.rstrip("\n").split("\t")[1:] for l in open(tsv).readlines()[1:]}
    lines = ["reads\tsketch_id\tshared_hashes\tmlst\tmeca\tpvl"] if header else []
    for r in range(n):
        for j in range(top):
            nm = names[exp["topk_idx"][r, j]]
            lines.append(f"{r + 1}\t{nm}\t{exp['topk_sum'][r, j]}\t" + "\t".join(geno[nm]))
    return "\n".join(lines) + "\n"


@pytest.mark.gpu
def test_predict_stream_rows_fastq_fasta_gz_stdin(gpu, tmp_path):
    ref, names, msh, tsv, reads, bases, offsets = _fixture(tmp_path)
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write(f"@read{i} desc\n{r.decode()}\n+\n{'I' * len(r)}\n")
    fa = str(tmp_path / "reads.fa.gz")
    with gzip.open(fa, "wt") as f:
        for i, r in enumerate(reads):
            s = r.decode()
            f.write(f">read{i}\n" + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + "\n")
    want = _expected_stream(ref, names, tsv, bases, offsets, top=3, header=True)
    for src in (fq, fa):
        rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", src, "-t", "3", "-s", "-H", "-b", "7")
        assert rc == 0, err
        assert out == want
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-t", "3", "--stream", "--header", stdin=open(fq, "rb").read())
    assert rc == 0 and out == want
    # --limit (src/sketchy.rs:350-353) and default top = 1
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-l", "11")
    assert rc == 0 and out == _expected_stream(ref, names, tsv, bases, offsets, top=1, limit=11)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-t", "41")
    assert rc == 1 and "--top exceeds" in err


@pytest.mark.gpu
def test_predict_stream_with_sketches_of_unequal_length(gpu, tmp_path):
    """A collection whose first sketch is SHORTER than the others: the reference sketches the reads with s = |sketch 0|
    (src/sketchy.rs:82, :520-527) and still intersects the longer sketches in full; `predict -s` must print the oracle's
    rows (it used to refuse such collections)."""
    ref, bases, offsets = workload(30, 128, 25, read_len=600, genome_len=12000, rng_seed=654)
    names = [f"g{i:02d}.fa" for i in range(30)]
    lens = np.random.default_rng(3).integers(40, 129, size=30)
    lens[0] = 24
    lens[3] = 128
    msh = str(tmp_path / "ragged.msh")
    write_msh(msh, names, [ref["ref"][g, :lens[g]] for g in range(30)], kmer=16, seed=0, lengths=[12000] * 30)
    tsv = str(tmp_path / "geno.tsv")
    with open(tsv, "w") as f:
        f.write("id\tmlst\n")
        for i, nm in enumerate(names):
            f.write(f"{nm}\tST{i % 4}\n")
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(unpack_reads(bases, offsets)):
            f.write(f"@r{i}\n{r.decode()}\n+\n{'I' * len(r)}\n")
    exp = orc.stream(16, 0, 24, ref["ref"], lens.astype(np.uint32), bases, offsets, top_k=2)
    assert exp["topk_sum"][-1, 0] > 0
    want = "".join(f"{r + 1}\t{names[exp['topk_idx'][r, j]]}\t{exp['topk_sum'][r, j]}\tST{exp['topk_idx'][r, j] % 4}\n"
                   for r in range(25) for j in range(2))
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-t", "2", "-s")
    assert rc == 0, err
    assert out == want


@pytest.mark.gpu
def test_predict_offline_pools_all_reads(gpu, tmp_path):
    """Offline mode (src/sketchy.rs:281-315): ONE sketcher over all reads, one ranking."""
    ref, names, msh, tsv, reads, bases, offsets = _fixture(tmp_path)
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write(f"@r{i}\n{r.decode()}\n+\n{'I' * len(r)}\n")
    s = ref["ref"].shape[1]
    for limit in (0, 9):
        use = reads if not limit else reads[:limit]
        pooled = sorted({int(h) for r in use for h in orc.sketch(r, 16, 0, 10 ** 7)})[:s]
        common = np.array([orc.common_hashes(ref["ref"][g], np.array(pooled, np.uint64)) for g in range(len(names))])
        order = orc.stable_rank(common.astype(np.uint64))[:5]
        geno = {l.split("\t")[0]: l.rstrip("\n").split("\t")[1:] for l in open(tsv).readlines()[1:]}
        want = "".join(f"{len(use)}\t{names[g]}\t{common[g]}\t" + "\t".join(geno[names[g]]) + "\n" for g in order)
        args = ["predict", "-r", msh, "-g", tsv, "-i", fq, "-t", "5", "-b", "8"] + (["-l", str(limit)] if limit else [])
        rc, out, err = _run(*args)
        assert rc == 0, err
        assert out == want


@pytest.mark.gpu
def test_shared_subcommand(gpu, tmp_path):
    ref, names, msh, tsv, reads, _, _ = _fixture(tmp_path)
    q = str(tmp_path / "query.msh")
    write_msh(q, names[:3], ref["ref"][:3])
    rc, out, err = _run("shared", "-r", msh, "-q", q)
    assert rc == 0, err
    want = "".join(f"{names[r]} {names[i]} {orc.common_hashes(ref['ref'][r], ref['ref'][i])}\n"
                   for r in range(len(names)) for i in range(3))
    assert out == want
    assert f"{names[1]} {names[1]} 128\n" in out  # docs/index.md:145-149


def test_msh_python_roundtrip(tmp_path):
    from mshio import read_msh
    rng = np.random.default_rng(3)
    hs = [np.sort(rng.integers(0, 2 ** 63, size=n, dtype=np.uint64)) for n in (5, 0, 17)]
    write_msh(str(tmp_path / "a.msh"), ["x.fa", "y", "zzzzzzzzz.fasta"], hs, kmer=21, seed=7, lengths=[10, 0, 2 ** 33])
    k, seed, recs = read_msh(str(tmp_path / "a.msh"))
    assert (k, seed) == (21, 7) and [r["name"] for r in recs] == ["x.fa", "y", "zzzzzzzzz.fasta"]
    assert recs[2]["length"] == 2 ** 33
    for r, h in zip(recs, hs):
        np.testing.assert_array_equal(r["hashes"], h)


@pytest.mark.gpu
def test_sketch_subcommand_builds_msh(gpu, tmp_path):
    """`sketchy sketch` (src/sketchy.rs:128-167, :465-494): one sketch per FILE over all of its records, name = file
    name; genomes go through the device's long-read kernels.  Checked hash by hash against the oracle."""
    from mshio import read_msh
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    files, contigs_of = [], []
    for gi, sizes in enumerate(([60000, 3000, 10], [25000] * 4, [900])):  # multi-contig, a contig shorter than k, a tiny genome
        contigs = [bytes(acgt[rng.integers(0, 4, n)]) for n in sizes]
        if gi == 0:
            contigs[1] = contigs[1][:1000] + b"NNNNNRYK" + contigs[1][1000:].lower()  # ambiguity codes break k-mers; case folds
        path = str(tmp_path / f"genome{gi}.fa") + (".gz" if gi == 1 else "")
        opener = gzip.open if gi == 1 else open
        with opener(path, "wt") as f:
            for ci, c in enumerate(contigs):
                s = c.decode()
                f.write(f">contig{ci} some description\n" + "\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + "\n")
        files.append(path); contigs_of.append(contigs)
    out = str(tmp_path / "db.msh")
    for s, k, seed in ((1000, 16, 0), (64, 21, 5)):
        rc, so, err = _run("sketch", "-i", *files, "-o", out, "-s", str(s), "-k", str(k), "-e", str(seed))
        assert rc == 0, err
        kk, sd, recs = read_msh(out)
        assert (kk, sd) == (k, seed)
        assert [r["name"] for r in recs] == [os.path.basename(p) for p in files]
        for r, contigs in zip(recs, contigs_of):
            want = sorted({int(h) for c in contigs for h in orc.sketch(c, k, seed, 10 ** 7)})[:s]
            np.testing.assert_array_equal(r["hashes"], np.array(want, np.uint64))
            assert r["length"] == sum(len(c) for c in contigs)
            valid = sum(sum(1 for i in range(len(c) - k + 1) if all(ch in b"ACGTacgt" for ch in c[i:i + k])) for c in contigs if len(c) < 5000) \
                if max(len(c) for c in contigs) < 5000 else None
            if valid is not None:
                assert r["num_valid_kmers"] == valid
    # paths on stdin when -i is absent (src/sketchy.rs:137-146); the result feeds `info` and `shared`
    rc, so, err = _run("sketch", "-o", out, "-s", "200", stdin="\n".join(files).encode() + b"\n")
    assert rc == 0, err
    rc, so, err = _run("info", "-i", out)
    assert rc == 0 and so.split("\n")[0] == f"genome0.fa {60000 + 3008 + 10} 200"
    rc, so, err = _run("shared", "-r", out, "-q", out)
    assert rc == 0 and "genome0.fa genome0.fa 200\n" in so and "genome0.fa genome1.fa.gz 0\n" in so
    rc, so, err = _run("sketch", "-i", files[0], "-o", str(tmp_path / "x.fsh"))
    assert rc == 1 and "outside the accelerated path" in err
    rc, so, err = _run("sketch", "-i", files[0], "-o", str(tmp_path / "x.txt"))
    assert rc == 1 and "extension" in err


@pytest.mark.gpu
def test_predict_stream_many_batches_through_the_pipeline(gpu, tmp_path):
    """Reader -> device -> writer pipeline: 700 reads in batches of 64 (11 batches over 3 recycled buffers)."""
    ref, bases, offsets = workload(60, 200, 700, read_len=400, genome_len=40000, rng_seed=555)
    names = [f"g{i}" for i in range(60)]
    msh, tsv, fq = str(tmp_path / "r.msh"), str(tmp_path / "g.tsv"), str(tmp_path / "reads.fq")
    write_msh(msh, names, ref["ref"])
    with open(tsv, "w") as f:
        f.write("id\ta\tb\n" + "".join(f"{n}\tA{i % 4}\tB{i % 3}\n" for i, n in enumerate(names)))
    with open(fq, "w") as f:
        for i, r in enumerate(unpack_reads(bases, offsets)):
            f.write(f"@r{i}\n{r.decode()}\n+\n{'#' * len(r)}\n")
    want = _expected_stream(ref, names, tsv, bases, offsets, top=2)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-t", "2", "-b", "64")
    assert rc == 0, err
    assert out == want
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-t", "2", "-b", "64", "-l", "130")
    assert rc == 0 and out == _expected_stream(ref, names, tsv, bases, offsets, top=2, limit=130)
    (tmp_path / "bad.fq").write_text("@r0\nACGT\n+\nIIII\n@r1\nACGT\nIIII\n")  # malformed second record: error, no hang
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", str(tmp_path / "bad.fq"), "-s")
    assert rc == 1 and "malformed FASTQ" in err
