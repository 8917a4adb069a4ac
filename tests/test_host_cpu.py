"""The C++ host's front-end on the CPU, under AddressSanitizer + UndefinedBehaviorSanitizer: `.msh` / genotype / FASTX parsers and
the whole `predict -s` pipeline (mapped-file chunking, parallel parse + 4-bit packing, ordered submit, ordered rows), linked against
tests/stub/skx_stub.cpp instead of the HIP library.  The stub's row of a read is (n_bases mod n_genomes, FNV-1a of the read's 4-bit
codes as they reached the C ABI), so the printed text pins which bases of which read arrived, in which order -- recomputed here
from the sequences.  (What the device makes of them is the GPU tests' business: tests/test_host.py.)

Fuzzing: truncated and bit-flipped `.msh`, FASTQ and TSV files must end in exit code 1 with a message (or 0 when the damage left
a valid file) -- never in a sanitizer report, a crash or a hang (src/sketchy.rs:497-536 panics on some of
these, the host must not)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from mshio import write_msh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "stub", "sketchy-hip-asan")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module")
def asan_bin():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "stub")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return BIN


def _run(*args, stdin=None, timeout=120):
    p = subprocess.run([BIN, *args], input=stdin, capture_output=True, env=ENV, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


CODE = np.full(256, 4, np.uint8)
for _ch, _v in zip(b"ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3]):
    CODE[_ch] = _v


def _fnv(seq: bytes) -> int:
    h = 1469598103934665603
    for c in CODE[np.frombuffer(seq, np.uint8)].tolist():
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h >> 8


def _expected(reads, names, geno, top, limit=0, header=None):
    n = len(reads) if not limit else min(limit, len(reads))
    lines = [header] if header else []
    for r in range(n):
        for t in range(top):
            g = (len(reads[r]) + t) % len(names)
            lines.append(f"{r + 1}\t{names[g]}\t{_fnv(reads[r])}\t" + "\t".join(geno[names[g]]))
    return "\n".join(lines) + ("\n" if lines else "")


def _reference(tmp_path, n=23):
    rng = np.random.default_rng(5)
    names = [f"genome{i:02d}.fa" for i in range(n)]
    hs = [np.sort(rng.choice(2 ** 40, size=int(rng.integers(3, 40)), replace=False).astype(np.uint64)) for _ in range(n)]
    msh = str(tmp_path / "ref.msh")
    write_msh(msh, names, hs, kmer=16, seed=0, lengths=[1000] * n)
    tsv = str(tmp_path / "geno.tsv")
    geno = {}
    with open(tsv, "w") as f:
        f.write("id\tmlst\tmeca\n")
        for i, nm in enumerate(names):
            geno[nm] = [f"ST{i % 5}", "R" if i % 2 else "S"]
            f.write(f"{nm}\t" + "\t".join(geno[nm]) + "\n")
    return names, geno, msh, tsv


def _reads(n, lo, hi, seed=1, dirty=True):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        ln = int(rng.integers(lo, hi + 1))
        s = alpha[rng.integers(0, 4, ln)].copy()
        if dirty and ln > 8 and i % 3 == 0:
            s[rng.integers(0, ln, 3)] = np.frombuffer(b"NnR", np.uint8)          # not bases: code 4
            s[ln // 2:] = np.char.lower(s[ln // 2:].tobytes().decode()).encode() if False else s[ln // 2:] | 0x20  # lower case folds
        out.append(s.tobytes())
    return out


def _fastq(path, reads, crlf=False, opener=open):
    nl = "\r\n" if crlf else "\n"
    with opener(path, "wt", newline="") as f:
        for i, r in enumerate(reads):
            q = "@" * len(r) if i % 4 == 1 else "I" * len(r)   # quality lines that START with '@': the chunker must not take them for headers
            f.write(f"@read{i} x{nl}{r.decode()}{nl}+{nl}{q}{nl}")


def _fasta(path, reads, width=60, opener=open):
    with opener(path, "wt") as f:
        for i, r in enumerate(reads):
            s = r.decode()
            f.write(f">read{i}\n" + "".join(s[j:j + width] + "\n" for j in range(0, len(s), width)))


def test_stream_rows_from_a_mapped_fastq_in_many_chunks(asan_bin, tmp_path):
    """Reads of 1 .. 900 bases (odd lengths: reads start on odd nibbles of the packed stream), N / IUPAC / lower case, quality
    lines starting with '@', CR LF line ends; 16-read chunks parsed by 4 threads, rows in file order."""
    names, geno, msh, tsv = _reference(tmp_path)
    reads = _reads(500, 1, 900)
    want = _expected(reads, names, geno, top=2, header="reads\tsketch_id\tshared_hashes\tmlst\tmeca")
    for crlf in (False, True):
        fq = str(tmp_path / f"reads{int(crlf)}.fq")
        _fastq(fq, reads, crlf=crlf)
        rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-t", "2", "-H", "-b", "16", "-j", "6")
        assert rc == 0, err
        assert out == want
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-b", "16", "-j", "6", "-l", "137")
    assert rc == 0 and out == _expected(reads, names, geno, top=1, limit=137)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-b", "100000", "--timing")   # one chunk, default threads
    assert rc == 0 and out == _expected(reads, names, geno, top=1)
    assert '"sketchy_hip_timing"' in err and '"reads": 500' in err and "mapped fastq" in err


def test_stream_rows_from_fasta_gzip_and_stdin(asan_bin, tmp_path):
    names, geno, msh, tsv = _reference(tmp_path)
    reads = _reads(300, 0, 400, seed=9)
    reads[7] = b""            # a record without sequence: still a read (src/sketchy.rs:349-350)
    reads[-1] = b""
    want = _expected(reads, names, geno, top=1)
    fa = str(tmp_path / "reads.fa")
    _fasta(fa, reads)         # multi-line, mapped: records are assembled from their lines
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fa, "-s", "-b", "9", "-j", "5")
    assert rc == 0, err
    assert out == want
    gz = str(tmp_path / "reads.fa.gz")
    _fasta(gz, reads, opener=gzip.open)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", gz, "-s", "-b", "9", "--timing")
    assert rc == 0 and out == want and "streamed" in err
    fq = str(tmp_path / "reads.fq")
    _fastq(fq, [r for r in reads if r])   # (FASTQ through stdin: the sequential reader)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-s", "-b", "11", "-l", "40", stdin=open(fq, "rb").read())
    assert rc == 0 and out == _expected([r for r in reads if r], names, geno, top=1, limit=40)


def test_stream_rows_from_bgzf_inflated_in_parallel(asan_bin, tmp_path):
    """A BGZF file (bgzip: gzip members of <= 64 KB with their sizes in the header) is inflated by all threads at once and then cut
    and parsed like an uncompressed file; plain gzip of the same reads is one sequential stream.  Same rows, byte for byte; a BGZF
    file cut in the middle of a member ends with a message."""
    from helpers import write_bgzf
    names, geno, msh, tsv = _reference(tmp_path)
    reads = _reads(700, 1, 900, seed=21)
    want = _expected(reads, names, geno, top=2)
    fq = str(tmp_path / "reads.fq")
    _fastq(fq, reads)
    raw = open(fq, "rb").read()
    bg = str(tmp_path / "reads.fq.gz")
    for block in (65280, 777):   # (members far smaller than a record line: records span many members)
        write_bgzf(bg, raw, block=block)
        assert gzip.open(bg).read() == raw
        rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", bg, "-s", "-t", "2", "-b", "16", "-j", "6", "--timing")
        assert rc == 0, err
        assert out == want and "bgzf fastq" in err
    plain = str(tmp_path / "plain.fq.gz")
    with gzip.open(plain, "wb") as f:
        f.write(raw)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", plain, "-s", "-t", "2", "-b", "16", "--timing")
    assert rc == 0 and out == want and "streamed" in err
    cut = str(tmp_path / "cut.fq.gz")
    open(cut, "wb").write(open(bg, "rb").read()[:-400])
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", cut, "-s", "-b", "16")
    assert rc in (0, 1) and "Sanitizer" not in err   # (a truncated tail: either the streaming reader's error or the reads that were whole)


def test_fast_inflate_against_zlib_under_sanitizers():
    """sketchy_amd/host/fast_inflate.hpp: stored / fixed / dynamic blocks of every zlib strategy and level, sizes around the fast loop's
    margins, skewed alphabets (15-bit codes, subtables) -- byte-identical to the input, wrong output sizes refused; 20 000 mutated
    and truncated members end in `false` or garbage of the right length (the CRC catches that), never in a sanitizer report;
    crc32_fast == zlib's crc32 for every length class."""
    exe = os.path.join(ROOT, "tests", "stub", "inflate-check-asan")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "stub"), exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    p = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=900)
    assert p.returncode == 0 and " fails 0;" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_bgzf_members_are_checked_against_their_crc(asan_bin, tmp_path):
    """A BGZF member whose bytes were damaged but still inflate to the right length is refused (CRC32 of the member's trailer; needletail /
    flate2 do the same, src/sketchy.rs:89-92); a file without BGZF's empty end-of-file member is scored with a warning; members written
    at other compression levels (stored blocks at level 0, longer matches at 9) give the same rows."""
    from helpers import write_bgzf
    names, geno, msh, tsv = _reference(tmp_path)
    reads = _reads(400, 1, 2000, seed=33)
    want = _expected(reads, names, geno, top=1)
    fq = str(tmp_path / "r.fq")
    _fastq(fq, reads)
    raw = open(fq, "rb").read()
    bg = str(tmp_path / "r.fq.gz")
    for level in (0, 6, 9):
        write_bgzf(bg, raw, block=20000, level=level)
        rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", bg, "-s", "-b", "32", "-j", "4")
        assert rc == 0 and out == want, (level, err)
    write_bgzf(bg, raw, block=20000, level=0)   # stored blocks: a flipped payload byte still "inflates"
    data = bytearray(open(bg, "rb").read())
    data[200] ^= 0x01
    bad = str(tmp_path / "bad.fq.gz")
    open(bad, "wb").write(bytes(data))
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", bad, "-s", "-b", "32")
    assert rc == 1 and "CRC" in err and "Sanitizer" not in err, (rc, err)
    noeof = str(tmp_path / "noeof.fq.gz")
    write_bgzf(bg, raw, block=20000, level=6)
    open(noeof, "wb").write(open(bg, "rb").read()[:-28])
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", noeof, "-s", "-b", "32")
    assert rc == 0 and out == want and "end-of-file block" in err, err


def test_a_chunk_with_far_more_reads_than_the_first_records_promised(asan_bin, tmp_path):
    """Chunks are sized from the first 64 records; when the rest of the file holds 100-fold shorter reads a chunk carries far
    more reads than its slot holds: the overflow goes through heap batches and the spill slot, rows and order unchanged."""
    names, geno, msh, tsv = _reference(tmp_path)
    reads = _reads(70, 1800, 2000, seed=3, dirty=False) + _reads(3000, 5, 25, seed=4)
    fq = str(tmp_path / "reads.fq")
    _fastq(fq, reads)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-b", "8", "-j", "4", "--timing")
    assert rc == 0, err
    assert out == _expected(reads, names, geno, top=1)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", fq, "-s", "-b", "8", "-j", "4", "-l", "1234")
    assert rc == 0 and out == _expected(reads, names, geno, top=1, limit=1234)


def test_malformed_inputs_end_with_a_message(asan_bin, tmp_path):
    names, geno, msh, tsv = _reference(tmp_path)
    cases = {
        "@r0\nACGT\n+\nIIII\n@r1\nACGT\nIIII\n": "malformed FASTQ",
        "@r0\nACGT\n+\nIIII\n@r1\nACGT\n+\n": "truncated FASTQ",
        "@r0\nACGT\n+\nIIII\n@r1\n": "truncated FASTQ",
        "hello\nworld\n": "neither FASTA nor FASTQ",
        ">r0\nACGT\n@r1\nAC\n+\nII\n": None,   # a FASTA record whose sequence lines look odd: still one record, no error
    }
    for i, (text, msg) in enumerate(cases.items()):
        p = tmp_path / f"bad{i}.fx"
        p.write_text(text)
        for extra in ((), ("-b", "1", "-j", "3")):
            rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", str(p), "-s", *extra)
            if msg is None:
                assert rc == 0, err
            else:
                assert rc == 1 and msg in err and err.startswith("Error:"), (text, rc, err)
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", str(tmp_path / "nope.fq"), "-s")
    assert rc == 1 and "failed to open Fastx file" in err
    (tmp_path / "empty.fq").write_text("")
    rc, out, err = _run("predict", "-r", msh, "-g", tsv, "-i", str(tmp_path / "empty.fq"), "-s")
    assert rc == 0 and out == ""


def test_a_mash_sketch_with_32_bit_hashes_fails_loudly(asan_bin, tmp_path):
    """A genuine Mash k <= 16 sketch stores hashes32; finch's reader (and this one) takes hashes64 only and would see EMPTY
    sketches.  The host must say so instead of scoring every read 0 against everything."""
    names, geno, msh, tsv = _reference(tmp_path)
    rng = np.random.default_rng(2)
    hs = [np.sort(rng.choice(2 ** 32, size=20, replace=False).astype(np.uint64)) for _ in names]
    m32 = str(tmp_path / "mash32.msh")
    write_msh(m32, names, hs, kmer=16, seed=42, hashes32=True)
    for args in (("info", "-i", m32), ("check", "-r", m32, "-g", tsv), ("predict", "-r", m32, "-g", tsv, "-i", m32, "-s")):
        rc, out, err = _run(*args)
        assert rc == 1 and "32-bit hashes" in err and "hashes64" in err, (args, rc, err)


def test_fuzzed_files_never_crash_the_parsers(asan_bin, tmp_path):
    names, geno, msh, tsv = _reference(tmp_path)
    reads = _reads(40, 10, 200, seed=12)
    fq = str(tmp_path / "reads.fq")
    _fastq(fq, reads)
    rng = np.random.default_rng(77)
    originals = {"msh": open(msh, "rb").read(), "tsv": open(tsv, "rb").read(), "fq": open(fq, "rb").read()}
    n_bad = 0
    for kind, raw in originals.items():
        for trial in range(60):
            data = bytearray(raw)
            if trial % 2 == 0:
                data = data[:int(rng.integers(0, len(data)))]                       # truncated
            else:
                for _ in range(int(rng.integers(1, 9))):
                    data[int(rng.integers(0, len(data)))] ^= 1 << int(rng.integers(0, 8))   # bit flips
            p = str(tmp_path / f"fuzz.{kind}")
            open(p, "wb").write(bytes(data))
            a_msh, a_tsv, a_fq = (p if kind == "msh" else msh), (p if kind == "tsv" else tsv), (p if kind == "fq" else fq)
            for args in ((["info", "-i", a_msh], ["check", "-r", a_msh, "-g", a_tsv]) if kind != "fq" else ()) + \
                        (["predict", "-r", a_msh, "-g", a_tsv, "-i", a_fq, "-s", "-b", "5", "-j", "3"],):
                rc, out, err = _run(*args)
                assert rc in (0, 1), (kind, trial, args, rc, err[-2000:])        # 98 / 99: a sanitizer report; < 0: a signal
                assert "Sanitizer" not in err and "runtime error" not in err, err[-2000:]
                if rc == 1:
                    assert err.startswith("Error:"), err[-500:]
                    n_bad += 1
    assert n_bad > 40   # (most damage is detected; some leaves a valid file)
    # the damage every reader must refuse
    open(str(tmp_path / "t.msh"), "wb").write(originals["msh"][:len(originals["msh"]) // 2])
    rc, out, err = _run("info", "-i", str(tmp_path / "t.msh"))
    assert rc == 1 and err.startswith("Error:")
