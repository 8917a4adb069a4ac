#!/usr/bin/env python3
"""bench.py -- reads/sec of the streaming read-vs-reference MinHash path on MI355X.

One "step" = one skx_stream_push_device of a batch of B synthetic reads that are already
resident in HBM: sketch every read, score it against every genome of the resident reference
sketch(es) (one scan of the s x N matrix), update the running sum-of-shared-hashes table and
rank the top row(s) after every read -- the whole loop body of the reference's
_sum_of_shared_hashes (src/sketchy.rs:328-354), nothing skipped.

Workloads (BASELINE.json configs): default "c2" = ~100k x 1.5 kb reads per step vs a 40 000-genome
s=10 000 k=16 reference (3.2 GB of hashes, the HBM-bound scan the metric is quoted on); "c4" = five
species' reference sketches resident together (150 000 genomes, 12 GB) and a mixed-length stream.

The timed region is a stream that starts from a FRESH table (skx_stream_reset after the warm-up)
and runs exactly K steps of distinct batches + the final table all-reduce.  What is timed is also
what is checked: the rows the first timed step wrote are compared with the CPU oracle, the rows
of the last timed step with the same reads pushed in 4096-read cuts into a second stream seeded
with the table before that step; any mismatch makes the run fail (parity_error, exit code 1).

N > 1: one process per GPU (`python bench.py --gpus N` starts them itself through
torch.distributed.run when WORLD_SIZE is unset), reference replicated, the read stream sharded; the
only exchange is the final RCCL all-reduce of the u64 table (inside the timed region).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {
    # name: (genomes per species, s, read_len, lognormal sigma, description)
    "c0": ([500], 1000, 1500, 0.0, "C0: 1.5 kb reads vs 500-genome s=1000 k=16 sketch (plumbing)"),
    "c1": ([5000], 1000, 1500, 0.0, "C1: 1.5 kb reads vs 5k-genome s=1000 k=16 sketch (cache-resident)"),
    "c2": ([40000], 10000, 1500, 0.0, "C2: ~100k x 1.5 kb reads per step vs 40000-genome s=10000 k=16 sketch (HBM-bound scan)"),
    "c6g": ([40000, 35000], 10000, 1500, 0.0, "experiment: two species resident (75000 genomes, 6 GB), 1.5 kb reads (scan size sweep)"),
    "c4s": ([1600, 1600, 1200, 1000, 600], 1000, 1500, 1.0,
            "C4 in small (tests): 5 species' sketches resident (6000 genomes, s=1000 k=16), log-normal read lengths, a stream mixed over "
            "all 5, every read scored against all 5"),
    "c4": ([40000, 40000, 30000, 25000, 15000], 10000, 1500, 1.0,
           "C4: 5 species' sketches resident (150000 genomes, s=10000 k=16), log-normal read lengths 200..50000 (median 1.5 kb), "
           "every read scored against all 5"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CHECK_CUT = 4096       # reads per push of the verification stream


def _profile_key(config, batch, workload):
    """key of a committed profile in profiles/scan_traffic.json / valu_insts.json: per config, batch AND workload (round 5 quoted the
    truth-strain run's traffic beside the ancestor stream's line)"""
    return f"{config}{'truth' if workload == 'truth' else ''}_b{batch}"


def _pmc_traffic(config, batch, workload="truth"):
    """HBM bytes per scan_kernel launch from the COMMITTED rocprofv3 PMC passes (profiles/scan_traffic.json; FETCH_SIZE
    x 1024 x 2 [gfx950 counts wide streaming reads at half their bytes] + WRITE_SIZE x 1024) for the same config and
    batch -- not measured in this run; (None, None) when no matching profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "scan_traffic.json")) as f:
            t = json.load(f)
        key = _profile_key(config, batch, workload)
        e = t.get(key)
        if e:
            return (e.get("bytes_per_launch"), f"profiles/scan_traffic.json:{key} ({e.get('profile', 'committed rocprofv3 --pmc passes')}; not measured in this run)",
                    e.get("source_sha"))
    except (OSError, ValueError):
        pass
    return None, None, None


def _stale(profile_sha):
    """True when a quoted profile was taken from other kernel sources than this tree's (or does not say which)."""
    from sketchy_amd.build import source_sha
    return profile_sha != source_sha()


N_SIMDS = 1024                 # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
VALU_CYCLES_PER_WAVE_INST = 4  # a wave64 VALU instruction occupies its SIMD (16 lanes wide) for 4 cycles
CLOCK_GHZ = 2.4


def _valu_insts(config, batch, workload="truth"):
    """VALU wave instructions one step issues, from the COMMITTED per-kernel SQ_INSTS_VALU profile (profiles/valu_insts.json,
    written from tools/pmc_all.sh output) -- not measured in this run; None when no matching profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_insts.json")) as f:
            e = json.load(f).get(_profile_key(config, batch, workload))
        if e:
            return {"wave_insts_per_step": e["wave_insts_per_step"], "per_kernel": e.get("per_kernel"), "source_sha": e.get("source_sha"),
                    "source": f"profiles/valu_insts.json:{_profile_key(config, batch, workload)} ({e.get('profile', 'committed rocprofv3 --pmc SQ_INSTS_VALU pass')}; "
                              "not measured in this run)"}
    except (OSError, ValueError, KeyError):
        pass
    return None


def _usable_cores():
    """Host threads this process can really run at once: the affinity mask, capped by the container's CPU quota
    (cgroup v2 cpu.max / v1 cfs quota) -- os.cpu_count() reports the machine, not the container."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _end_to_end(refs, batches, batch_bases, B, read_len, top, n_use, S, step_rows, extra_args=()):
    """value_end_to_end: the C++ host as a user runs it -- `sketchy-hip predict -s` on an uncompressed FASTQ file in /dev/shm with
    the C2 reference as a `.msh` file, rows to a file in /dev/shm -- timed by the host itself (--timing: from the start of its
    parser threads to the last written row; loading the 3.2 GB reference is start-up, reported beside it).  The rows it printed are compared
    with the rows the device-resident path wrote for the same reads (step_rows: (idx, sum) arrays of the first n_use batches)."""
    import re
    import shutil
    import tempfile
    from sketchy_amd import build, mshio
    exe = build.build_host()
    # the files go where there is room for them (reference 3.2 GB + reads 4.7 GB + rows): memory-backed /dev/shm if it has it
    need = int(refs[0]["ref"].size * 8 + n_use * B * (2 * read_len + 12) * 1.2) + (1 << 30)
    where = None
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(cand) and shutil.disk_usage(cand).free > need:
                where = cand
                break
        except OSError:
            pass
    if where is None:
        return {"error": f"no directory with {need / 1e9:.1f} GB free for the reference and read files", "not_run": True}
    d = tempfile.mkdtemp(prefix="skx_e2e_", dir=where)
    try:
        t0 = time.time()
        n_g = refs[0]["ref"].shape[0]
        names = [f"genome{i:05d}.fa" for i in range(n_g)]
        mshio.write_msh(d + "/ref.msh", names, refs[0]["ref"], col_len=refs[0]["col_len"], kmer=16, seed=0)
        with open(d + "/geno.tsv", "w") as f:
            f.write("id\tmlst\tmeca\n" + "".join(f"{n}\tST{i % 97}\t{'R' if i % 3 else 'S'}\n" for i, n in enumerate(names)))
        L = read_len
        with open(d + "/reads.fq", "wb") as f:
            for j in range(n_use):
                seq = batches[j][0].cpu().numpy().reshape(B, L)
                rec = np.empty((B, 8 + L + 3 + L + 1), np.uint8)
                rec[:, :8] = np.frombuffer(b"@read/1\n", np.uint8)
                rec[:, 8:8 + L] = seq
                rec[:, 8 + L:8 + L + 3] = np.frombuffer(b"\n+\n", np.uint8)
                rec[:, 8 + L + 3:8 + 2 * L + 3] = ord("I")
                rec[:, -1] = 10
                rec.tofile(f)
        t_files = time.time() - t0
        # four times: the first pass over a file that was written a moment ago pays for the kernel moving its 1.2 M tmpfs pages to the
        # active list under ten threads (measured with nothing behind the C ABI, tools/frontend_rate.sh: 10 M reads/s the first
        # time, 27-32 M from then on) -- an artefact of generating the input right here; all runs are listed, the median of the later ones counts
        runs = []
        for _ in range(4):
            t1 = time.time()
            with open(d + "/rows.tsv", "wb") as out:
                p = subprocess.run([exe, "predict", "-r", d + "/ref.msh", "-g", d + "/geno.tsv", "-i", d + "/reads.fq", "-s", "-t", str(max(top, 1)),
                                    "--timing", *extra_args], stdout=out, stderr=subprocess.PIPE, text=True, timeout=900)
            wall = time.time() - t1
            m = re.search(r'\{"sketchy_hip_timing".*\}', p.stderr)
            if p.returncode != 0 or not m:
                return {"error": f"sketchy-hip predict failed (rc {p.returncode}): {p.stderr[-500:]}"}
            runs.append((json.loads(m.group(0))["sketchy_hip_timing"], wall))
        # (the first run touches the input's pages for the first time: it is listed, the MEDIAN of the others counts)
        later = sorted(runs[1:], key=lambda r: r[0]["reads_per_s"])
        tm, wall = later[(len(later) - 1) // 2]
        # ---- the same reads gzip-compressed, as read streams usually come: (a) BGZF (bgzip: members of 64 KB with their sizes in the
        # header: inflated by all threads at once), (b) plain gzip (one member: one sequential inflate thread).  The first n_gz batches
        # of the file (compressing 4.7 GB in Python would take minutes); rows must be the first rows of the uncompressed run.
        gz = {}
        try:
            import zlib
            n_gz = min(n_use, 4)
            rec_bytes = 8 + L + 3 + L + 1
            raw = np.fromfile(d + "/reads.fq", np.uint8, count=n_gz * B * rec_bytes)
            t2 = time.time()
            from sketchy_amd import synth as _synth
            _synth.write_bgzf(d + "/reads.bgzf.fq.gz", raw)
            zc = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(d + "/reads.plain.fq.gz", "wb") as f:
                for a in range(0, len(raw), 1 << 26):
                    f.write(zc.compress(memoryview(raw[a:a + (1 << 26)])))
                f.write(zc.flush())
            t_comp = time.time() - t2
            del raw
            for kind in ("bgzf", "plain"):
                best = None
                for _ in range(2):
                    with open(d + f"/rows_{kind}.tsv", "wb") as out:
                        p = subprocess.run([exe, "predict", "-r", d + "/ref.msh", "-g", d + "/geno.tsv", "-i", d + f"/reads.{kind}.fq.gz", "-s", "-t",
                                            str(max(top, 1)), "--timing", *extra_args], stdout=out, stderr=subprocess.PIPE, text=True, timeout=900)
                    m = re.search(r'\{"sketchy_hip_timing".*\}', p.stderr)
                    if p.returncode != 0 or not m:
                        gz[kind] = {"error": f"rc {p.returncode}: {p.stderr[-300:]}"}
                        break
                    tmg = json.loads(m.group(0))["sketchy_hip_timing"]
                    # (a BGZF file is inflated BEFORE the parser threads start -- seconds_open_input --; a plain gzip stream inside the parse
                    # loop.  value = reads / (opening + inflating the input + parser start to last row): what value_end_to_end counts, plus the
                    # inflate.  seconds_stream also holds the one-time set-up of the stream and its page-locked slots (0.4 s: more than
                    # everything else for a sample of a few batches; round 5 quoted that figure)
                    tmg["rate"] = tmg["reads"] / max(tmg.get("seconds_open_input", 0.0) + tmg["seconds_parse_start_to_last_row"], 1e-9)
                    tmg["rate_from_open"] = tmg["reads"] / max(tmg["seconds_stream"], 1e-9)
                    if best is None or tmg["rate"] > best["rate"]:
                        best = tmg
                if best:
                    gz[kind] = {"value": best["rate"], "unit": "reads/s", "reads": best["reads"], "input": best.get("input"),
                                "seconds_open_and_inflate": best.get("seconds_open_input"), "seconds_parse_start_to_last_row": best["seconds_parse_start_to_last_row"],
                                "seconds_from_open_to_last_row_with_setup": best["seconds_stream"], "value_with_setup": best["rate_from_open"],
                                "parse_threads": best.get("parse_threads"),
                                "compressed_GB": round(os.path.getsize(d + f"/reads.{kind}.fq.gz") / 1e9, 3),
                                "best_of": 2}
            gz["compress_s"] = round(t_comp, 1)
        except Exception as e:  # noqa: BLE001
            gz["error"] = f"{type(e).__name__}: {e}"[:300]
        import pandas as pd
        for kind in ("bgzf", "plain"):
            if isinstance(gz.get(kind), dict) and "value" in gz[kind]:
                rz = pd.read_csv(d + f"/rows_{kind}.tsv", sep="\t", header=None, usecols=[0, 1, 2], names=["read", "name", "sum"], dtype={"name": str})
                wi = np.concatenate([a for a, _ in step_rows[:n_gz]]).reshape(-1).astype(np.int64)
                ws = np.concatenate([b for _, b in step_rows[:n_gz]]).reshape(-1).astype(np.uint64)
                gz[kind]["rows_match_device_path"] = bool(len(rz) == len(wi) and np.array_equal(rz["name"].str.slice(6, 11).astype(np.int64).to_numpy(), wi)
                                                          and np.array_equal(rz["sum"].to_numpy().astype(np.uint64), ws))
        rows = pd.read_csv(d + "/rows.tsv", sep="\t", header=None, usecols=[0, 1, 2], names=["read", "name", "sum"], dtype={"name": str})
        got_idx = rows["name"].str.slice(6, 11).astype(np.int64).to_numpy()
        got_sum = rows["sum"].to_numpy().astype(np.uint64)
        want_idx = np.concatenate([a for a, _ in step_rows]).reshape(-1).astype(np.int64)
        want_sum = np.concatenate([b for _, b in step_rows]).reshape(-1).astype(np.uint64)
        ok = bool(len(got_idx) == len(want_idx) and np.array_equal(got_idx, want_idx) and np.array_equal(got_sum, want_sum)
                  and np.array_equal(rows["read"].to_numpy(), np.repeat(np.arange(1, n_use * B + 1), max(top, 1))))
        return {"value": tm["reads_per_s"], "unit": "reads/s", "reads": tm["reads"], "seconds": tm["seconds_parse_start_to_last_row"],
                "batches": tm["batches"], "batch_reads": tm["batch_reads"], "parse_threads": tm["parse_threads"], "format_threads": tm["format_threads"],
                "cpus_pinned_near_device": tm.get("cpus_pinned_near_device"), "device_thread_s": tm.get("device_thread_s"),
                "host_cpus_usable": _usable_cores(), "input": f"uncompressed FASTQ in {where}, {os.path.getsize(d + '/reads.fq') / 1e9:.2f} GB",
                "runs_reads_per_s": [r[0]["reads_per_s"] for r in runs], "best_run_reads_per_s": max(r[0]["reads_per_s"] for r in runs),
                "process_wall_s": wall, "process_reads_per_s": tm["reads"] / wall, "files_written_s": t_files,
                "rows_match_device_path": ok,
                "gz": gz,
                "what": "`sketchy-hip predict -s` (sketchy_amd/host: mapped file cut at record boundaries, parser threads that pack 4-bit "
                        "bases into page-locked slots, skx_stream_submit, formatter threads, rows in order to a file in /dev/shm), timed by "
                        "the host from the start of its parser threads to its last written row; process_wall_s adds HIP start-up, reading the "
                        f"{os.path.getsize(d + '/ref.msh') / 1e9:.1f} GB .msh and uploading it"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(n):
    """`python bench.py --gpus N` from a plain shell: this process has made no GPU call; it starts N fresh ranks and
    hands their output through (rank 0 prints the JSON line), failing if any rank fails."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps (batches of --batch reads)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=7,
                    help="repetitions of the timed region (each: table reset, exactly --steps steps, the table all-reduce); "
                         "`value` is their median, every repetition is listed in `values_all`")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=98304, help="reads per step (one reference scan is amortised over this many reads)")
    ap.add_argument("--api", default="enqueue", choices=["enqueue", "push"],
                    help="device-resident entry point of the timed stream: skx_stream_enqueue_device (halves of consecutive "
                         "batches interleaved) or skx_stream_push_device (one batch per call, round 1's)")
    ap.add_argument("--top", type=int, default=1, help="rows ranked after every read (sketchy default 1)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--profile-all", action="store_true", help="experiment: HIP events around EVERY stage in the timed region (with the "
                    "experiments build and SKX_SPAN_DUMP=1: the stages' device timeline on stderr)")
    ap.add_argument("--coalesce", type=int, default=0, help="set the library option stream_coalesce (1 .. 8) before the stream is created; 0 = leave the default")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip value_cold / value_steady_state / stage breakdown (profiling runs)")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the value_batch_x2 leg (batches of twice --batch reads)")
    ap.add_argument("--e2e-args", default="", help="experiment: extra arguments for the sketchy-hip command of the value_end_to_end leg, e.g. '-b 65536 --pin'")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the value_end_to_end leg (FASTQ file -> sketchy-hip predict -s -> rows file)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity checks of the timed steps (profiling runs only)")
    ap.add_argument("--workload", default="truth", choices=["ancestor", "truth"],
                    help="truth (default, `value`): SURVEY.md 8(d)'s generator -- SNP clone tree with real 16-mer hashes, reads sampled from ONE "
                         "truth strain (a leader emerges, as in a real sample); ancestor: rounds 1-4's -- random-hash clone tree, every read drawn from "
                         "the tree's common ancestor, a 40 000-way near-tie.  The default run reports the second as `value_ancestor`")
    ap.add_argument("--no-truth-leg", "--no-other-workload-leg", dest="no_truth_leg", action="store_true",
                    help="skip the leg of the OTHER workload (a child run of this script: value_ancestor by default, value_truth_strain under --workload ancestor)")
    ap.add_argument("--oracle-steps", default="first,last", help="timed steps whose EVERY row is compared with the CPU oracle (fast checker); "
                    "'none' skips the whole-step oracle check")
    ap.add_argument("--no-shuffle", action="store_true", help="experiment: keep genomes grouped by lineage")
    ap.add_argument("--lineages", type=int, default=0, help="experiment: number of lineages of the synthetic clone tree")
    ap.add_argument("--lognormal", type=float, default=None,
                    help="read lengths log-normal around the config's length with this sigma, 200..50000 (default: the config's)")
    ap.add_argument("--read-len", type=int, default=0, help="experiment: reads of this length instead of the config's (instruction-count fits)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: every rank uses device 0 (exercises the multi-rank control flow on a 1-GPU box)")
    ap.add_argument("--allow-host-allreduce", action="store_true",
                    help="testing only: accept a run whose table was reduced through gloo on the host (RCCL refuses ranks that "
                         "share a device); without it a multi-GPU run that did not reduce through RCCL over all ranks FAILS")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))

    import torch  # first: its bundled HIP runtime must be the one the process ends up with
    from sketchy_amd import api, shard, synth

    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        shard.init_process_group()
    if args.share_gpu:
        local_rank = 0
    if api.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no HIP device {local_rank} (the bench has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = local_rank
    tdev = f"cuda:{local_rank}"

    species, s, read_len, sigma, desc = CONFIGS[args.config]
    if args.lognormal is not None:
        sigma = args.lognormal
    if args.read_len:
        read_len = args.read_len
    B, K, W = args.batch, args.steps, args.warmup
    k, hash_seed, top = 16, 0, args.top
    n_sp = len(species)
    n_total = sum(species)

    # ---- synthetic data (identical reference on every rank; each rank its own shard of the stream)
    t0 = time.time()
    truth = args.workload == "truth"
    refs = [synth.make_reference(n, s, k=k, hash_seed=hash_seed, rng_seed=1 + i, device=tdev, shuffle=not args.no_shuffle,
                                 n_lineages=args.lineages, mode="snp" if truth else "pool") for i, n in enumerate(species)]
    t_ref = time.time() - t0
    # the sample: reads of the ancestor (default) / of ONE truth strain (--workload truth) of every species, every step its own batch,
    # generated straight into HBM.  Several species (C4): the stream is MIXED -- a batch holds reads of all of them, shuffled (one
    # `sketchy predict` run per species over the same reads is the reference's semantics, src/sketchy.rs:81-82)
    n_distinct = min(W + K, 40)
    sources = [torch.from_numpy(r["truth_genome"] if truth else r["genome"]).to(tdev) for r in refs]
    genome_t = sources[0]

    def make_batch(n_reads, seed):
        if n_sp == 1:
            return synth.make_reads_torch(sources[0], n_reads, read_len, err=0.05, rng_seed=seed, lognormal_sigma=sigma, device=tdev)
        per = [n_reads // n_sp + (1 if i < n_reads % n_sp else 0) for i in range(n_sp)]
        parts = [synth.make_reads_torch(sources[i], per[i], read_len, err=0.05, rng_seed=seed * 31 + i, lognormal_sigma=sigma, device=tdev)
                 for i in range(n_sp)]
        return synth.mix_reads_torch(parts, rng_seed=seed)
    batches = [make_batch(B, 1000 + 1000 * rank + i) for i in range(n_distinct)]
    batch_bases = [int(o[-1].item()) for _, o in batches]
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    R = api.ReferenceSketch([r["ref"] for r in refs], [r["col_len"] for r in refs], k=k, seed=hash_seed, device=dev)
    if args.coalesce:
        api.set_option("stream_coalesce", args.coalesce)
    coalesce_policy = api.get_option("stream_coalesce")
    S = api.SumOfSharedHashes(R, top=top, max_batch_reads=B, max_batch_bases=max(batch_bases))
    rows = max(top, 1) * n_sp
    # rows of every step are kept (the timed steps are checked afterwards)
    d_ti = torch.zeros((W + K, B, rows), dtype=torch.int32, device=tdev)
    d_ts = torch.zeros((W + K, B, rows), dtype=torch.int64, device=tdev)
    reducer = shard.TableReducer(dev)  # (N > 1: under a watchdog -- a rank whose peers never arrive exits with code 86 and says so)
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    # skx_stream_enqueue_device (default): the sketch of batch i + 1 is queued before the host waits for batch i's summary;
    # --api push: skx_stream_push_device, both halves of a batch per call (round 1's entry point).  Same rows, same table.
    def step(i, slot=None, call=None):
        j = i % n_distinct
        slot = i if slot is None else slot
        bases, offs = batches[j]
        call = call or (S.enqueue_device if args.api == "enqueue" else S.push_device)
        call(bases.data_ptr(), offs.data_ptr(), B, batch_bases[j], d_ti[slot].data_ptr() if top else None,
             d_ts[slot].data_ptr() if top else None)

    # ---- warmup (untimed), then a fresh table: the timed stream starts like a new sample
    for i in range(W):
        step(i)
    S.sync()
    if world > 1:
        reducer.allreduce(S)  # the first collective on a fresh RCCL communicator pays the connection setup
    S.reset()
    S.profile()  # clear counters
    if not args.no_profile:
        S.set_profiling(1 if args.profile_all else 2)  # the timed region records HIP events around the roofline kernel only (every stage: -2 %)

    # ---- timed region: exactly K steps + the final table all-reduce, from a fresh table; repeated --reps times (a single
    # 26 ms shot was the whole headline before) and reported as the median, every repetition listed
    reps = max(1, args.reps)
    rep_s, rep_ar_ms, rep_own_s = [], [], []
    for rep in range(reps):
        if rep:
            S.reset()
        shard.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(W, W + K):
            step(i)
        reducer.allreduce(S)  # (flushes the enqueued batch; its own wall time is kept in reducer.last_ms)
        S.sync()
        torch.cuda.synchronize()
        own = time.perf_counter() - t1
        # (the region is 15 ms: a gloo barrier inside it would be 3-7 % of an N-rank number.  Every rank's own time -- from the
        # common start, through its K steps and the RCCL all-reduce, which itself waits for the slowest rank -- is gathered AFTER
        # the region; `value` = all ranks' reads / the slowest rank's time)
        rep_s.append(shard.max_over_ranks(own))
        rep_own_s.append(own)
        rep_ar_ms.append(shard.max_over_ranks(reducer.last_ms or 0.0))
    order = sorted(range(reps), key=lambda i: rep_s[i])
    med = order[(reps - 1) // 2]  # (lower median: an actual repetition, so ms_per_step * K is a measured time)
    elapsed = rep_s[med]
    prof = S.profile() if not args.no_profile else None
    S.set_profiling(False)
    table_final = S.table()
    stats = S.stats()
    per_rank_values = shard.gather_floats(K * B / rep_own_s[med])

    total_reads = K * B * world
    value = total_reads / elapsed
    out = {
        "metric": "reads/sec streamed (s=10000,k=16,40k-genome ref)" if args.config == "c2" else f"reads/sec streamed ({args.config})",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "values_all": [total_reads / t for t in rep_s], "reps": reps,
        "values_per_rank": per_rank_values,
        "config": {"workload": desc,
                   "workload_kind": ("SURVEY 8(d): SNP clone tree (lineage divergence 1 %, strain 0.05 %), sketches = bottom-s of the strains' REAL 16-mer "
                                     "hashes, reads sampled from ONE truth strain per species, 5 % substitution errors") if truth else
                                    ("random-hash clone tree, reads sampled from the tree's common ancestor, 5 % substitution errors "
                                     "(rounds 1-4's stream: a near-tie of all genomes; NOT SURVEY 8(d)'s)"),
                   "stream": f"{K} distinct batches = the first {K * B} reads of a sample per GPU, table fresh at the first timed step",
                   "reads_per_step": B, "read_len": read_len, "read_len_lognormal_sigma": sigma,
                   "mean_read_len": round(float(np.mean(batch_bases)) / B, 1), "n_species": n_sp, "n_genomes": species, "s": s, "k": k,
                   "top": top, "rccl_ranks": reducer.rccl_ranks if reducer.how == "rccl" else 0,
                   "allreduce": dict(reducer.report(), ms=rep_ar_ms[med],
                                     what="final sum all-reduce of the u64 table, inside the timed region (max over ranks; includes "
                                          "queueing the last batch's passes and waiting for them)"),
                   "api": "skx_stream_enqueue_device + final sync" if args.api == "enqueue" else "skx_stream_push_device + final sync",
                   "kmer_prefilter": dict(zip(("keys", "table_bytes"), R.kmer_filter)),
                   "rare_index": R.rare_index, "long_lists": R.patterns,
                   "static_dense_dictionary": dict(zip(("on", "hashes"), R.static_dense)),
                   "batches_per_pass": (f"up to {coalesce_policy} (option stream_coalesce: batches enqueued back to back share one scan of "
                                        f"the reference; {stats['passes_shared']} shared passes so far on this stream)") if args.api == "enqueue" else "1",
                   "parallelism": f"reads sharded x{world}, reference replicated, final table all-reduce via {reducer.how}"},
    }
    err = None
    # a multi-GPU line is only valid when RCCL really reduced the table over `world` ranks (a run that silently fell back to
    # the host path would otherwise print a green line): --allow-host-allreduce (testing: ranks sharing one device, which
    # RCCL refuses) is the only way to run N > 1 without it
    if world > 1 and not args.allow_host_allreduce and not (reducer.how == "rccl" and reducer.rccl_ranks == world):
        err = (f"--gpus {world}: the table was not reduced by RCCL over {world} ranks (transport {reducer.how}, RCCL counted "
               f"{reducer.rccl_ranks} ranks; {reducer.err})")

    # ---- parity of what was timed (every rank checks its own shard)
    if not args.no_check and top:
        # (a) last timed step vs the same reads in CHECK_CUT-read pushes on a second stream that starts from the table
        #     before that step (replayed: same batches, same order, table reset as in the timed run)
        ti_last = d_ti[W + K - 1].cpu().numpy().view(np.uint32).copy()
        ts_last = d_ts[W + K - 1].cpu().numpy().view(np.uint64).copy()
        S.reset()
        for i in range(W, W + K - 1):
            step(i, slot=0)
        S.sync()
        table_pre = S.table()
        j = (W + K - 1) % n_distinct
        bases_j, offs_j = batches[j]
        cut_lo = torch.arange(0, B, CHECK_CUT, device=tdev)
        cut_bases = int((offs_j[torch.clamp(cut_lo + CHECK_CUT, max=B)] - offs_j[cut_lo]).max().item())
        V = api.SumOfSharedHashes(R, top=top, max_batch_reads=min(B, CHECK_CUT), max_batch_bases=max(cut_bases, 1))
        V.table_add(table_pre)
        v_ti = torch.zeros((B, rows), dtype=torch.int32, device=tdev)
        v_ts = torch.zeros((B, rows), dtype=torch.int64, device=tdev)
        for a in range(0, B, CHECK_CUT):
            n = min(CHECK_CUT, B - a)
            o = offs_j[a:a + n + 1]
            V.push_device(bases_j.data_ptr(), o.data_ptr(), n, int((o[-1] - o[0]).item()), v_ti[a:].data_ptr(), v_ts[a:].data_ptr())
        V.sync()
        cut_ok = bool(np.array_equal(v_ti.cpu().numpy().view(np.uint32), ti_last) and np.array_equal(v_ts.cpu().numpy().view(np.uint64), ts_last))
        table_ok = True
        if world == 1:
            table_ok = bool(np.array_equal(V.table(), table_final))
        V.close()
        del v_ti, v_ts
        out["parity"] = {"last_timed_step_vs_4096_read_cuts": cut_ok, "final_table_vs_4096_read_cuts": table_ok if world == 1 else None}
        if not (cut_ok and table_ok):
            err = "rows / table of the last timed step differ from the same reads pushed in 4096-read cuts"

    if rank == 0:
        pass_bytes = R.pass_bytes
        if prof and prof["scan"]["launches"]:
            scan_ms = prof["scan"]["ms"] / prof["scan"]["launches"]
            achieved = pass_bytes / (scan_ms * 1e-3) / 1e9
            traffic, traffic_src, traffic_sha = _pmc_traffic(args.config, B, args.workload)
            launches_per_step = prof["scan"]["launches"] / (K * reps)
            step_gbs = pass_bytes * launches_per_step / (elapsed / K) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "traffic_stale": _stale(traffic_sha) if traffic is not None else None,  # the committed PMC profile predates this tree's kernels
                               "kernel": "scan_lean_kernel", "avg_launch_ms": scan_ms, "launches_per_step": launches_per_step,
                               "algorithmic_bytes_per_launch": pass_bytes,
                               "quoted_not_measured": {"pattern_ceiling": {"reads_alone": 0.81, "reads_nontemporal": 0.92, "with_result_slabs": 0.63,
                                                                           "with_compact_m_atomics_nt": 0.81, "unit": "fraction of peak"},
                                                       "source": "profiles/r03f_stream_rates.txt (tools/ubench/stream_rates.hip: this kernel's "
                                                                 "geometry as a pure stream over 3.2 GB, measured on an MI355X in round 3) -- "
                                                                 "constants of a committed micro-benchmark, NOT measured in this run"},
                               "note": "achieved = algorithmic bytes (8*s*N) / HIP-event time of the kernel on its own stream in the timed "
                                       "region, where the sketch of the next batch and the ranking of the previous one run beside it "
                                       "(multi-stream pipeline); `isolated` = the same kernel alone.  launches_per_step < 1: enqueued "
                                       "batches share launches (option stream_coalesce)",
                               "whole_step": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
                                              "note": "algorithmic bytes x launches_per_step over the whole step time (sketch, dictionary, ranking "
                                                      "included): what the step needs from HBM for the reference, not a quality figure -- it "
                                                      "falls as batches share scans while the reads/s go up"}}
        vi = _valu_insts(args.config, B, args.workload)
        if vi:
            floor_ms = vi["wave_insts_per_step"] * VALU_CYCLES_PER_WAVE_INST / (N_SIMDS * CLOCK_GHZ * 1e9) * 1e3
            out["roofline_valu"] = {"bound": "valu_issue", "wave_insts_per_step": vi["wave_insts_per_step"],
                                    "insts_source": vi["source"], "insts_stale": _stale(vi.get("source_sha")), "per_kernel": vi.get("per_kernel"),
                                    "simds": N_SIMDS, "cycles_per_wave_inst": VALU_CYCLES_PER_WAVE_INST, "clock_ghz": CLOCK_GHZ,
                                    "floor_ms": floor_ms, "frac": floor_ms / (1e3 * elapsed / K),
                                    "note": "VALU wave instructions one step issues (all kernels of the step, shared launches counted at their share; committed rocprofv3 "
                                            "SQ_INSTS_VALU pass, not measured in this run) x 4 cycles per wave64 instruction / "
                                            "(1024 SIMDs x clock): the issue floor of the step; frac = floor / measured step"}
        out["pass_stats"] = stats  # (read, hash) pairs / passes of the last timed push, dictionary size, ...
        if "roofline" in out and stats and stats.get("passes") and stats.get("passes_lean_scan", 0) < stats["passes"]:
            # (a dictionary too dense for the lean kernel's 254-entry slices -- five species' hashes in one mixed stream: C4 -- is
            # scanned by scan_kernel's split-array variant; same bytes, its own kernel)
            out["roofline"]["kernel"] = ("scan_kernel<2040, 0, true> (dense dictionary)" if stats.get("passes_lean_scan", 0) == 0
                                         else "scan_lean_kernel / scan_kernel<2040, 0, true> (mixed)")
        out["setup_s"] = {"reference": round(t_ref, 2), "reads": round(t_gen - t_ref, 2), "total": round(t_setup, 2)}

    # ---- extra legs, outside the contract's timed region
    if not args.no_extra_legs:
        # value_cold: the literal "one fresh ~100k-read sample": table reset, ONE push, first-pass ranking included
        cold = []
        S.profile()
        if not args.no_profile:
            S.set_profiling(1 if args.profile_all else 2)
        for rep in range(5):
            S.reset()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            step(W, slot=0)
            S.sync()
            cold.append(time.perf_counter() - tc)
        cold_s = shard.max_over_ranks(float(np.median(cold)))
        cold_prof = S.profile() if not args.no_profile else None
        S.set_profiling(False)
        sd_static, sd_n = R.static_dense
        if rank == 0 and "roofline" in out and sd_static and sd_n:
            # (static dense dictionary: a pass's scan is queued beside its first batch's sketch -- the synchronised pushes of value_cold no
            # longer show the kernel by itself; skx_stream_scan_alone does)
            ms_alone = S.scan_alone(5)
            ach = R.pass_bytes / (ms_alone * 1e-3) / 1e9
            out["roofline"]["isolated"] = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": ms_alone,
                                           "note": "scan kernel alone on the GPU (skx_stream_scan_alone: 5 launches back to back behind a synchronisation)"}
        elif rank == 0 and cold_prof and cold_prof["scan"]["launches"] and "roofline" in out:
            # the same kernel with nothing beside it (these pushes are synchronised one by one): its own quality, whereas
            # the timed region's figure is stretched by the sketch of the next batch sharing the CUs on purpose
            ms_alone = cold_prof["scan"]["ms"] / cold_prof["scan"]["launches"]
            ach = R.pass_bytes / (ms_alone * 1e-3) / 1e9
            out["roofline"]["isolated"] = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": ms_alone,
                                           "note": "scan kernel alone on the GPU (the synchronised pushes of value_cold)"}
        # value_push_device: the timed stream once more through the other entry point (whichever --api did not select)
        other_call, other_name = (S.push_device, "skx_stream_push_device") if args.api == "enqueue" else (S.enqueue_device, "skx_stream_enqueue_device")
        S.reset()
        shard.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(W, W + K):
            step(i, slot=0, call=other_call)
        S.sync()
        other_s = shard.max_over_ranks(time.perf_counter() - tc)
        if rank == 0:
            out["value_other_api"] = {"value": K * B * world / other_s, "unit": "reads/s", "api": other_name,
                                      "what": "the same K batches from a fresh table through the other device-resident entry point"}
        # value_one_pass_per_batch: the timed stream once more on a stream created with "stream_coalesce" = 1 (every batch scans
        # the reference on its own -- what `value` was before two enqueued batches could share a pass).  Rows must be identical.
        if args.api == "enqueue":
            api.set_option("stream_coalesce", 1)
            try:
                S1 = api.SumOfSharedHashes(R, top=top, max_batch_reads=B, max_batch_bases=max(batch_bases))
            finally:
                api.set_option("stream_coalesce", coalesce_policy)
            for i in range(min(W, 2)):
                step(i, slot=0, call=S1.enqueue_device)
            S1.sync()
            t_one = []
            for rep in range(3):
                S1.reset()
                shard.barrier()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for i in range(W, W + K):
                    step(i, slot=0 if i < W + K - 1 else W + K - 1, call=S1.enqueue_device)
                S1.sync()
                t_one.append(shard.max_over_ranks(time.perf_counter() - tc))
            one_ok = None
            if not args.no_check and top:
                one_ok = bool(np.array_equal(d_ti[W + K - 1].cpu().numpy().view(np.uint32), ti_last) and
                              np.array_equal(d_ts[W + K - 1].cpu().numpy().view(np.uint64), ts_last))
                if not one_ok:
                    err = err or "rows of the last timed step differ between shared passes and one pass per batch"
            if rank == 0:
                t1 = float(np.median(t_one))
                out["value_one_pass_per_batch"] = {"value": K * B * world / t1, "unit": "reads/s", "ms_per_step": 1e3 * t1 / K, "median_of": 3,
                                                   "passes_shared": S1.stats()["passes_shared"], "rows_match_timed_run": one_ok,
                                                   "what": "the same K batches on a stream created with option stream_coalesce = 1: one "
                                                           "scan of the reference per batch"}
            S1.close()
        # value_membership_reused: the timed stream once more on a stream created with the policy "reuse_membership" = 1 (references with a
        # static dense dictionary: the bits the scan finds depend on the reference alone -- this stream scans once per buffer set and keeps
        # them).  NOT `value`: the contract's pass streams the reference every time.  Rows must be identical.
        if args.api == "enqueue" and R.static_dense[0] and R.static_dense[1]:
            api.set_option("reuse_membership", 1)
            try:
                Sr = api.SumOfSharedHashes(R, top=top, max_batch_reads=B, max_batch_bases=max(batch_bases))
            finally:
                api.set_option("reuse_membership", 0)
            for i in range(min(W, 2)):
                step(i, slot=0, call=Sr.enqueue_device)
            Sr.sync()
            t_r = []
            for rep in range(3):
                Sr.reset()
                shard.barrier()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for i in range(W, W + K):
                    step(i, slot=0 if i < W + K - 1 else W + K - 1, call=Sr.enqueue_device)
                Sr.sync()
                t_r.append(shard.max_over_ranks(time.perf_counter() - tc))
            reuse_ok = None
            if not args.no_check and top:
                reuse_ok = bool(np.array_equal(d_ti[W + K - 1].cpu().numpy().view(np.uint32), ti_last) and
                                np.array_equal(d_ts[W + K - 1].cpu().numpy().view(np.uint64), ts_last))
                if not reuse_ok:
                    err = err or "rows of the last timed step differ when the static dense rows are kept between passes"
            Sr.reset()
            torch.cuda.synchronize()
            tcold = []
            for rep in range(5):
                Sr.reset()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                step(W, slot=0, call=Sr.enqueue_device)
                Sr.sync()
                tcold.append(time.perf_counter() - tc)
            if rank == 0:
                tr = float(np.median(t_r))
                out["value_membership_reused"] = {"value": K * B * world / tr, "unit": "reads/s", "ms_per_step": 1e3 * tr / K, "median_of": 3,
                                                  "value_cold": B * world / float(np.median(tcold)), "rows_match_timed_run": reuse_ok,
                                                  "what": "the same K batches from a fresh table on a stream created with option reuse_membership = 1: the "
                                                          "membership bits of the reference's static dense hashes are scanned for once and kept -- "
                                                          "no pass streams the reference again (a side leg: `value` scans every pass, as the contract's "
                                                          "roofline figure assumes)"}
            Sr.close()
        # value_top16: the same K batches from a fresh table with SIXTEEN rows after every read (`sketchy predict -t 16`,
        # src/cli.rs:118-119, src/sketchy.rs:348) -- a side leg of the top-1 line.  Its first row of every read of the last batch must be the
        # timed run's row (rank 0 of 16 = the top-1); all sixteen rows against the oracle: tests/test_gpu_patterns.py, `--top 16` itself.
        if args.api == "enqueue" and top == 1 and n_sp == 1 and not args.no_large_batch:
            S16 = api.SumOfSharedHashes(R, top=16, max_batch_reads=B, max_batch_bases=max(batch_bases))
            d16_i = torch.zeros((2, B, 16), dtype=torch.int32, device=tdev)
            d16_s = torch.zeros((2, B, 16), dtype=torch.int64, device=tdev)

            def step16(i, slot=0):
                bb, oo = batches[i % n_distinct]
                S16.enqueue_device(bb.data_ptr(), oo.data_ptr(), B, batch_bases[i % n_distinct], d16_i[slot].data_ptr(), d16_s[slot].data_ptr())
            for i in range(min(W, 2)):
                step16(i)
            S16.sync()
            t16 = []
            for rep in range(3):
                S16.reset()
                shard.barrier()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for i in range(W, W + K):
                    step16(i, slot=1 if i == W + K - 1 else 0)
                S16.sync()
                t16.append(shard.max_over_ranks(time.perf_counter() - tc))
            ok16 = None
            if not args.no_check:
                ok16 = bool(np.array_equal(d16_i[1, :, 0].cpu().numpy().view(np.uint32), ti_last.reshape(B, -1)[:, 0]) and
                            np.array_equal(d16_s[1, :, 0].cpu().numpy().view(np.uint64), ts_last.reshape(B, -1)[:, 0]))
                s16 = d16_s[1].cpu().numpy().view(np.uint64)
                ok16 = ok16 and bool((s16[:, :-1] >= s16[:, 1:]).all())   # rows in rank order
                if not ok16:
                    err = err or "the first of sixteen rows differs from the top-1 row of the last timed step"
            if rank == 0:
                tt = float(np.median(t16))
                out["value_top16"] = {"value": K * B * world / tt, "unit": "reads/s", "ms_per_step": 1e3 * tt / K, "median_of": 3,
                                      "first_row_matches_timed_run": ok16,
                                      "what": "the same K batches from a fresh table on a stream created with top = 16: sixteen ranked rows after every read"}
            S16.close()
            del d16_i, d16_s
        # value_steady_state: the same stream far from its start (no reset, batches cycled), three regions of >= 0.5 s
        n_long = max(32, int(0.5 / (elapsed / K)))
        S.reset()
        for i in range(W + K):
            step(i, slot=0)
        S.sync()
        steady = []
        for rep in range(3):
            shard.barrier()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for i in range(n_long):
                step(i, slot=0)
            S.sync()
            steady.append(shard.max_over_ranks(time.perf_counter() - tc))
        if rank == 0:
            out["value_cold"] = {"value": B * world / cold_s, "unit": "reads/s", "ms": 1e3 * cold_s, "median_of": 5,
                                 "what": f"table reset, one push of {B} reads, synchronised: the first pass of a new sample (nothing to prune, nothing to overlap)"}
            out["value_steady_state"] = {"value": n_long * B * world / float(np.median(steady)), "unit": "reads/s", "steps": n_long,
                                         "median_of": 3, "all": [n_long * B * world / t for t in steady],
                                         "what": f"the stream continued past {(W + K) * B} reads per GPU without reset (batches cycled)"}
        # value_batch_x2: the same stream in batches twice as large (one reference scan amortised over twice the reads; the
        # rows are the same whatever the batching).  `value` stays on the batch the workload names (one ~100k-read sample per step).
        if not args.no_large_batch:
            B2 = 2 * B
            n2 = 5
            big = [make_batch(B2, 5000 + 1000 * rank + i) for i in range(n2)]
            big_bases = [int(o[-1].item()) for _, o in big]
            S2 = api.SumOfSharedHashes(R, top=top, max_batch_reads=B2, max_batch_bases=max(big_bases))
            d2_ti = torch.zeros((B2, rows), dtype=torch.int32, device=tdev)
            d2_ts = torch.zeros((B2, rows), dtype=torch.int64, device=tdev)

            def step2(i):
                bb, oo = big[i % n2]
                S2.enqueue_device(bb.data_ptr(), oo.data_ptr(), B2, big_bases[i % n2], d2_ti.data_ptr() if top else None, d2_ts.data_ptr() if top else None)
            for i in range(2):
                step2(i)
            S2.sync()
            k2 = max(4, K // 2)
            t_big = []
            for rep in range(3):
                S2.reset()
                shard.barrier()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for i in range(k2):
                    step2(i)
                S2.sync()
                t_big.append(shard.max_over_ranks(time.perf_counter() - tc))
            # the rows of the last step against the same reads in 4096-read cuts (table before that step replayed)
            big_ok = None
            if not args.no_check and top:
                last_i = d2_ti.cpu().numpy().view(np.uint32).copy()
                last_s = d2_ts.cpu().numpy().view(np.uint64).copy()
                S2.reset()
                for i in range(k2 - 1):
                    step2(i)
                S2.sync()
                tp = S2.table()
                bb, oo = big[(k2 - 1) % n2]
                cut_lo2 = torch.arange(0, B2, CHECK_CUT, device=tdev)
                cb = int((oo[torch.clamp(cut_lo2 + CHECK_CUT, max=B2)] - oo[cut_lo2]).max().item())
                V2 = api.SumOfSharedHashes(R, top=top, max_batch_reads=CHECK_CUT, max_batch_bases=max(cb, 1))
                V2.table_add(tp)
                v2 = torch.zeros((B2, rows), dtype=torch.int32, device=tdev)
                v2s = torch.zeros((B2, rows), dtype=torch.int64, device=tdev)
                for a in range(0, B2, CHECK_CUT):
                    n = min(CHECK_CUT, B2 - a)
                    o = oo[a:a + n + 1]
                    V2.push_device(bb.data_ptr(), o.data_ptr(), n, int((o[-1] - o[0]).item()), v2[a:].data_ptr(), v2s[a:].data_ptr())
                V2.sync()
                big_ok = bool(np.array_equal(v2.cpu().numpy().view(np.uint32), last_i) and
                              np.array_equal(v2s.cpu().numpy().view(np.uint64), last_s))
                V2.close()
                del v2, v2s
                if not big_ok:
                    err = err or "rows of a double-size batch differ from the same reads pushed in 4096-read cuts"
            if rank == 0:
                tm = float(np.median(t_big))
                out["value_batch_x2"] = {"value": k2 * B2 * world / tm, "unit": "reads/s", "reads_per_step": B2, "steps": k2, "median_of": 3,
                                         "ms_per_step": 1e3 * tm / k2, "all": [k2 * B2 * world / t for t in t_big],
                                         "rows_match_4096_read_cuts": big_ok,
                                         "what": "the same stream from a fresh table in batches of twice the size (one scan of the reference per "
                                                 "batch is amortised over twice the reads); same per-read rows"}
            S2.close()
            del big, d2_ti, d2_ts
        # value_host_fed: the same stream with every batch coming from PAGE-LOCKED HOST memory through
        # skx_stream_submit (copy of batch i+1 over PCIe while batch i is in the kernels) and the rows going back
        if top:
            n_h = min(n_distinct, 4)
            h_b, h_o = [], []
            for j in range(n_h):
                hb, ho = api.HostBuffer(batch_bases[j], dev), api.HostBuffer((B + 1) * 8, dev)
                hb.view(np.uint8, batch_bases[j])[:] = batches[j][0].cpu().numpy()
                ho.view(np.int64, B + 1)[:] = batches[j][1].cpu().numpy()
                h_b.append(hb); h_o.append(ho)
            h_ti = [api.HostBuffer(B * rows * 4, dev) for _ in range(2)]
            h_ts = [api.HostBuffer(B * rows * 8, dev) for _ in range(2)]
            S.reset()
            for i in range(2):
                S.submit(h_b[i % n_h].ptr, h_o[i % n_h].ptr, B, h_ti[i & 1].ptr, h_ts[i & 1].ptr)
            S.drain()
            n_sub = max(16, int(0.5 / (elapsed / K) / 2))
            shard.barrier()
            tc = time.perf_counter()
            for i in range(n_sub):
                S.submit(h_b[i % n_h].ptr, h_o[i % n_h].ptr, B, h_ti[i & 1].ptr, h_ts[i & 1].ptr)
            S.drain()
            t_host = shard.max_over_ranks(time.perf_counter() - tc)
            if rank == 0:
                bytes_in = float(np.mean(batch_bases[:n_h])) + 8.0 * (B + 1)
                out["value_host_fed"] = {"value": n_sub * B * world / t_host, "unit": "reads/s", "steps": n_sub,
                                         "h2d_GBps_per_gpu": n_sub * bytes_in / t_host / 1e9,
                                         "what": "batches in page-locked host memory, skx_stream_submit (H2D of batch i+1 overlaps the "
                                                 "kernels of batch i), rows copied back to the host; PCIe-bound when the GPU step is "
                                                 "shorter than the copy"}
            # value_host_fed_packed: the same with the bases as 4 bits each (skx_pack_bases on the host, outside the timed
            # loop -- a FASTQ parser would pack as it parses): half of the bytes over PCIe
            from sketchy_amd import _lib
            L = _lib.load()
            h_p = []
            for j in range(n_h):
                hp = api.HostBuffer(batch_bases[j] // 2 + 2, dev)
                pos = int(L.skx_pack_bases(h_b[j].ptr, batch_bases[j], hp.ptr, 0))
                assert pos == batch_bases[j]  # (the synthetic reads hold no whitespace: offsets are the same in bases)
                h_p.append(hp)
            S.set_packed_input(True)
            S.reset()
            for i in range(2):
                S.submit(h_p[i % n_h].ptr, h_o[i % n_h].ptr, B, h_ti[i & 1].ptr, h_ts[i & 1].ptr)
            S.drain()
            rows_packed = h_ti[1].view(np.uint32, B * rows).copy()
            shard.barrier()
            tc = time.perf_counter()
            for i in range(n_sub):
                S.submit(h_p[i % n_h].ptr, h_o[i % n_h].ptr, B, h_ti[i & 1].ptr, h_ts[i & 1].ptr)
            S.drain()
            t_pack = shard.max_over_ranks(time.perf_counter() - tc)
            S.set_packed_input(False)
            # (same reads, same table history as the ASCII warm-up above: the rows of its second batch must be identical)
            S.reset()
            for i in range(2):
                S.submit(h_b[i % n_h].ptr, h_o[i % n_h].ptr, B, h_ti[i & 1].ptr, h_ts[i & 1].ptr)
            S.drain()
            packed_ok = bool(np.array_equal(rows_packed, h_ti[1].view(np.uint32, B * rows)))
            if rank == 0:
                out["value_host_fed_packed"] = {"value": n_sub * B * world / t_pack, "unit": "reads/s", "steps": n_sub,
                                                "h2d_GBps_per_gpu": n_sub * (bytes_in - float(np.mean(batch_bases[:n_h])) / 2) / t_pack / 1e9,
                                                "rows_match_ascii": packed_ok,
                                                "what": "the same with 4-bit packed bases (skx_stream_set_packed_input; packed on the host "
                                                        "before the timed loop): half the bytes over PCIe"}
            if not packed_ok:
                err = err or "rows of a 4-bit packed batch differ from the same batch as ASCII"
            for h in h_b + h_o + h_ti + h_ts + h_p:
                h.free()
        # value_end_to_end: FASTQ file -> C++ host -> rows file (one species, fixed-length reads: the C2 workload)
        if rank == 0 and world == 1 and n_sp == 1 and sigma == 0.0 and top and not args.no_end_to_end:
            n_use = min(n_distinct, 16)
            S.reset()
            e_ti = torch.zeros((n_use, B, rows), dtype=torch.int32, device=tdev)
            e_ts = torch.zeros((n_use, B, rows), dtype=torch.int64, device=tdev)
            for j in range(n_use):
                S.enqueue_device(batches[j][0].data_ptr(), batches[j][1].data_ptr(), B, batch_bases[j], e_ti[j].data_ptr(), e_ts[j].data_ptr())
            S.sync()
            step_rows = [(e_ti[j].cpu().numpy().view(np.uint32), e_ts[j].cpu().numpy().view(np.uint64)) for j in range(n_use)]
            del e_ti, e_ts
            # (a leg that could not RUN -- no room for its 9 GB of files, no compiler for the host binary -- says so and does not
            # fail the line: only rows that differ from the device path's do)
            try:
                e2e = _end_to_end(refs, batches, batch_bases, B, read_len, top, n_use, S, step_rows, tuple(args.e2e_args.split()))
            except Exception as e:  # noqa: BLE001
                e2e = {"error": f"{type(e).__name__}: {e}"[:500], "not_run": True}
            out["value_end_to_end"] = e2e
            g = e2e.get("gz") or {}
            out["value_end_to_end_gz"] = {"bgzf": g.get("bgzf"), "plain_gzip": g.get("plain"), "compress_s": g.get("compress_s"), "error": g.get("error"),
                                          "what": "the same FASTQ (its first four batches) compressed: BGZF -- gzip members of 64 KB whose sizes are in "
                                                  "their headers, inflated by all host threads at once -- and plain single-member gzip (one "
                                                  "sequential inflate thread); sketchy-hip predict -s; value = reads / (opening + inflating the input + "
                                                  "parser start to last row), value_with_setup also counts creating the stream and its page-locked slots"}
            for kind in ("bgzf", "plain"):
                if isinstance(g.get(kind), dict) and g[kind].get("rows_match_device_path") is False:
                    err = err or f"rows printed from the {kind} gzip input differ from the device-resident path's rows"
            if e2e.get("rows_match_device_path") is False:
                err = err or "rows printed by `sketchy-hip predict -s` differ from the device-resident path's rows for the same reads"
        if not args.no_profile:
            # per-stage breakdown from a few extra, untimed steps with every stage bracketed by events
            S.set_profiling(1)
            n_extra = min(4, K)
            for i in range(n_extra):
                step(i, slot=0)
            S.sync()
            sp = S.profile()
            S.set_profiling(False)
            if rank == 0:
                out["stage_ms_per_step"] = {n: v["ms"] / n_extra for n, v in sp.items()}

    # ---- CPU baseline + oracle check of the FIRST timed step (rank 0, N=1 only)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import oracle as orc  # checker / baseline only
        bases_w, offs_w = batches[W % n_distinct]
        n_max = min(B, 4096)
        h_offs = offs_w[:n_max + 1].cpu().numpy().astype(np.uint64)
        h_bases = bases_w[:int(h_offs[-1])].cpu().numpy()
        ti_first = d_ti[W, :n_max].cpu().numpy().view(np.uint32).reshape(n_max, n_sp, max(top, 1))
        ts_first = d_ts[W, :n_max].cpu().numpy().view(np.uint64).reshape(n_max, n_sp, max(top, 1))
        # (i) one thread, the honest equivalent of the reference (single-threaded on this path): read by read until the
        #     budget is spent; every species' table is one independent `sketchy predict` run over the same reads
        n_done, t_cpu = 0, 0.0
        cums = [None] * n_sp
        ok = True
        while n_done < min(64, n_max) and (t_cpu < args.cpu_seconds or n_done < 2):
            a, b = int(h_offs[n_done]), int(h_offs[n_done + 1])
            for sp_i, r in enumerate(refs):
                tc = time.perf_counter()
                e = orc.stream(k, hash_seed, s, r["ref"], r["col_len"], h_bases[a:b], np.array([0, b - a], np.uint64),
                               top_k=max(top, 1), cum=cums[sp_i])
                t_cpu += time.perf_counter() - tc
                cums[sp_i] = e["cum"]
                if top:
                    ok = ok and np.array_equal(e["topk_idx"][0], ti_first[n_done, sp_i]) and np.array_equal(e["topk_sum"][0], ts_first[n_done, sp_i])
            n_done += 1
        out["cpu_baseline"] = {"value": n_done / t_cpu, "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": f"first {n_done} reads of the first timed batch, full reference ({n_total} genomes x {s}), "
                                         f"oracle/orc_stream single thread ({t_cpu:.1f} s)",
                               "host_cpus": os.cpu_count(), "host_cpus_usable": _usable_cores(),
                               "timed_rows_match_oracle": bool(ok) if top else None}
        if top and not ok:
            err = err or "rows of the first timed step differ from the CPU oracle"
        # (ii) all host cores (OpenMP over genomes): the generous upper bound of SURVEY 8(d)(ii); continues the same reads
        ncpu = _usable_cores()
        m_done, t_mt, chunk = 0, 0.0, 8
        ok_mt = True
        while n_done + m_done + chunk <= n_max and t_mt < args.cpu_seconds:
            lo, hi = n_done + m_done, n_done + m_done + chunk
            a, b = int(h_offs[lo]), int(h_offs[hi])
            for sp_i, r in enumerate(refs):
                tc = time.perf_counter()
                e = orc.stream_mt(k, hash_seed, s, r["ref"], r["col_len"], h_bases[a:b], h_offs[lo:hi + 1] - h_offs[lo],
                                  top_k=max(top, 1), cum=cums[sp_i], n_threads=ncpu)
                t_mt += time.perf_counter() - tc
                cums[sp_i] = e["cum"]
                if top:
                    ok_mt = ok_mt and np.array_equal(e["topk_idx"], ti_first[lo:hi, sp_i]) and np.array_equal(e["topk_sum"], ts_first[lo:hi, sp_i])
            m_done += chunk
            if t_mt * 4 < args.cpu_seconds:
                chunk = min(chunk * 2, 1024)
        if m_done:
            out["cpu_baseline_all_cores"] = {"value": m_done / t_mt, "unit": "reads/s", "cores": ncpu, "kind": "port",
                                             "sample": f"the next {m_done} reads, oracle/orc_stream_mt (OpenMP over genomes, {ncpu} threads = the CPUs this "
                                                       f"container may use of the host's {os.cpu_count()}, {t_mt:.1f} s)",
                                             "timed_rows_match_oracle": bool(ok_mt) if top else None}
            if top and not ok_mt:
                err = err or "rows of the first timed step differ from the CPU oracle (all-cores leg)"

    # ---- EVERY row of the first and of the last timed step against the CPU oracle (rank 0, N = 1).  orc_stream_fast (oracle/oracle.c;
    # pinned against the literal loop orc_stream in tests/test_oracle.py) runs the whole timed stream from ITS OWN fresh table:
    # table only for the steps in between, all rows + table where asked (--oracle-steps)
    if rank == 0 and world == 1 and top and not args.no_check and args.oracle_steps != "none":
        from oracle import oracle as orc  # checker only
        want = set()
        for tok in args.oracle_steps.split(","):
            tok = tok.strip()
            want.add(0 if tok == "first" else K - 1 if tok == "last" else int(tok))
        want = {i for i in want if 0 <= i < K}
        tc = time.perf_counter()
        cums = [None] * n_sp
        ok_rows, pairs, distinct, checked = True, 0, 0, 0
        for i in range(max(want) + 1 if want else 0):
            bases_i, offs_i = batches[(W + i) % n_distinct]
            hb, ho = bases_i.cpu().numpy(), offs_i.cpu().numpy().astype(np.uint64)
            full = i in want
            for sp_i, r in enumerate(refs):
                e = orc.stream_fast(k, hash_seed, s, r["ref"], r["col_len"], hb, ho, top_k=max(top, 1), cum=cums[sp_i], rows=full)
                cums[sp_i] = e["cum"]
                if sp_i == 0:
                    pairs += e["stats"]["pairs"]
                    distinct += e["stats"]["distinct_sum"]
                if full:
                    gi = d_ti[W + i].cpu().numpy().view(np.uint32).reshape(B, n_sp, max(top, 1))[:, sp_i]
                    gs = d_ts[W + i].cpu().numpy().view(np.uint64).reshape(B, n_sp, max(top, 1))[:, sp_i]
                    ok_rows = ok_rows and bool(np.array_equal(gi, e["topk_idx"]) and np.array_equal(gs, e["topk_sum"]))
            checked += B if full else 0
        table_ok = None
        if want and max(want) == K - 1:
            table_ok = bool(np.array_equal(np.concatenate(cums), table_final))
        out["oracle_whole_steps"] = {"steps": sorted(want), "rows_compared": checked * n_sp * max(top, 1), "timed_rows_match_oracle": ok_rows,
                                     "final_table_matches_oracle": table_ok, "seconds": round(time.perf_counter() - tc, 1),
                                     "threads": orc.usable_threads(),
                                     "in_range_hashes_per_read": round(pairs / max(1, (max(want) + 1) * B), 2) if want else None,
                                     "what": "orc_stream_fast over the timed stream from its own fresh table: every row of the listed steps and "
                                             "the final table (species 0's in-range hashes per read: before the membership filter)"}
        if not ok_rows:
            err = err or "rows of a whole timed step differ from the CPU oracle (orc_stream_fast)"
        if table_ok is False:
            err = err or "the final table of the timed stream differs from the CPU oracle's (orc_stream_fast)"

    # ---- the OTHER workload (value_ancestor: rounds 1-4's near-tie stream; under --workload ancestor: value_truth_strain), as a child
    # run of this script (its own reference, stream, timed region from a fresh table, whole-step oracle check); not `value`
    other = "ancestor" if args.workload == "truth" else "truth"
    other_key = "value_ancestor" if other == "ancestor" else "value_truth_strain"
    if rank == 0 and world == 1 and not args.no_truth_leg and not args.no_extra_legs:
        # (the child builds its own reference: this process's stream, reference and batches are released first -- at C4 the parent's
        # ~70 GB left the child's reference without room for its bit rows, and with them without its static dictionary: the child of
        # round 6's first c4 line ran the dense-dictionary scan at 32.5 M reads/s where the same command on its own does 42 M)
        S.close()
        reducer.close()
        R.close()
        del d_ti, d_ts, batches, sources, genome_t
        torch.cuda.empty_cache()
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", other, "--config", args.config, "--steps", str(K), "--warmup", str(W),
               "--batch", str(B), "--top", str(top), "--reps", str(min(reps, 5)), "--no-extra-legs", "--cpu-seconds", "0", "--api", args.api,
               "--oracle-steps", args.oracle_steps] + (["--no-check"] if args.no_check else [])
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            child = json.loads(line[-1]) if line else None
        except Exception as e:  # noqa: BLE001
            p, child = None, None
            out[other_key] = {"error": f"{type(e).__name__}: {e}"[:300], "not_run": True}
        if child:
            keep = ("value", "unit", "ms_per_step", "values_all", "reps", "steps", "parity", "oracle_whole_steps", "pass_stats", "parity_error")
            tl = {k_: child[k_] for k_ in keep if k_ in child}
            if "roofline" in child:
                tl["scan"] = {k_: child["roofline"].get(k_) for k_ in ("achieved", "frac", "avg_launch_ms", "launches_per_step")}
            tl["workload"] = child["config"]["workload_kind"]
            out[other_key] = tl
            if child.get("parity_error"):
                err = err or other_key + ": " + child["parity_error"]
        elif p is not None:
            out[other_key] = {"error": f"child run failed (rc {p.returncode}): {p.stderr[-400:]}", "not_run": True}

    # every rank's verdict decides the exit code
    n_bad = shard.sum_over_ranks_int(1 if err else 0)
    if rank == 0:
        if err or n_bad:
            out["parity_error"] = err or f"{n_bad} rank(s) failed the parity check of their timed steps"
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        shard.barrier()
        dist.destroy_process_group()
    reducer.close()
    S.close()
    if n_bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
