#!/usr/bin/env python3
"""bench.py -- reads/sec of the streaming read-vs-reference MinHash path on MI355X.

One "step" = one skx_stream_push_device of a batch of B synthetic reads that are already
resident in HBM: sketch every read, score it against every genome of the resident reference
sketch (one or more scans of the s x N matrix), update the running sum-of-shared-hashes table
and rank the top row(s) after every read -- the whole loop body of the reference's
_sum_of_shared_hashes (src/sketchy.rs:328-354), nothing skipped.

Workload (BASELINE.json configs): default "c2" = 100k x 1.5 kb reads vs a 40 000-genome
s=10 000 k=16 reference (3.2 GB of hashes, the HBM-bound scan the metric is quoted on).
N > 1: one process per GPU (torchrun), reference replicated, the read stream sharded; the only
exchange is the final RCCL all-reduce of the u64 table (inside the timed region).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {
    # name: (n_genomes, s, read_len, description)
    "c0": (500, 1000, 1500, "C0: 1.5 kb reads vs 500-genome s=1000 k=16 sketch (plumbing)"),
    "c1": (5000, 1000, 1500, "C1: 1.5 kb reads vs 5k-genome s=1000 k=16 sketch (cache-resident)"),
    "c2": (40000, 10000, 1500, "C2: 100k x 1.5 kb reads vs 40000-genome s=10000 k=16 sketch (HBM-bound scan)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _pmc_traffic(config, batch):
    """HBM bytes per scan_kernel launch from the committed rocprofv3 PMC passes (profiles/scan_traffic.json:
    FETCH_SIZE x 1024 x 2 [gfx950 under-reports wide streaming reads 2x] + WRITE_SIZE x 1024), for the
    same config and batch; None when no matching profile is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "scan_traffic.json")) as f:
            t = json.load(f)
        return t.get(f"{config}_b{batch}", {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 12 batches (~1.2 M reads at the default batch)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=98304, help="reads per step (one reference scan is amortised over this many reads)")
    ap.add_argument("--top", type=int, default=1, help="rows ranked after every read (sketchy default 1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--no-shuffle", action="store_true", help="experiment: keep genomes grouped by lineage")
    ap.add_argument("--lineages", type=int, default=0, help="experiment: number of lineages of the synthetic clone tree")
    ap.add_argument("--lognormal", type=float, default=0.0,
                    help="read lengths log-normal around the config's length with this sigma, 200..50000 (C4-style mixed stream)")
    ap.add_argument("--small-batch-leg", action="store_true",
                    help="also report the scan kernel's roofline at 4096 reads per launch (roofline_small_batch)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: every rank uses device 0 (exercises the multi-rank control flow on a 1-GPU box)")
    args = ap.parse_args()

    import torch  # first: its bundled HIP runtime must be the one the process ends up with
    from sketchy_amd import api, shard, synth

    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        shard.init_process_group()
    if args.share_gpu:
        local_rank = 0
    if api.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no HIP device {local_rank} (the bench has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = local_rank

    n_genomes, s, read_len, desc = CONFIGS[args.config]
    B = args.batch
    if args.steps is None:
        # ~1.2 M reads by default (the per-GPU shard of BASELINE config C3; 12 batches of the ~100k-read C2 stream): the
        # last batch's ranking has nothing to overlap with, which is ~10 % of a 4-step run and ~3 % of this one
        args.steps = max(12, (12 * 98304) // B)
    k, hash_seed, K, W = 16, 0, args.steps, args.warmup

    # ---- synthetic data (identical reference on every rank; each rank its own shard of the stream)
    t0 = time.time()
    ref = synth.make_reference(n_genomes, s, k=k, hash_seed=hash_seed, rng_seed=1, device=f"cuda:{local_rank}",
                               shuffle=not args.no_shuffle, n_lineages=args.lineages)
    t_ref = time.time() - t0
    n_steps = K + W
    n_distinct = min(n_steps, 8)  # distinct read batches held in HBM; longer runs cycle through them
    bases, offsets = synth.make_reads(ref["genome"], n_distinct * B, read_len, err=0.05, rng_seed=1000 + rank,
                                      lognormal_sigma=args.lognormal)
    t_gen = time.time() - t0

    R = api.ReferenceSketch(ref["ref"], ref["col_len"], k=k, seed=hash_seed, device=dev)
    S = api.SumOfSharedHashes(R, top=args.top, max_batch_reads=B, max_batch_bases=len(bases))
    d_bases = api.DeviceBuffer.from_numpy(bases, dev)
    d_offs = [api.DeviceBuffer.from_numpy(offsets[i * B:(i + 1) * B + 1], dev) for i in range(n_distinct)]
    d_ti = api.DeviceBuffer(B * max(args.top, 1) * 4, dev)
    d_ts = api.DeviceBuffer(B * max(args.top, 1) * 8, dev)
    reducer = shard.TableReducer(dev)
    t_setup = time.time() - t0

    def step(i):
        j = i % n_distinct
        S.push_device(d_bases.ptr, d_offs[j].ptr, B, int(offsets[(j + 1) * B] - offsets[j * B]), d_ti.ptr if args.top else None,
                      d_ts.ptr if args.top else None)

    # ---- warmup (untimed)
    for i in range(W):
        step(i)
    S.sync()
    if world > 1:
        # the first collective on a fresh RCCL communicator pays the connection setup: do it before the clock starts
        reducer.allreduce(S)
        S.reset()
    S.profile()  # clear counters
    if not args.no_profile:
        S.set_profiling(2)  # the timed region records HIP events around the roofline kernel only (every stage: -2 %)

    # ---- timed region: exactly K steps + the final table all-reduce
    shard.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    reducer.allreduce(S)
    S.sync()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = time.perf_counter() - t1
    elapsed = shard.max_over_ranks(elapsed)
    prof = S.profile() if not args.no_profile else None
    S.set_profiling(False)
    stage_prof = None
    if not args.no_profile:
        # per-stage breakdown from a few extra, untimed steps with every stage bracketed by events
        S.set_profiling(1)
        n_extra = min(4, K)
        for i in range(W + K, W + K + n_extra):
            step(i)
        S.sync()
        stage_prof = {n: v["ms"] / n_extra for n, v in S.profile().items()}
        S.set_profiling(False)

    total_reads = K * B * world
    value = total_reads / elapsed

    out = {
        "metric": "reads/sec streamed (s=10000,k=16,40k-genome ref)" if args.config == "c2" else f"reads/sec streamed ({args.config})",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": desc, "reads_per_step": B, "read_len": read_len, "n_genomes": n_genomes, "s": s, "k": k,
                   "top": args.top, "parallelism": f"reads sharded x{world}, reference replicated, final table all-reduce via {reducer.how}"},
    }

    if rank == 0:
        pass_bytes = R.pass_bytes
        if prof and prof["scan"]["launches"]:
            scan_ms = prof["scan"]["ms"] / prof["scan"]["launches"]
            achieved = pass_bytes / (scan_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(args.config, B),
                               "kernel": "scan_kernel", "avg_launch_ms": scan_ms, "launches_per_step": prof["scan"]["launches"] / K,
                               "algorithmic_bytes_per_launch": pass_bytes}
            out["stage_ms_per_step"] = stage_prof  # (from extra untimed steps after the timed region)
        out["setup_s"] = {"reference": round(t_ref, 2), "reads": round(t_gen - t_ref, 2), "total": round(t_setup, 2)}
        # optional, outside the timed region: the same scan kernel at a small batch (4096 reads per launch).  Since
        # the membership filter keeps the dictionary small the fraction is about the same at every batch size.
        if args.small_batch_leg and world == 1 and args.config == "c2" and B > 4096:
            nb_small = int(offsets[4096] - offsets[0])
            Sb = api.SumOfSharedHashes(R, top=args.top, max_batch_reads=4096, max_batch_bases=nb_small)
            d_o = api.DeviceBuffer.from_numpy(offsets[:4097], dev)
            for _ in range(2):
                Sb.push_device(d_bases.ptr, d_o.ptr, 4096, nb_small, None, None)
            Sb.sync(); Sb.profile(); Sb.set_profiling(True)
            for _ in range(6):
                Sb.push_device(d_bases.ptr, d_o.ptr, 4096, nb_small, None, None)
            Sb.sync()
            pb = Sb.profile()
            ms_b = pb["scan"]["ms"] / max(1, pb["scan"]["launches"])
            ach_b = pass_bytes / (ms_b * 1e-3) / 1e9
            out["roofline_small_batch"] = {"reads_per_launch": 4096, "achieved": ach_b, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": ach_b / HBM_PEAK_GBS, "avg_launch_ms": ms_b,
                                           "traffic": _pmc_traffic(args.config, 4096)}
            Sb.close(); d_o.free()

        # ---- CPU baseline + full-size parity sample (rank 0, N=1 only)
        if world == 1 and args.cpu_seconds > 0:
            from oracle import oracle as orc  # checker / baseline only
            S2 = api.SumOfSharedHashes(R, top=max(args.top, 1), max_batch_reads=64, max_batch_bases=int(offsets[64]) + 1)
            n_done, t_cpu, cum = 0, 0.0, None
            exp_idx, exp_sum = [], []
            while n_done < 64 and (t_cpu < args.cpu_seconds or n_done < 4):
                a, b = int(offsets[n_done]), int(offsets[n_done + 1])
                tc = time.perf_counter()
                e = orc.stream(k, hash_seed, s, ref["ref"], ref["col_len"], bases[a:b], np.array([0, b - a], np.uint64),
                               top_k=max(args.top, 1), cum=cum)
                t_cpu += time.perf_counter() - tc
                cum = e["cum"]
                exp_idx.append(e["topk_idx"][0]); exp_sum.append(e["topk_sum"][0])
                n_done += 1
            got = S2.push(bases, offsets[:n_done + 1])
            ok = (np.array_equal(got["topk_idx"], np.array(exp_idx)) and np.array_equal(got["topk_sum"], np.array(exp_sum))
                  and np.array_equal(S2.table(), cum))
            out["cpu_baseline"] = {"value": n_done / t_cpu, "unit": "reads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_done} reads of the same stream, full {n_genomes}x{s} reference, "
                                             f"oracle/orc_stream single thread ({t_cpu:.1f} s)",
                                   "host_cpus": os.cpu_count(), "gpu_matches_cpu_on_sample": bool(ok)}
            if not ok:
                out["parity_error"] = "GPU top rows / table differ from the CPU oracle on the sample"
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        shard.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
