"""Read-sharding across GPUs: one process per GPU, the reference sketch is
replicated in every GPU's HBM, the read stream is cut into contiguous shards, and the only
exchange is one sum all-reduce of the u64 running table at the end (exact: integer sums).

The control plane (rendezvous, barrier, timing max) uses torch.distributed; the table itself is
reduced by RCCL through the C ABI (skx_stream_allreduce) when GPUs are present, and by a
gloo all-reduce of the host copy otherwise (CPU tests, or if RCCL cannot initialise).
"""
import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank; shards differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


COMM_TIMEOUT_S = 180.0     # what a rank waits for its peers (rendezvous, RCCL bring-up, a collective) before it gives up
COMM_TIMEOUT_EXIT = 86     # SKX_COMM_TIMEOUT_EXIT (include/sketchy_hip.h): the exit code of a rank that gave up


def init_process_group(backend=None, timeout_s=None):
    """Initialise torch.distributed from the torchrun environment (MASTER_ADDR/PORT, RANK, ...).  Every collective of the
    control plane gets `timeout_s` (default COMM_TIMEOUT_S): a rank whose peers never arrive fails instead of hanging."""
    import datetime
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    rank, _, world = env_rank()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        import torch
        backend = "cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=timeout_s or COMM_TIMEOUT_S))
    return dist


class deadline:
    """`with deadline(seconds, "what"):` -- a watchdog thread around a block that may wait for other ranks for ever (a
    collective whose peer died, a rendezvous nobody joins).  When the block has not finished in time the watchdog says which
    rank was stuck in what on stderr and ends the PROCESS with COMM_TIMEOUT_EXIT -- never a re-exec: the process may have
    initialised the GPU; a rank that exits non-zero lets the launcher tear the job down.  ctypes calls and torch collectives
    release the GIL, so the watchdog runs while the main thread is blocked inside them."""

    def __init__(self, seconds, what):
        self.seconds, self.what, self._timer = float(seconds), what, None

    def _expired(self):
        import sys
        rank, local, world = env_rank()
        sys.stderr.write(f"[sketchy_amd.shard] rank {rank} of {world} (local rank {local}): {self.what} did not finish within "
                         f"{self.seconds:.0f} s -- a peer is missing or the fabric is down; exiting with code {COMM_TIMEOUT_EXIT}\n")
        sys.stderr.flush()
        os._exit(COMM_TIMEOUT_EXIT)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self._timer = threading.Timer(self.seconds, self._expired)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def barrier():
    """Control-plane barrier on the CPU (gloo) side."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.zeros(1, dtype=torch.int64)
        dist.all_reduce(t)


def max_over_ranks(x: float) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return x
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks_int(x: int) -> int:
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return x
    t = torch.tensor([x], dtype=torch.int64)
    dist.all_reduce(t)
    return int(t.item())


def gather_floats(x: float):
    """Every rank's x, in rank order (a one-element list for a single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(x)]
    mine = torch.tensor([x], dtype=torch.float64)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


def broadcast_bytes(b: bytes, n: int, src=0) -> bytes:
    import torch
    import torch.distributed as dist
    t = torch.zeros(n, dtype=torch.uint8)
    if dist.get_rank() == src:
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).clone()
    dist.broadcast(t, src=src)
    return bytes(t.numpy().tobytes())


def allreduce_table_host(table: np.ndarray) -> np.ndarray:
    """Sum of u64 tables over ranks through gloo (int64 two's-complement add == u64 add mod 2^64)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return table.copy()
    t = torch.from_numpy(np.ascontiguousarray(table, np.uint64).view(np.int64).copy())
    dist.all_reduce(t)
    return t.numpy().view(np.uint64).copy()


def exclusive_prefix_tables(table: np.ndarray) -> np.ndarray:
    """Sum of the tables of all LOWER ranks (shard r adds it with table_add to reproduce the
    single-stream per-read cumulative sums): all-gather then local prefix."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return np.zeros_like(table)
    mine = torch.from_numpy(np.ascontiguousarray(table, np.uint64).view(np.int64).copy())
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    acc = np.zeros(len(table), np.uint64)
    for r in range(dist.get_rank()):
        acc += parts[r].numpy().view(np.uint64)
    return acc


class TableReducer:
    """Reduces a stream's running table over ranks: RCCL over xGMI through the C ABI, with a
    gloo fallback.  `how` records which transport ran."""

    def __init__(self, device: int, timeout_s=None):
        import torch.distributed as dist
        from . import api
        self.timeout_s = COMM_TIMEOUT_S if timeout_s is None else float(timeout_s)
        self.how = "single"
        self.comm = None
        self.world = 1        # ranks of the job (torch.distributed)
        self.rccl_ranks = 0   # ranks a LIVE RCCL communicator reports (ncclCommCount); 0 whenever RCCL is not the transport
        self.err = None       # why RCCL was not used (fallback path only)
        self.last_ms = None   # wall time of the most recent allreduce() on this rank (collective + its synchronisation)
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        rank, world = dist.get_rank(), dist.get_world_size()
        self.world = world
        self.how = "gloo-host"
        with deadline(self.timeout_s, "the RCCL bring-up of the table reducer"):
            self._bring_up(device, rank, world, api)
        self.say(f"table reducer ready: {self.report()}")

    def say(self, msg):
        """One line per rank on stderr -- also (above all) when the bring-up went wrong: the first N-rank run should not have
        to be debugged from rank 0's summary alone."""
        import sys
        rank, local, _ = env_rank()
        sys.stderr.write(f"[sketchy_amd.shard rank {rank} dev {local}] {msg}\n")
        sys.stderr.flush()

    def _bring_up(self, device, rank, world, api):
        # the C ABI's own watchdog (ncclCommInitRank / the collective inside the library) a little inside ours
        try:
            api.set_option("comm_timeout_ms", int(max(1.0, self.timeout_s - 5.0) * 1000))
        except Exception as e:  # noqa: BLE001  (a library without the option: the Python watchdog still stands)
            self.err = repr(e)
        # every rank first proves it can load RCCL (making an id does; only rank 0's is used): a rank that cannot
        # would otherwise leave the others blocked inside ncclCommInitRank
        uid = b"\0" * 128
        can_load = 1
        try:
            mine = api.Comm.unique_id()
            if rank == 0:
                uid = mine
        except Exception as e:  # noqa: BLE001  (RCCL unavailable: everybody stays on the host path)
            self.err = repr(e)
            can_load = 0
        all_can = sum_over_ranks_int(can_load) == world
        uid = broadcast_bytes(uid, 128, src=0)  # always: the other ranks are waiting in it
        if all_can and any(uid):
            try:
                self.comm = api.Comm(device, rank, world, uid)
                self.how = "rccl"
                self.rccl_ranks = self.comm.n_ranks
            except Exception as e:  # noqa: BLE001  (RCCL unavailable: stay on the host path)
                self.err = repr(e)
                self.comm = None
        # every rank must take the same path
        ok = sum_over_ranks_int(1 if self.comm is not None else 0)
        if ok != world:
            if self.comm is not None:
                self.comm.close()
            self.comm, self.how, self.rccl_ranks = None, "gloo-host", 0

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    @property
    def n_ranks(self) -> int:
        """Ranks the table is summed over, whatever the transport (1 for a single process)."""
        return self.world

    def report(self) -> dict:
        """What bench.py prints as config.allreduce: transport, ranks RCCL itself counted (0 unless how == "rccl"), and the
        time of the last collective."""
        return {"how": self.how, "ranks": self.rccl_ranks if self.how == "rccl" else 0, "world": self.world,
                "ms": self.last_ms, "fallback_reason": self.err if self.how == "gloo-host" else None}

    def allreduce(self, stream):
        import time
        if self.how == "single":
            self.last_ms = 0.0
            return
        t0 = time.perf_counter()
        with deadline(self.timeout_s, f"the table all-reduce ({self.how})"):
            if self.comm is not None:
                stream.allreduce(self.comm)  # (flushes the stream's pending batch, then ncclAllReduce + stream synchronisation)
            else:
                mine = stream.table()
                total = allreduce_table_host(mine)
                stream.table_add(total - mine)  # u64 wrap-around arithmetic is exact here
        self.last_ms = 1e3 * (time.perf_counter() - t0)
