#!/bin/bash
# usage (GPU box, repo root): tools/bgzf_rate.sh <tag>  -- the BGZF inflate phase alone on this box's CPUs (a 0.6 GB FASTQ with noisy qualities)
TAG=$1; P=gpurun_out/prof; mkdir -p $P
D=$(mktemp -d -p /dev/shm skx_bgzf_XXXX)
python3 - "$D/r.fq.gz" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from sketchy_amd.synth import write_bgzf
rng = np.random.default_rng(1)
n, L = 200000, 1500
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]
qual = (33 + rng.integers(0, 41, (n, L))).astype(np.uint8)
rec = np.empty((n, 2 * L + 24), np.uint8); rec[:] = ord(" ")
hdr = np.frombuffer(b"@read_000000000 x=1\n", np.uint8)
rec[:, :20] = hdr
ids = np.arange(n)
for d in range(9):
    rec[:, 14 - d] = ord("0") + (ids // 10 ** d) % 10
rec[:, 20:20 + L] = seq; rec[:, 20 + L] = 10; rec[:, 21 + L] = ord("+"); rec[:, 22 + L] = 10; rec[:, 23 + L:23 + 2 * L] = qual; rec[:, 23 + 2 * L] = 10
write_bgzf(sys.argv[1], rec.reshape(-1), level=6)
PY
ls -la $D
g++ -O2 -std=c++17 -pthread -I sketchy_amd/host -I include tools/bgzf_rate.cpp -o gpurun_out/bgzf_rate -lz && gpurun_out/bgzf_rate $D/r.fq.gz | tee $P/${TAG}_bgzf_rate.txt
g++ -O2 -std=c++17 -I sketchy_amd/host tests/stub/inflate_check.cpp -o gpurun_out/inflate_check_bin -lz && gpurun_out/inflate_check_bin speed | tee -a $P/${TAG}_bgzf_rate.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; uptime
rm -rf $D gpurun_out/bgzf_rate gpurun_out/inflate_check_bin
