#!/bin/bash
# usage (any box, repo root): tools/frontend_rate.sh [n_reads] -- what the C++ host's front-end (mapped FASTQ -> parser threads + 4-bit
# packer -> ordered submit -> formatter threads -> rows) does on this machine's CPUs with NOTHING behind the C ABI (tests/stub with
# SKX_STUB_FAST=1, the library's own packer): the ceiling the end-to-end rate of bench.py's value_end_to_end cannot exceed
N=${1:-400000}
D=$(mktemp -d /dev/shm/skx_fe_XXXX)
X=$(mktemp -d $PWD/gpurun_out/fe_XXXX)  # (/dev/shm is mounted noexec on the GPU boxes)
g++ -O2 -std=c++17 -pthread -DSKX_STUB_NO_PACK -I include -I sketchy_amd/host sketchy_amd/host/sketchy_host.cpp tests/stub/skx_stub.cpp -o $X/sketchy-stub -lz -L sketchy_amd -lsketchy_hip -Wl,-rpath,$PWD/sketchy_amd -Wl,-rpath,/opt/rocm/lib || exit 1
python3 - $D $N <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from mshio import write_msh
d, N = sys.argv[1], int(sys.argv[2]); L = 1500
rng = np.random.default_rng(1)
names = [f"g{i}" for i in range(30)]
write_msh(d + "/ref.msh", names, [np.sort(rng.choice(2 ** 40, 20, replace=False).astype(np.uint64)) for _ in names])
open(d + "/g.tsv", "w").write("id\ta\tb\n" + "".join(f"{x}\tST{i}\tR\n" for i, x in enumerate(names)))
with open(d + "/reads.fq", "wb") as f:
    for a in range(0, N, 100000):
        n = min(100000, N - a)
        rec = np.empty((n, 8 + L + 3 + L + 1), np.uint8)
        rec[:, :8] = np.frombuffer(b"@read/1\n", np.uint8); rec[:, 8:8 + L] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]
        rec[:, 8 + L:8 + L + 3] = np.frombuffer(b"\n+\n", np.uint8); rec[:, 8 + L + 3:8 + 2 * L + 3] = ord("I"); rec[:, -1] = 10
        rec.tofile(f)
PY
for J in ${THREADS:-0 16 12 8 4 2}; do
  A=""; [ $J -gt 0 ] && A="-j $J"
  for REP in 1 2 3; do SKX_STUB_FAST=1 $X/sketchy-stub predict -r $D/ref.msh -g $D/g.tsv -i $D/reads.fq -s $A $EXTRA --timing 2>&1 >/dev/null | python3 -c "
import sys, json, re
txt = sys.stdin.read()
m = re.search(r'\{.*\}', txt)
if not m: print('no timing line:', txt[-300:]); sys.exit(0)
t = json.loads(m.group(0))['sketchy_hip_timing']
print('threads=$J $EXTRA parse=%d format=%d  %.2f M reads/s  (%.3f s)' % (t['parse_threads'], t['format_threads'], t['reads_per_s'] / 1e6, t['seconds_parse_start_to_last_row']))"; done
done
rm -rf $D $X
